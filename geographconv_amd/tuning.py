"""The ONE table of switches and thresholds of the hot path.

Environment variables read by product code -- four, all about WHICH configuration runs, none about how a kernel runs:

  GEOGCN_GEMM_PRECISION  bf16x3 (default) | f32 | bf16      default of GraphConv(gemm_precision=...) / -gemm-precision
  GEOGCN_HIP_GRAPH       0 (default) | 1                    default of GraphConv(hip_graph=...): capture + replay the step
  GEOGCN_DIST_EXCHANGE   auto (default) | a2a | allgather | agpipe | halo   default of TorchDistComm(exchange=...)
  GEOGCN_DIST_BACKEND    torch (default) | native | staged-gloo   transport of the partitioned path (dist.backend_name)

(`GEOGCN_BUILD_DEFINES` is read by build.py only: ablation builds.)  Two more are TEST SEAMS of the library, read by csrc/ at every
call (csrc/common.h `test_seam_i64`) so that a test's monkeypatch.setenv reaches kernels only large operands reach by default; they
replace the exported debug hook of round 5 and are not configuration:

  GEOGCN_X3_ROWS_MIN_M   rows from which x3_rows_kernel takes an A . B (default 4,096; 32,768 until round 6); tests set 1 so that the model-level oracle
                         tests at fixture sizes (12-wide layers on 96 nodes included) run the split-bf16 kernel the TwitterUS step runs (tests/conftest.py)
  GEOGCN_TN_SLAB_LIMIT   bytes one buffer descriptor is taken to bound in the A^T . B slab kernels (default 2^31 - 1): the fallback's test

Everything else is a module attribute below: a
documented constant with the measurement that set it.  Tests that need an A/B flip the attribute (monkeypatch), nothing
reads the environment behind the caller's back.  Kernel-side constants that used to be getenv() switches are now
`constexpr` next to the kernel they belong to (spmm.hip kRowBlock, xt.hip kDocBlock / kUnitCap, elementwise.hip hw_parts,
gemm.hip wide_bn); the experiments they guarded are recorded in DESIGN_NOTEBOOK.md sections 4.1-4.3 (summaries: DESIGN.md section 4) and were removed from the
code in round 3: the SpMM hub hint (non-temporal tail loads), the split X.W0 forward (dense head GEMM + CSR tail, and its
column-slab variant), the side-stream overlap of the graph product with the gate GEMMs, the X^T sweep's L2 prefetch and
per-XCD rendezvous."""
from __future__ import annotations

import os

# ---- configuration defaults (the four environment variables) --------------------------------------------------------
# 'bf16x3' since round 5: fp32-class split-bf16 products (csrc/gemm_x3.hip) where a kernel takes the shape, exact fp32 elsewhere --
# the whole GPU suite holds it to the tolerances stated for the exact kernels (0 argmax mismatches over the 440,000 TwitterUS rows
# included); 'f32' = the exact fp32 MFMA everywhere (26.98 against 23.32 ms per TwitterUS step on the same box when the switch was made)
GEMM_PRECISION = os.environ.get('GEOGCN_GEMM_PRECISION', 'bf16x3')
HIP_GRAPH = os.environ.get('GEOGCN_HIP_GRAPH', '0') == '1'
DIST_EXCHANGE = os.environ.get('GEOGCN_DIST_EXCHANGE', 'auto')


def dist_backend():
    return os.environ.get('GEOGCN_DIST_BACKEND', 'torch')


# ---- fusion A/B (both on; tests flip them to prove the fused launches equal the separate ones) ----------------------
# highway block: (Z, T) = (H.Wh, sigmoid(H.Wt + bt)), (dWh, dWt), dH in one GEMM launch each (-0.46 ms per TWUS step)
FUSE_GEMMS = True
# gating mix T*Hc + (1-T)*H in the SpMM's epilogue: 'all', 'f32' (default since round 6) = for the fp32 gathered operand only, 'none'.
# (On the bf16 operand the fused epilogue was slower in round 2 -- 1.82 ms against 1.10 + 0.41 at 300 wide; re-measured at the end
#  of round 3, same box, two alternations: 6x600 bf16 step 69.95 -> 69.56 ms, 3x300 bf16 18.06 -> 17.98: now ahead, so 'all'.)
#  Round 6, same box, two alternations: 6x600 bf16 67.99 / 67.98 ms with 'all', 67.65 / 67.67 with 'f32'; 3x300 bf16 17.40 / 17.17: the separate
#  highway_fwd pass is ahead again on the bf16 operand -> 'f32' (the fp32 configurations are unaffected: their operand is fp32).)
FUSE_HIGHWAY = 'f32'

# the highway block's carry gradient G * (1 - T) formed in the epilogue of dH = dZ.Wh^T + dU.Wt^T (geogcn_gemm_kcat_gated_f32) instead
# of written by highway_bwd and read back: 0.53 GB less written per 300-wide block at the TwitterUS size; same bits
FUSE_GATE_CARRY = True

# ... and under the FIRST block the dropout + tanh gradient of the sparse-input layer in the same epilogue
# (geogcn_gemm_kcat_gated_tanhbwd_f32): the act_bwd pass over dH disappears, its bias gradient becomes a column sum of dS0; same bits
FUSE_ACT_BWD = True

# the output layer's softmax in the epilogue of its graph product (geogcn_spmm_csr_softmax_f32): the logits are never written and
# the softmax pass is gone; probabilities agree with the separate pass to rounding (the row sum is taken over another layout)
FUSE_SOFTMAX = True

# bf16 configuration, one GPU: the highway block's H . Wh (bf16 result, the SpMM's operand) and sigmoid(H . Wt + bt) in one launch of
# the bf16 whole-rows kernel -- H read and rounded once (geogcn_gemm_dual_bf16); same bits as the two launches
FUSE_BF16_DUAL = True

# bf16 configuration, one GPU (round 6): the highway block's dH = dZ . Wh^T + dU . Wt^T (+ the carry) in ONE launch of the bf16 whole-rows
# kernel (both reduction segments in LDS, one accumulator) instead of a writing and an accumulating launch: dH written once instead of
# written, read and written (-2.1 GB per 600-wide block at the TwitterUS size).  The sum is associated differently from the two launches
# (one fp32 accumulator over both reductions): equal to rounding, not bit for bit
FUSE_BF16_KCAT = True
# ... and its two weight gradients (dWh, dWt) = H^T . [dZ | dU] in one launch of the bf16 A^T . B kernel (H read once; geogcn_gemm_dual_f32
# with transA = 1 and GEOGCN_GEMM_BF16).  Built, tested -- and OFF: same box, two alternations, configs[4] on one GPU 68.07 / 67.90 ms with
# it against 67.31 / 66.34 without the reverse sweep's fusions, where the one-launch dH alone measured 65.61 against 66.74: the 16 tiles
# of a slab (4 x 2 x 2 of 160 x 320) on one XCD run slower than two launches of 8
FUSE_BF16_DUAL_TN = False

# bf16 configuration, one GPU: highway_bwd stores the branch gradient dS as bf16 (what A^T . dS gathers) instead of fp32 + a cast
# pass (-0.37 ms per 600-wide block; same bits)
FUSE_BF16_DS = True

# dropout after the sparse-input layer in the epilogue of X . W0 (geogcn_spmm_csr_hot_dropout_f32): one launch instead of
# product + mask kernel + apply pass (-0.2 ms per TWUS step); same Philox bits, same arithmetic
FUSE_DROPOUT = True

# ---- X path thresholds ---------------------------------------------------------------------------------------------
# X^T . G is split into a dense N x K head panel on the MFMA pipe and a sparse tail.  K is chosen by a cost model (round 3;
# ops.dense_head_size): the K densest columns cost padded(K) rows of a split-K GEMM (tiles of 128 / 160 rows: 180 columns pay
# for 256), the tail costs its stored entries at the sweep's MARGINAL gather rate.  Rates fitted to same-box measurements of the
# whole product at the TwitterUS shape (profiles/r03_m_head_size.txt): head of 160 / 180 / 256 / 320 / 384 / 480 columns = 2.03 /
# 2.13 / 1.91 / 2.02 / 2.03 / 2.28 ms -- the head GEMM costs ~2.6 us per padded row (101 TFLOP/s of padded work), an entry moved
# from the tail saves ~0.2 ns (the densest tail words are split into many units: their entries cost more than the average
# 0.13 ns).  Until round 3 the rule was a density threshold (3.5 %), which took 180 columns = 256 padded rows with 76 of them empty.
DENSE_HEAD_GEMM_FLOPS = 101e12
DENSE_HEAD_GATHER_BYTES_PER_S = 6.0e12
DENSE_HEAD_SIZES = (160, 256, 320, 384, 480, 512)
DENSE_HEAD_MAX_COLS = 512
# X^T . dS0 goes through the document-blocked sweep (geogcn_xt_dot_f32) from this many stored tail entries on; below,
# dS0 sits in the L2 / Infinity Cache anyway and the plain row gather is as fast
XT_MIN_NNZ = 1_000_000
# ... in column slabs of at most this width (2 x F/64 float4 accumulators per lane: up to 320 columns 1024-thread
# workgroups fit; F = 600 in one piece 5.5 ms, as two slabs of 300 3.0 ms)
XT_MAX_F = 320
# X . W0: rows worked on in an order that puts rows of similar cold length into the same wave (ops.HotCSR.row_order); False = natural
# order (the A/B)
HOT_ROW_ORDER = True
# X . W0 with the hot rows of W0 in LDS (geogcn_spmm_csr_hot_f32) from this many stored entries on
HOT_MIN_NNZ = 2_000_000

# rows of a 300-wide fp32 operand one XCD's 4 MB L2 holds (1.2 KB each): the window inside which a gather is an L2 hit.
# ops.CSR calls a numbering "local" -- and runs a long row's chunks on the XCD that owns the row -- when at least half of
# the long rows' entries lie within this distance of their row
L2_WINDOW_ROWS = 3300

# ---- partitioned path ----------------------------------------------------------------------------------------------
# all-gather scheme: cost of one row in stored-edge equivalents when the row split is balanced (dense work of a row
# ~44 ns over ~0.97 ns per stored edge for the 6 products of a step)
ROW_COST_IN_EDGES = 45.0
# `auto` exchange: take the halo scheme when the largest per-rank halo (rows received per exchange) is at most this
# fraction of the rows an all-gather would deliver to that rank.  From 3 ranks on the alternative is the feature
# repartition (a quarter of the all-gather's bytes at 8 ranks, but ~1 ms more compute per rank and step): the halo wins
# below 0.44 (8 ranks) ... 0.6 (4 ranks) of an all-gather -> 0.5.  At 2 ranks the alternative IS the all-gather, over the one
# link of the pair: any smaller halo pays for its pack launch (60 us against 3.4 ms of wire per exchange) -> 0.9.
# The pinned power-law graph sits at 0.87-0.95, a community graph numbered by label propagation at 0.27 (8 ranks) / 0.67
# (2 ranks) (dist.halo_sizes)
DIST_HALO_MAX_FRACTION = 0.5
DIST_HALO_MAX_FRACTION_2_RANKS = 0.9
# 'agpipe' exchange: feature slabs per all-gather (slab q + 1 on the wire while slab q is multiplied).  4: at F = 300 a slab is 76
# columns = 304 B of a row (3 lines); unmeasured on real links (1-GPU boxes) -- the arithmetic is in DESIGN.md section 5
DIST_AG_SLABS = 4
