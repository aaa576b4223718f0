"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.

NumPy/SciPy restatement of the reference's GCN hot path (gcnmodel.py), used as the checker by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.  Nothing under
``geographconv_amd/`` may import this module: the product path is HIP-only.

PARITY UNPINNED: the arithmetic of the reference lives in Theano 1.0.3 / Lasagne 0.1
(requirements.txt:9,5 -- third-party, not vendored, not installable here: no network), the
reference has no tests / golden vectors / fixtures for this path and its modules
cannot be imported in this container (``import theano`` fails).  This restatement therefore
follows the reference's *call sites* line by line and the published Theano/Lasagne semantics
(SURVEY.md Appendix A), and is checked by (a) fp64 finite differences of its own backward,
(b) analytic known-answer cases and (c) an independent torch-CPU autograd cross-check
(tests/test_oracle.py) -- none of which is the reference itself.

What each function follows:
  spmm / structured_dot      gcnmodel.py:39,130,153  S.structured_dot(CSR, dense): row loop,
                             sequential accumulation in stored index order (scipy csr@dense runs
                             the same loop as Theano's StructuredDotCSR C code)
  forward()  layer 0         gcnmodel.py:29-42,353   tanh(X.W0 + b0)
             dropout         gcnmodel.py:357         lasagne DropoutLayer: x/(1-p) * mask
             highway block   gcnmodel.py:268-288,369 conv gcnmodel.py:114-136, gate DenseLayer
                             sigma(H.Wt+bt) :285-286, mix t*h1+(1-t)*h2 :266
             plain block     gcnmodel.py:372
             output          gcnmodel.py:138-157,374 softmax(A.(H.Wo) + bo)
  metrics()                  gcnmodel.py:376-382,389 row gather, argmax, mean eq, mean CE
  backward()                 what Theano autodiff derives for train_loss (gcnmodel.py:382-387,407)
  adam_step()                lasagne.updates.adam as called at gcnmodel.py:407
  f_train()/f_val()          gcnmodel.py:409-411
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sps


# --------------------------------------------------------------------------------------------
# primitive ops
# --------------------------------------------------------------------------------------------
def spmm(A: sps.csr_matrix, B: np.ndarray) -> np.ndarray:
    """S.structured_dot(A_csr, B_dense) -> dense (gcnmodel.py:39,130,153)."""
    A = A.astype(B.dtype, copy=False)
    return np.asarray(A @ B)


def spmm_t(A: sps.csr_matrix, G: np.ndarray) -> np.ndarray:
    """Gradient of structured_dot wrt the dense operand: structured_dot(A^T, G) where A^T is the
    CSC view of the CSR arrays (Theano's StructuredDotCSC scatter loop)."""
    A = A.astype(G.dtype, copy=False)
    return np.asarray(A.T @ G)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 -> nearest bfloat16 (ties to even), returned as fp32: what `v_cvt_pk_bf16_f32` / a bf16 store does to a finite
    value.  No reference counterpart -- the reference is fp32 throughout; this models BASELINE configs[4] ("bf16 H.W on
    MFMA with fp32 accumulate")."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7fff)
    return ((u + r) & np.uint32(0xffff0000)).view(np.float32)


def dense_product(H, W, gemm_operands=None):
    """T.dot(H, W) (gcnmodel.py:126,149,285).  gemm_operands='bf16': both operands rounded to bf16 (RNE) first, products
    exact, accumulation in (at least) fp32 -- formed here in fp64 and rounded once, i.e. the ideal 'fp32 accumulate' an
    MFMA chain approximates to ~K * 2^-24 relative."""
    if gemm_operands is None:
        return H @ W
    if gemm_operands != 'bf16':
        raise ValueError("gemm_operands must be None or 'bf16'")
    return (bf16_round(H).astype(np.float64) @ bf16_round(W).astype(np.float64)).astype(np.float32)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))         # T.nnet.sigmoid


def softmax_rows(x):
    """T.nnet.softmax: exp(x - rowmax) / rowsum."""
    e = np.exp(x - x.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


# --------------------------------------------------------------------------------------------
# parameter bookkeeping.  Lasagne get_all_params order:
#   highway    : [W0, b0, (Wt, bt, Wh, bh) x blocks, Wo, bo]
#   no highway : [W0, b0, (Wi, bi) x blocks,          Wo, bo]
# --------------------------------------------------------------------------------------------
def n_blocks(hid):
    return len(hid) - 1


def split_params(params, hid, highway):
    nb = n_blocks(hid)
    W0, b0 = params[0], params[1]
    blocks = []
    k = 2
    for _ in range(nb):
        if highway:
            Wt, bt, Wh, bh = params[k:k + 4]
            blocks.append((Wh, bh, Wt, bt))
            k += 4
        else:
            Wi, bi = params[k:k + 2]
            blocks.append((Wi, bi, None, None))
            k += 2
    Wo, bo = params[k], params[k + 1]
    assert k + 2 == len(params)
    return (W0, b0), blocks, (Wo, bo)


def random_params(input_size, hid, C, highway, seed=0, dtype=np.float32, scale=None):
    """Deterministic test weights (NOT the Lasagne initialisers -- those live in the product's
    init.py; fixtures always carry explicit weights)."""
    rng = np.random.RandomState(seed)

    def glorot(n_in, n_out):
        a = np.sqrt(6.0 / (n_in + n_out)) if scale is None else scale
        return rng.uniform(-a, a, size=(n_in, n_out)).astype(dtype)

    params = [glorot(input_size, hid[0]), (0.1 * rng.randn(hid[0])).astype(dtype)]
    width = hid[0]
    for i in range(1, len(hid)):
        if highway:
            params += [glorot(width, width), (-4.0 + 0.1 * rng.randn(width)).astype(dtype),
                       glorot(width, width), (0.1 * rng.randn(width)).astype(dtype)]
        else:
            params += [glorot(width, hid[i]), (0.1 * rng.randn(hid[i])).astype(dtype)]
            width = hid[i]
    params += [glorot(width, C), (0.1 * rng.randn(C)).astype(dtype)]
    return params


# --------------------------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------------------------
def forward(params, X, A, hid, highway=True, p_drop=0.0, mask=None, deterministic=True,
            dtype=np.float32, gemm_operands=None):
    """Full forward pass; returns a cache with every intermediate (for backward and for
    layer-by-layer parity tests).  ``mask`` is the injected Bernoulli(1-p) keep-mask (N x hid[0])
    -- Theano's MRG stream cannot be reproduced.

    ``gemm_operands='bf16'`` is the bf16-AWARE mode for BASELINE configs[4] (no reference counterpart): every dense
    H.W product takes bf16-rounded operands with fp32 accumulation (dense_product), and the product Z that the graph
    convolution gathers is itself STORED as bf16 (the SpMM's arithmetic stays fp32); X.W0, biases, activations, the
    gating mix and the softmax stay fp32 -- exactly where the HIP path's bf16 configuration rounds."""
    dt = np.dtype(dtype)
    if gemm_operands is not None and dt != np.float32:
        raise ValueError("the bf16-aware mode models an fp32 path")
    mm = lambda H_, W_: dense_product(H_, W_, gemm_operands)
    store = (lambda Z_: bf16_round(Z_)) if gemm_operands == 'bf16' else (lambda Z_: Z_)
    params = [np.asarray(p, dtype=dt) for p in params]
    (W0, b0), blocks, (Wo, bo) = split_params(params, hid, highway)
    c = {'blocks': []}
    S0 = spmm(X, W0) + b0                       # gcnmodel.py:39-41
    H0 = np.tanh(S0)                            # gcnmodel.py:42 / :347
    c['H0'] = H0
    if deterministic or p_drop == 0:            # lasagne DropoutLayer (A.2)
        H = H0
        c['drop_scale'] = None
    else:
        assert mask is not None, "training-mode forward with p>0 needs an injected mask"
        keep = dt.type(1.0) - dt.type(p_drop)
        c['drop_scale'] = np.asarray(mask, dtype=dt) / keep
        H = H0 * c['drop_scale']                # x / (1-p) * mask
    c['Hd'] = H
    for (Wh, bh, Wt, bt) in blocks:
        b = {'Hin': H}
        Z = store(mm(H, Wh))                    # gcnmodel.py:126  T.dot(input, W)
        S = spmm(A, Z) + bh                     # gcnmodel.py:130-133
        Hc = np.tanh(S)                         # gcnmodel.py:136
        b['Z'], b['Hc'] = Z, Hc
        if Wt is not None:
            U = mm(H, Wt) + bt                  # gcnmodel.py:285  DenseLayer
            Tg = sigmoid(U).astype(dt)          # gcnmodel.py:286
            b['T'] = Tg
            H = Tg * Hc + (dt.type(1.0) - Tg) * H   # gcnmodel.py:266
        else:
            H = Hc
        b['Hout'] = H
        c['blocks'].append(b)
    c['Hlast'] = H
    Zo = store(mm(H, Wo))                       # gcnmodel.py:149
    So = spmm(A, Zo) + bo                       # gcnmodel.py:153-156
    c['Zo'], c['logits'] = Zo, So
    c['P'] = softmax_rows(So).astype(dt)        # gcnmodel.py:157 / :374
    return c


def metrics(P, idx, y):
    """(mean CE, accuracy, argmax) on gathered rows (gcnmodel.py:376-382,389)."""
    rows = P[idx]
    pred = rows.argmax(-1)
    acc = np.mean(pred == y)
    loss = np.mean(-np.log(rows[np.arange(len(idx)), y]))
    return loss, acc, pred


def reg_penalty(params, hid, highway, reg):
    """regularize_network_params(l1) + (l2), W's only (gcnmodel.py:383-387)."""
    if reg <= 0:
        return 0.0
    Ws = [p for p in params if p.ndim == 2]
    return reg * (sum(np.abs(W).sum() for W in Ws) + sum((W * W).sum() for W in Ws))


# --------------------------------------------------------------------------------------------
# backward (hand-derived)
# --------------------------------------------------------------------------------------------
def backward(params, cache, X, A, train_idx, y_train, hid, highway=True, reg=0.0,
             dtype=np.float32, A_symmetric=None):
    dt = np.dtype(dtype)
    params = [np.asarray(p, dtype=dt) for p in params]
    (W0, b0), blocks, (Wo, bo) = split_params(params, hid, highway)
    P = cache['P']
    N, C = P.shape
    n_tr = len(train_idx)
    # gradient of mean CE over output[train_idx]: Theano's AdvancedIncSubtensor1 ACCUMULATES rows that are indexed
    # more than once (the reference's index vectors are unique, gcnmain.py:207; duplicates are still well defined)
    g_rows = P[train_idx].copy()
    g_rows[np.arange(n_tr), y_train] -= dt.type(1.0)
    g_rows /= dt.type(max(1, n_tr))
    dSo = np.zeros((N, C), dtype=dt)
    np.add.at(dSo, np.asarray(train_idx), g_rows)
    dbo = dSo.sum(axis=0)
    dZo = spmm_t(A, dSo)
    Hl = cache['Hlast']
    dWo = Hl.T @ dZo
    G = dZo @ Wo.T
    grads_blocks = []
    for (Wh, bh, Wt, bt), b in zip(reversed(blocks), reversed(cache['blocks'])):
        Hin, Hc = b['Hin'], b['Hc']
        if Wt is not None:
            Tg = b['T']
            dHc = G * Tg
            dT = G * (Hc - Hin)
            dH = G * (dt.type(1.0) - Tg)
        else:
            dHc = G
            dH = None
        dS = dHc * (dt.type(1.0) - Hc * Hc)
        dbh = dS.sum(axis=0)
        dZ = spmm_t(A, dS)
        dWh = Hin.T @ dZ
        dHz = dZ @ Wh.T
        dH = dHz if dH is None else dH + dHz
        if Wt is not None:
            dU = dT * Tg * (dt.type(1.0) - Tg)
            dbt = dU.sum(axis=0)
            dWt = Hin.T @ dU
            dH = dH + dU @ Wt.T
            grads_blocks.append([dWt, dbt, dWh, dbh])
        else:
            grads_blocks.append([dWh, dbh])
        G = dH
    if cache['drop_scale'] is not None:
        G = G * cache['drop_scale']
    H0 = cache['H0']
    dS0 = G * (dt.type(1.0) - H0 * H0)
    db0 = dS0.sum(axis=0)
    dW0 = spmm_t(X, dS0)
    grads = [dW0, db0]
    for g in reversed(grads_blocks):
        grads += g
    grads += [dWo, dbo]
    if reg > 0:
        for i, p in enumerate(params):
            if p.ndim == 2:
                grads[i] = grads[i] + dt.type(reg) * (np.sign(p) + dt.type(2.0) * p)
    return [np.asarray(g, dtype=dt) for g in grads]


# --------------------------------------------------------------------------------------------
# optimiser: lasagne.updates.adam as called at gcnmodel.py:407
# --------------------------------------------------------------------------------------------
class AdamState:
    def __init__(self, params):
        self.t = 0
        self.m = [np.zeros_like(p) for p in params]
        self.v = [np.zeros_like(p) for p in params]


def adam_step(params, grads, st: AdamState, lr=2e-3, b1=0.9, b2=0.999, eps=1e-8):
    st.t += 1
    dt = params[0].dtype
    t = dt.type(st.t)
    one = dt.type(1.0)
    a_t = dt.type(lr) * np.sqrt(one - dt.type(b2) ** t) / (one - dt.type(b1) ** t)
    out = []
    for i, (p, g) in enumerate(zip(params, grads)):
        st.m[i] = dt.type(b1) * st.m[i] + (one - dt.type(b1)) * g
        st.v[i] = dt.type(b2) * st.v[i] + (one - dt.type(b2)) * g * g
        out.append((p - a_t * st.m[i] / (np.sqrt(st.v[i]) + dt.type(eps))).astype(dt))
    return out


# --------------------------------------------------------------------------------------------
# compiled-function stand-ins (gcnmodel.py:409-411)
# --------------------------------------------------------------------------------------------
def f_train(params, st, X, y_train, y_dev, A, train_idx, dev_idx, hid, highway=True, p_drop=0.0,
            mask=None, reg=0.0, dtype=np.float32):
    """One full-graph fwd + bwd + Adam step.  Returns (new_params, [loss_tr, acc_tr, loss_dev,
    acc_dev, P], grads).  Dev metrics come from the SAME dropout-on pass (gcnmodel.py:378)."""
    c = forward(params, X, A, hid, highway, p_drop, mask, deterministic=False, dtype=dtype)
    l_tr, a_tr, _ = metrics(c['P'], train_idx, y_train)
    l_tr = l_tr + reg_penalty(params, hid, highway, reg)
    l_dev, a_dev, _ = metrics(c['P'], dev_idx, y_dev)
    grads = backward(params, c, X, A, train_idx, y_train, hid, highway, reg, dtype)
    new_params = adam_step([np.asarray(p, dtype=dtype) for p in params], grads, st)
    return new_params, [l_tr, a_tr, l_dev, a_dev, c['P']], grads


def f_val(params, X, A, test_idx, hid, highway=True, dtype=np.float32, gemm_operands=None):
    """Deterministic forward -> (argmax int64, probs[idx]) (gcnmodel.py:392-394,411)."""
    c = forward(params, X, A, hid, highway, deterministic=True, dtype=dtype, gemm_operands=gemm_operands)
    rows = c['P'][test_idx]
    return rows.argmax(-1).astype(np.int64), rows


# --------------------------------------------------------------------------------------------
# one isolated conv layer fwd+bwd (the unit bench.py's metric is quoted on: SURVEY.md §8d (ii))
# --------------------------------------------------------------------------------------------
def conv_layer_fwd_bwd(H, W, b, A, G):
    """ConvolutionDenseLayer2 (gcnmodel.py:114-136) forward + what autodiff derives for it."""
    Z = H @ W
    Hc = np.tanh(spmm(A, Z) + b)
    dS = G * (1.0 - Hc * Hc)
    db = dS.sum(axis=0)
    dZ = spmm_t(A, dS)
    dW = H.T @ dZ
    dH = dZ @ W.T
    return Hc, dH, dW, db


# --------------------------------------------------------------------------------------------
# the training loop (gcnmodel.py:418-450)
# --------------------------------------------------------------------------------------------
def fit(params, st, step, n_epochs=10000, max_down=10, report_k_epoch=1):
    """GraphConv.fit restated line for line (gcnmodel.py:421-449).  ``step(params, st) -> (new_params, outs)`` stands for the
    compiled f_train of gcnmodel.py:430 (its parameter update included; outs[:4] = loss_tr, acc_tr, loss_dev, acc_dev), so
    the loop can be driven by oracle.f_train or by a scripted sequence of dev losses.  Returns a dict: 'best_params' (what
    gcnmodel.py:448-449 restores -- the values AFTER the update of the best epoch, because get_all_param_values at :437
    runs after f_train has applied its updates), 'best_epoch', 'stop_epoch' (last epoch run), 'stopped_early',
    'best_val_loss', 'best_val_acc' and the per-epoch 'history' of (loss_tr, acc_tr, loss_dev, acc_dev, n_validation_down)."""
    import sys
    best_params = None                                       # gcnmodel.py:421
    best_val_loss = sys.maxsize                              # :422
    best_val_acc = 0.0                                       # :423
    n_validation_down = 0                                    # :424
    best_epoch, stopped, history, n = -1, False, [], -1
    for n in range(n_epochs):                                # :429
        params, outs = step(params, st)                      # :430  (updates applied inside f_train)
        l_train, acc_train, l_val, acc_val = (float(v) for v in outs[:4])       # :431-432  .item()
        if l_val < best_val_loss:                            # :434  strict; a NaN never improves
            best_val_loss = l_val                            # :435
            best_val_acc = acc_val                           # :436
            best_params = [np.array(p, copy=True) for p in params]              # :437  values after this epoch's update
            n_validation_down = 0                            # :438
            best_epoch = n
        else:
            n_validation_down += 1                           # :441
        history.append((l_train, acc_train, l_val, acc_val, n_validation_down))
        if n_validation_down > max_down and n > 2 * report_k_epoch * max_down:  # :445
            stopped = True
            break                                            # :447
    return dict(best_params=best_params, best_epoch=best_epoch, stop_epoch=n, stopped_early=stopped,
                best_val_loss=best_val_loss, best_val_acc=best_val_acc, history=history, last_params=params)
