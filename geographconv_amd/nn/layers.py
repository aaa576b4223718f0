"""Lasagne-like layers over the gfx950 kernels.

The constructors, ``get_output_for`` protocol, parameter tags and helper functions follow what
the reference uses from ``lasagne.layers`` (reference gcnmodel.py:8-14, :29-294, :351-414).  What
differs from Lasagne -- by necessity, there is no Theano here -- is that evaluation is eager on the
device and that every layer carries a hand-written ``backward`` (Theano's autodiff derived those
for the reference).  Values flowing between layers are backend ``DMat`` (dense, row-major fp32 on
the GPU) or ``SparseOperand`` (constant CSR input such as X)."""
from __future__ import annotations

from collections import OrderedDict, deque

import numpy as np

from .. import backend, tuning
from . import init as _init
from . import nonlinearities as _nl


# --------------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------------
class Param:
    """Stand-in for a Theano shared variable: host value until bound to a device arena by
    ``ParamStore`` (then ``data``/``grad`` are views of the flat p / g arenas)."""

    def __init__(self, value: np.ndarray, name: str, tags: set):
        self.value = np.asarray(value, dtype=np.float32)
        self.shape = self.value.shape
        self.name = name
        self.tags = set(tags)
        self.data = None      # DMat (matrix) or 1-D tensor (vector) once bound
        self.grad = None
        self._store = None

    def get_value(self):
        if self._store is not None:
            return self._store.read(self)
        return self.value.copy()

    def set_value(self, v):
        v = np.asarray(v, dtype=np.float32)
        if v.shape != self.shape:
            raise ValueError("mismatch: parameter %s has shape %r but value to set has shape %r"
                             % (self.name, self.shape, v.shape))
        if self._store is not None:
            self._store.write(self, v)
        else:
            self.value = v.copy()

    def __repr__(self):
        return "<Param %s %r>" % (self.name, self.shape)


class ParamStore:
    """Flat device arenas for all parameters of a network: values p, gradients g, Adam m / v and
    the `regularizable` mask -- so the optimiser (lasagne.updates.adam, gcnmodel.py:407) is ONE
    kernel over one array and the multi-GPU gradient all-reduce is ONE collective."""

    def __init__(self, params, device):
        K = backend.active()
        import torch
        self.device = device
        self.params = list(params)
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            if len(p.shape) == 2:
                off += p.shape[0] * K.pad4(p.shape[1])
            else:
                off += K.pad4(p.shape[0])
        self.n = off
        z = lambda: torch.zeros(max(off, 4), dtype=torch.float32, device=device)
        self.p, self.g, self.m, self.v, self.regmask = z(), z(), z(), z(), z()
        for p, o in zip(self.params, self.offsets):
            if len(p.shape) == 2:
                r, c = p.shape
                ld = K.pad4(c)
                p.data = K.DMat(r, c, t=self.p[o:o + r * ld].view(r, ld))
                p.grad = K.DMat(r, c, t=self.g[o:o + r * ld].view(r, ld))
                if 'regularizable' in p.tags:
                    self.regmask[o:o + r * ld].view(r, ld)[:, :c] = 1.0
            else:
                c = p.shape[0]
                p.data = self.p[o:o + K.pad4(c)]
                p.grad = self.g[o:o + K.pad4(c)]
            p._store = self
            self.write(p, p.value)

    def write(self, p, v):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(self.device)
        if len(p.shape) == 2:
            p.data.t[:, :p.shape[1]].copy_(t)
        else:
            p.data[:p.shape[0]].copy_(t)

    def read(self, p):
        # .copy(): on a CPU device (gloo tests) .cpu().numpy() would alias the arena
        if len(p.shape) == 2:
            return p.data.t[:, :p.shape[1]].cpu().numpy().copy()
        return p.data[:p.shape[0]].cpu().numpy().copy()

    def read_grad(self, p):
        if len(p.shape) == 2:
            return p.grad.t[:, :p.shape[1]].cpu().numpy().copy()
        return p.grad[:p.shape[0]].cpu().numpy().copy()


# --------------------------------------------------------------------------------------------
# gradient token: "this is already the gradient w.r.t. the PRE-activation"
# --------------------------------------------------------------------------------------------
def y_device(x):
    return x.device if hasattr(x, 'device') else x.fwd.device


class PreAct:
    def __init__(self, m, bias_done=False):
        self.m = m
        self.bias_done = bias_done        # the producer already wrote this layer's bias gradient


class Masked:
    """Gradient token: the true gradient is m * keep_mask * scale (a dropout layer's backward that has not been
    applied yet) -- a DenseLayer below fuses it into its activation-gradient kernel (geogcn_act_bwd_f32 takes the
    mask), saving one pass over the N x hid gradient."""

    def __init__(self, m, keep_mask, scale):
        self.m, self.keep_mask, self.scale = m, keep_mask, scale

    def materialise(self, p):
        return backend.active().dropout_apply(self.m, self.keep_mask, p)


# --------------------------------------------------------------------------------------------
# base classes (lasagne.layers.base)
# --------------------------------------------------------------------------------------------
class Layer:
    def __init__(self, incoming, name=None):
        if isinstance(incoming, tuple):
            self.input_shape = incoming
            self.input_layer = None
        else:
            self.input_shape = incoming.output_shape
            self.input_layer = incoming
        self.name = name
        self.params = OrderedDict()

    @property
    def output_shape(self):
        return self.get_output_shape_for(self.input_shape)

    def get_params(self, unwrap_shared=True, **tags):
        result = list(self.params.keys())
        only = set(tag for tag, value in tags.items() if value)
        if only:
            result = [p for p in result if not (only - self.params[p])]
        exclude = set(tag for tag, value in tags.items() if not value)
        if exclude:
            result = [p for p in result if not (self.params[p] & exclude)]
        return result

    def get_output_shape_for(self, input_shape):
        return input_shape

    def add_param(self, spec, shape, name=None, **tags):
        if name is not None and self.name is not None:
            name = "%s.%s" % (self.name, name)
        tags['trainable'] = tags.get('trainable', True)
        tags['regularizable'] = tags.get('regularizable', True)
        if isinstance(spec, Param):
            param = spec
        else:
            value = spec(shape) if callable(spec) else np.asarray(spec, dtype=np.float32)
            if tuple(value.shape) != tuple(shape):
                raise ValueError("parameter %s: initialiser gave shape %r, expected %r" % (name, value.shape, shape))
            param = Param(value, name, set())
        self.params[param] = set(tag for tag, value in tags.items() if value)
        param.tags = self.params[param]
        return param

    # -- evaluation ---------------------------------------------------------------------------
    def get_output_for(self, input, **kwargs):
        """Lasagne's per-layer forward; here eager on the device."""
        return self.forward(input, None, **kwargs)

    def forward(self, input, tape, **kwargs):
        raise NotImplementedError

    def backward(self, grad, tape, into, **kwargs):
        """-> [grad wrt each input].  `into[i]` is an existing gradient buffer of input i to
        accumulate into (or None)."""
        raise NotImplementedError


class MergeLayer(Layer):
    def __init__(self, incomings, name=None):
        self.input_shapes = [inc if isinstance(inc, tuple) else inc.output_shape for inc in incomings]
        self.input_layers = [None if isinstance(inc, tuple) else inc for inc in incomings]
        self.name = name
        self.params = OrderedDict()

    @property
    def output_shape(self):
        return self.get_output_shape_for(self.input_shapes)


class InputLayer(Layer):
    def __init__(self, shape, input_var=None, name=None):
        self.shape = tuple(shape)
        self.input_var = input_var
        self.name = name
        self.params = OrderedDict()

    @property
    def output_shape(self):
        return self.shape


def _fuse_highway(fp32_operand):
    """tuning.FUSE_HIGHWAY = 'all' | 'f32' (default) | 'none': which SpMM operand formats get the gating mix fused."""
    mode = tuning.FUSE_HIGHWAY
    return mode == 'all' or (mode == 'f32' and fp32_operand)


def _fuse_gemms():
    """tuning.FUSE_GEMMS: the highway block's two weights in one launch (tests flip it for the A/B)."""
    return bool(tuning.FUSE_GEMMS)


def _accumulate(dst, src):
    backend.active().add_inplace(src, dst)
    return dst


# --------------------------------------------------------------------------------------------
# dense / dropout (lasagne.layers.dense, lasagne.layers.noise)
# --------------------------------------------------------------------------------------------
class DenseLayer(Layer):
    """y = nonlinearity(x . W + b)   (lasagne DenseLayer; the highway gate, gcnmodel.py:285)."""

    early_starts = {'fwd': 0, 'bwd': 0}     # exchanges started ahead of their layer (diagnostics / tests)

    def __init__(self, incoming, num_units, W=_init.GlorotUniform(), b=_init.Constant(0.), nonlinearity=_nl.rectify,
                 name=None, **kwargs):
        super().__init__(incoming, name)
        self.nonlinearity = _nl.resolve(nonlinearity)
        self.num_units = int(num_units)
        num_inputs = int(np.prod(self.input_shape[1:]))
        self.W = self.add_param(W, (num_inputs, self.num_units), name="W")
        self.b = None if b is None else self.add_param(b, (self.num_units,), name="b", regularizable=False)
        self._pending = {}          # started exchanges of the two-phase evaluation, keyed by (direction, tape)

    def get_output_shape_for(self, input_shape):
        return (input_shape[0], self.num_units)

    # the graph-convolution subclasses override these three hooks -----------------------------
    def _uses_graph(self, kwargs):
        return False

    def _check_input(self, input):
        K = backend.active()
        if not isinstance(input, K.DMat):
            raise ValueError("Input for this layer must be dense")

    def _matmul(self, input, out, precision=None):
        return backend.active().gemm(input, self.W.data, out=out, precision=precision)

    def _fused_act(self):
        if self.nonlinearity.act is None and self.nonlinearity is not _nl.softmax:
            raise NotImplementedError("nonlinearity %s has no gfx950 kernel yet" % self.nonlinearity.name)
        if self.nonlinearity is _nl.softmax or not self.nonlinearity.fusable:
            return 0                      # applied by its own kernel after the product
        return self.nonlinearity.act

    def forward(self, input, tape, **kwargs):
        K = backend.active()
        self._check_input(input)
        if self.W.data is None:
            raise RuntimeError("parameters are not on the device yet: bind them with ParamStore")
        bias = None if self.b is None else self.b.data
        act = self._fused_act()
        A = kwargs.get('A') if self._uses_graph(kwargs) else None
        prec = kwargs.get('gemm_precision')     # None = backend default (tuning.GEMM_PRECISION: 'bf16x3'; 'f32' = exact fp32 MFMA)
        saved = {'x': input}
        if A is None:
            conv = self._fusable_sibling(input, tape, kwargs)
            if conv is not None:
                # highway block (gcnmodel.py:281-286): this gate and the conv branch read the same input -- one launch
                # multiplies it by [Wh | Wt]: Z = H.Wh raw into the SpMM operand's pitch, T = sigmoid(H.Wt + bt) here
                zf = K.DMat.empty(input.n, conv.num_units, input.device, ld=K.gather_ld(conv.num_units))
                _, y = K.gemm_dual(input, conv.W.data, self.W.data, out0=zf, bias1=bias, act1=act, precision=prec)
                tape[('fused_z', conv)] = (input, zf)
                saved['fused_with'] = conv
            elif self._fusable_sibling(input, tape, kwargs, bf16=True) is not None:
                # the same pair in the bf16 configuration: Z leaves the accumulators as bf16 (the SpMM's operand), T as fp32
                conv = self._fusable_sibling(input, tape, kwargs, bf16=True)
                zf, y = K.gemm_dual_bf16(input, conv.W.data, self.W.data, bias1=bias, act1=act)
                tape[('fused_z', conv)] = (input, zf)
                # (round 6) the reverse sweep forms dH = dZ . Wh^T + dU . Wt^T in one launch where the bf16 whole-rows kernel takes both
                # reductions (tuning.FUSE_BF16_KCAT); the two weight gradients stay separate launches
                if K.kcat_gated_native(input.n, self.W.shape[0], 'bf16'):
                    saved['fused_with'] = conv
            elif isinstance(input, K.DMat):
                y = K.gemm(input, self.W.data, bias=bias, act=act, precision=prec)   # bias + act fused
            else:
                y = self._sparse_input_product(input, bias, act, tape, kwargs)      # sparse input: X.W0
        else:
            comm = kwargs.get('comm')
            if comm is None:
                # Z is gathered row-wise by the SpMM: give it the line-aligned pitch
                n_in = input.n if isinstance(input, K.DMat) else input.shape[0]
                if K.bf16_gather(prec) and isinstance(input, K.DMat) and type(self)._matmul is DenseLayer._matmul:
                    # bf16 configuration: Z goes from the MFMA accumulators to HBM as bf16 -- half the bytes per
                    # gathered row in the SpMM, no fp32 round trip
                    pre = tape.pop(('fused_z', self), None) if tape is not None else None
                    if pre is not None and pre[0] is input and isinstance(pre[1], K.HMat):
                        zf = pre[1]                 # the gate's launch already multiplied by this layer's W (geogcn_gemm_dual_bf16)
                    else:
                        zf = K.HMat(n_in, self.num_units, y_device(input))
                        self._matmul(input, zf, prec)
                else:
                    pre = tape.pop(('fused_z', self), None) if tape is not None else None
                    if pre is not None and pre[0] is input:
                        zf = pre[1]                 # the gate's launch already multiplied by this layer's W
                    else:
                        zf = K.DMat.empty(n_in, self.num_units, y_device(input), ld=K.gather_ld(self.num_units))
                        self._matmul(input, zf, prec)
                    if K.bf16_gather(prec):
                        zf = K.cast_bf16(zf)
                gate = getattr(self, 'highway_gate', None)
                T = tape.get(gate, {}).get('y') if (gate is not None and tape is not None) else None
                # (fp32 operand only: on the bf16 operand the epilogue's T / H loads cost more than the separate pass:
                #  1.82 ms fused against 1.10 + 0.41 ms)
                if (T is not None and self.nonlinearity is _nl.tanh and bias is not None and isinstance(input, K.DMat)
                        and _fuse_highway(isinstance(zf, K.DMat)) and tape[gate]['x'] is input and T.ld == input.ld):
                    # highway block: the gating mix T*Hc + (1-T)*H rides in the SpMM's epilogue (the gate was
                    # evaluated just before this layer); MultiplicativeGatingLayer picks the result up
                    y, saved['highway_out'] = K.spmm_highway(A.fwd, zf, bias, T, input)
                elif (self.nonlinearity is _nl.softmax and tuning.FUSE_SOFTMAX and not kwargs.get('keep_logits')
                      and K.spmm_softmax_ok(zf, self.num_units)):
                    # the output layer: softmax in the epilogue of its graph product; the logits are never written
                    # (get_output(..., keep_logits=True) keeps the two passes and leaves them on the tape)
                    import torch
                    amax = torch.empty(A.fwd.shape[0], dtype=torch.int32, device=zf.device)
                    y = K.spmm_softmax(A.fwd, zf, bias=bias, F=self.num_units, argmax=amax)
                    saved['logits'], saved['argmax'], saved['softmax_done'] = None, amax, True
                else:
                    y = K.spmm(A.fwd, zf, bias=bias, act=act, F=self.num_units)   # A_hat.(H.W) + b, act fused
            else:
                h = self._pending.pop(('fwd', id(tape)), None) or self._exchange_begin_fwd(input, A, comm, bias, act, prec)
                y = comm.graph_spmm_end(h)
        if self.nonlinearity.act is not None and not self.nonlinearity.fusable:
            y = K.bias_act(y, None, self.nonlinearity.act)       # relu / selu: bias was added in the epilogue
        if self.nonlinearity is _nl.softmax and not saved.get('softmax_done'):
            import torch
            amax = torch.empty(y.n, dtype=torch.int32, device=y.device)
            saved['logits'] = y
            y = K.softmax_rows(y, argmax=amax)
            saved['argmax'] = amax
        saved['y'] = y
        if tape is not None:
            tape[self] = saved
        return y

    def _sparse_input_product(self, input, bias, act, tape, kwargs):
        """act(X . W0 + b0) -- and, when the layer's only consumer is a dropout layer that is active in this pass
        (gcnmodel.py:353,357), that dropout in the SAME launch: the product's epilogue draws (or reads the injected) keep
        mask and stores both the activation and its dropped copy; the DropoutLayer then only picks the result up."""
        K = backend.active()
        drop = getattr(self, 'dropout_consumer', None)
        # (only when that DropoutLayer is part of THIS evaluation -- get_output passes the layer set --: drawing its mask for
        #  a sub-network that stops below it would advance its Philox stream once too often)
        if (drop is not None and tape is not None and not kwargs.get('deterministic', False) and drop.p > 0 and drop.rescale
                and tuning.FUSE_DROPOUT and drop in kwargs.get('evaluated_layers', ())):
            injected = kwargs.get('dropout_mask')
            n = input.shape[0]
            pos = {} if injected is not None else drop.stream_position(n, self.num_units, self.W.data.device, kwargs)
            res = K.spmm_x_dropout(input, self.W.data, bias, act, drop.p, mask_in=injected, seed=drop._seed, **pos)
            if res is not None:
                y, yd, mask = res
                if injected is None:
                    drop.advance(kwargs.get('device_counters'))
                tape[('fused_drop', drop)] = (y, yd, mask)
                return y
        return K.spmm_x(input, self.W.data, bias=bias, act=act)

    def _tanh_layer_below(self, tape, kwargs):
        """(layer, its output Y, the dropout's keep mask, 1 / (1 - p)) when this gate's input is the dropped output of a tanh layer
        with a bias fed by the sparse input, and nothing but this highway block reads it (tuning.FUSE_ACT_BWD): the block's input
        gradient can then leave the product's epilogue as that layer's pre-activation gradient."""
        K = backend.active()
        d = self.input_layer
        if not tuning.FUSE_ACT_BWD or not isinstance(d, DropoutLayer) or tape is None or K.bf16_gather(kwargs.get('gemm_precision')):
            return None          # (the bf16 whole-rows kernel has the carry epilogue only: that configuration keeps its act_bwd pass)
        below = d.input_layer
        keep = tape.get(d, {}).get('mask')
        if (keep is None or not isinstance(below, DenseLayer) or below.nonlinearity is not _nl.tanh or below.b is None
                or kwargs.get('consumers', {}).get(d) != 3 or kwargs.get('consumers', {}).get(below) != 1
                or isinstance(tape.get(below, {}).get('x'), K.DMat)):
            return None
        y0 = tape[below]['y']
        # (an injected mask of another dtype / layout / device -- parity runs -- takes the unfused Masked / act_bwd path, which accepts it)
        if (not isinstance(y0, K.DMat) or y0.F % 4 or tuple(keep.shape) != (y0.n, y0.F) or str(keep.dtype) != 'torch.uint8'
                or not keep.is_contiguous() or keep.device != y0.device or not K.kcat_gated_native(y0.n, y0.F, kwargs.get('gemm_precision'))):
            return None
        return below, y0, keep, 1.0 / (1.0 - d.p)

    def _fusable_sibling(self, input, tape, kwargs, bf16=False):
        """The highway block's conv branch, when its H.W can ride in this gate's launch: one GPU, exact-fp32 products (or, `bf16`,
        the bf16 configuration with its bf16 SpMM operand), a graph convolution (whose Z stays linear) with a plain dense product,
        evaluated with a tape."""
        K = backend.active()
        conv = getattr(self, 'highway_conv', None)
        prec = kwargs.get('gemm_precision') or K.GEMM_PRECISION
        if bf16 and not (prec == 'bf16' and K.bf16_gather(prec) and tuning.FUSE_BF16_DUAL):
            return None
        if (conv is None or tape is None or kwargs.get('comm') is not None or kwargs.get('A') is None
                or not isinstance(input, K.DMat) or conv.input_layer is not self.input_layer
                or (not bf16 and prec not in ('f32', 'bf16x3')) or not _fuse_gemms()
                or not conv._uses_graph(kwargs) or type(conv)._matmul is not DenseLayer._matmul
                or conv.W.data is None or conv.W.shape[0] != self.W.shape[0]):
            return None
        return conv

    # ---- two-phase evaluation (multi-GPU): start the exchange of the SpMM operand early, finish later --------
    def _exchange_begin_fwd(self, input, A, comm, bias, act, prec):
        # (a plain H.W product writes the exchange's layout itself -- send panels / my slot of the gathered buffer, bf16 in
        #  the bf16 configuration; other producers hand over a row-major matrix)
        direct = type(self)._matmul is DenseLayer._matmul and isinstance(input, backend.active().DMat)
        z = comm.matmul_target(self.num_units, tag='fwd', precision=prec, direct=direct)
        self._matmul(input, z, prec)
        return comm.graph_spmm_begin(A.fwd, z, bias, act, self.num_units, tag='fwd')

    def can_split(self, kwargs):
        """True when this layer's work contains an exchange that other layers' work may overlap."""
        return self._uses_graph(kwargs) and kwargs.get('A') is not None and kwargs.get('comm') is not None

    def forward_begin(self, input, tape, **kwargs):
        """Z = H.W and the start of its exchange; forward() picks the handle up.  get_output calls this as soon as
        the input is available, BEFORE the layers that precede this one in the topological order but do not feed
        it (the highway gate's GEMM then runs while RCCL moves Z)."""
        self._check_input(input)
        DenseLayer.early_starts['fwd'] += 1
        bias = None if self.b is None else self.b.data
        self._pending[('fwd', id(tape))] = self._exchange_begin_fwd(input, kwargs['A'], kwargs['comm'], bias,
                                                                    self._fused_act(), kwargs.get('gemm_precision'))

    def backward_begin(self, grad, tape, **kwargs):
        """Activation gradient, bias gradient and the start of the exchange of dS; backward() finishes."""
        DenseLayer.early_starts['bwd'] += 1
        self._pending[('bwd', id(tape))] = self._backward_pre(grad, tape, kwargs)

    def _pre_activation_grad(self, grad, tape, kwargs):
        """dS = gradient at the pre-activation, with this layer's bias gradient written on the way (fused into the
        activation-gradient kernel where there is one; skipped when the producer of `grad` already wrote it)."""
        K = backend.active()
        y = tape[self]['y']
        bias_done = isinstance(grad, PreAct) and grad.bias_done
        if isinstance(grad, PreAct):
            dS = grad.m
        elif self.nonlinearity is _nl.softmax:
            raise NotImplementedError("softmax output expects the fused CE gradient (PreAct)")
        elif self.nonlinearity.act == 0:
            dS = grad
        else:
            # (dS is gathered row-wise by the next product -- A^T . dS, or X^T . dS0 under a sparse input: line-aligned pitch)
            uses_graph = (self._uses_graph(kwargs) and kwargs.get('A') is not None) or not isinstance(tape[self]['x'], K.DMat)
            km, sc = (grad.keep_mask, grad.scale) if isinstance(grad, Masked) else (None, 1.0)
            g_in = grad.m if isinstance(grad, Masked) else grad
            out = K.DMat.empty(g_in.n, g_in.F, g_in.device, ld=K.gather_ld(g_in.F)) if uses_graph else None
            if self.b is not None:
                # activation gradient and bias gradient (its column sums) in one pass
                dS = K.act_bwd_colsum(g_in, y, self.nonlinearity.act, self.b.grad, out=out, keep_mask=km, scale=sc)
                bias_done = True
            else:
                dS = K.act_bwd(g_in, y, self.nonlinearity.act, out=out, keep_mask=km, scale=sc)
        if self.b is not None and not bias_done:
            K.colsum(dS, out=self.b.grad)
        return dS

    def _transpose_operand(self, A, grad, kwargs):
        """A^T for the backward product.  Structural zeros: when the incoming gradient is known to be zero outside a
        set of rows (the CE gradient lives on the training rows only), the caller may pass A^T with the other
        COLUMNS removed (`A_bwd_rows_hint`)."""
        hint = kwargs.get('A_bwd_rows_hint')
        if hint is not None and isinstance(grad, PreAct) and hint[0] is self:
            return hint[1]
        return A.bwd

    def _backward_pre(self, grad, tape, kwargs):
        """-> (dS, handle): the pre-activation gradient and the started exchange of it (partitioned graph)."""
        dS = self._pre_activation_grad(grad, tape, kwargs)
        A, comm = kwargs['A'], kwargs['comm']
        if getattr(A, 'head_dense', None) is not None:
            raise ValueError("a graph operand with a split transpose (dense head panel) cannot be exchanged: build "
                             "it with SparseOperand.from_scipy(..., dense_head=False)")
        g = comm.stage_operand(dS, self.num_units, tag='bwd', precision=kwargs.get('gemm_precision'))
        return dS, comm.graph_spmm_begin(self._transpose_operand(A, grad, kwargs), g, None, 0, self.num_units, tag='bwd')

    def backward_mid(self, tape, **kwargs):
        """Between the two exchanges of a started backward: wait for dS, multiply by A^T, start the return exchange."""
        pend = self._pending.get(('bwd', id(tape)))
        if pend is not None:
            kwargs['comm'].graph_spmm_mid(pend[1])

    def backward(self, grad, tape, into, need_input_grad=True, **kwargs):
        K = backend.active()
        x = tape[self]['x']
        A = kwargs.get('A') if self._uses_graph(kwargs) else None
        if A is not None and kwargs.get('comm') is not None:
            dS, handle = self._pending.pop(('bwd', id(tape)), None) or self._backward_pre(grad, tape, kwargs)
            dZ = kwargs['comm'].graph_spmm_end(handle)
            return self._backward_post(x, dZ, into, need_input_grad, kwargs)
        dS = self._pre_activation_grad(grad, tape, kwargs)
        if A is None:
            dZ = dS
        else:
            A_bwd = self._transpose_operand(A, grad, kwargs)
            if A_bwd is A.bwd and getattr(A, 'head_dense', None) is not None:
                dZ = K.spmm_t(A, dS, precision=kwargs.get('gemm_precision'))          # an operand whose transpose is split (dense head panel + CSR tail)
            else:
                as_is = not K.bf16_gather(kwargs.get('gemm_precision')) or not isinstance(dS, K.DMat)   # (HMat: already bf16)
                dZ = K.spmm(A_bwd, dS if as_is else K.cast_bf16(dS))
        return self._backward_post(x, dZ, into, need_input_grad, kwargs, tape)

    takes_gate_carry = True        # (backward sweep: `into[0]` may be an un-formed carry gradient, ops.GateCarry)

    def _backward_post(self, x, dZ, into, need_input_grad, kwargs, tape=None):
        K = backend.active()
        lazy = into[0] if isinstance(into[0], K.GateCarry) else None
        if isinstance(x, K.DMat):
            prec = kwargs.get('gemm_precision')
            gate = getattr(self, 'highway_gate', None)
            if (tape is not None and gate is not None and tape.get(gate, {}).get('fused_with') is self
                    and not tape[gate].get('bwd_done')):
                # the gate comes next in the reverse sweep and reads the same H: it multiplies H^T by [dZ | dU] and
                # [dZ | dU] by [Wh | Wt]^T in one launch each
                tape[('fused_dz', gate)] = dZ
                return [into[0]]
            fused = tape.pop(('fused_dz', self), None) if tape is not None else None
            if fused is not None:
                conv = tape[self]['fused_with']
                if K.bf16_gather(prec) and not tuning.FUSE_BF16_DUAL_TN:          # (bf16 configuration: two launches are faster, tuning.py)
                    K.gemm(x, fused, out=conv.W.grad, transA=True, precision=prec)
                    K.gemm(x, dZ, out=self.W.grad, transA=True, precision=prec)
                else:
                    K.gemm_dual(x, fused, dZ, out0=conv.W.grad, out1=self.W.grad, transA=True, precision=prec)     # dWh, dWt = H^T.[dZ | dU]
                tape[self]['bwd_done'] = True
                if not need_input_grad:
                    return [None]
                # dH = dZ.Wh^T + dU.Wt^T [+ the carry gradient]: one accumulator, one pass over dH
                if lazy is not None:          # ... the carry formed in the epilogue from the block's output gradient and its gate
                    post = self._tanh_layer_below(tape, kwargs)
                    if post is not None:
                        # ... and, under the FIRST block, the dropout + tanh gradient of the layer below in the same epilogue: what
                        # comes out is dS0, that layer's pre-activation gradient (its bias gradient: the column sums)
                        below, y0, keep, scale = post
                        dS0 = K.DMat.empty(y0.n, y0.F, y0.device, ld=K.gather_ld(y0.F))
                        K.gemm_kcat(fused, conv.W.data, dZ, self.W.data, out=dS0, transB=True, gate_carry=lazy, tanh_bwd=(y0, keep, scale), precision=prec)
                        K.colsum_rowblocks(dS0, out=below.b.grad)
                        return [PreAct(dS0, bias_done=True)]
                    return [K.gemm_kcat(fused, conv.W.data, dZ, self.W.data, transB=True, gate_carry=lazy, precision=prec)]
                return [K.gemm_kcat(fused, conv.W.data, dZ, self.W.data, out=into[0], transB=True,
                                    accumulate=into[0] is not None, precision=prec)]
            if lazy is not None and not need_input_grad:
                lazy = None
            K.gemm(x, dZ, out=self.W.grad, transA=True, precision=prec)    # dW = H^T . dZ
            after_dw = kwargs.get('after_dw')
            if after_dw is not None:
                after_dw()            # (partitioned sweep: the sibling convolution's SpMM + return exchange start here)
            if not need_input_grad:
                return [None]
            if lazy is not None:          # dH = dZ . W^T + the carry gradient, formed in the epilogue
                return [K.gemm(dZ, self.W.data, transB=True, precision=prec, gate_carry=lazy)]
            if into[0] is not None:
                return [K.gemm(dZ, self.W.data, out=into[0], transB=True, accumulate=True, precision=prec)]
            return [K.gemm(dZ, self.W.data, transB=True, precision=prec)]  # dH = dZ . W^T
        K.spmm_t(x, dZ, out=self.W.grad, precision=kwargs.get('gemm_precision'))      # dW0 = X^T . dS0
        return [None]


class DropoutLayer(Layer):
    """x / (1-p) * Bernoulli(1-p) mask; identity if deterministic or p == 0 (lasagne DropoutLayer,
    reference gcnmodel.py:357).  The mask comes from a counter-based Philox stream seeded like
    Lasagne seeds its MRG stream (one np.random.randint at construction); pass `dropout_mask=`
    (uint8 keep-mask on the device) through get_output to inject a mask for parity runs."""

    def __init__(self, incoming, p=0.5, rescale=True, name=None, **kwargs):
        super().__init__(incoming, name)
        self._seed = int(np.random.randint(1, 2147462579))
        self._calls = 0
        self._calls_dev = None
        self.p = p
        self.rescale = rescale
        # a sparse-input dense layer directly below may run this dropout in its product's epilogue (DenseLayer._sparse_input_product)
        if isinstance(incoming, DenseLayer) and getattr(incoming, 'dropout_consumer', None) is None:
            incoming.dropout_consumer = self

    def stream_position(self, n, F, device, kwargs):
        """Where in this layer's Philox stream the next mask of an n x F input starts: rows are numbered globally so that
        every rank draws its own slice of the same stream, successive calls advance it by N_total * F elements
        (rank-consistent when F % 4 == 0: Philox yields 4 values per counter).  -> keyword arguments for the mask kernels:
        `offset` (quads) from the host-side call counter, or `calls_dev` / `per_call` / `base` for a captured step, whose
        replays cannot change kernel arguments: offset = (calls * N_total * F + r0 * F) / 4 is then formed on the device."""
        comm = kwargs.get('comm')
        r0 = 0 if comm is None or comm.part is None else comm.part.r0
        N_total = n if comm is None or comm.part is None else comm.part.N
        ctr = kwargs.get('device_counters')          # None | 'sync' | 'captured'  (GraphConv hipGraph path)
        if ctr:
            import torch
            if self._calls_dev is None or self._calls_dev.device != device:
                self._calls_dev = torch.zeros(1, dtype=torch.int64, device=device)
                ctr = 'sync'
            if ctr == 'sync':
                self._calls_dev.fill_(self._calls)
            return dict(calls_dev=self._calls_dev, per_call=N_total * F, base=r0 * F)
        return dict(offset=((self._calls * N_total + r0) * F) // 4)

    def advance(self, ctr):
        """One mask drawn: move the stream on (after the kernel that read the device counter has been enqueued)."""
        if ctr:
            backend.active().counter_add(self._calls_dev, 1)
        self._calls += 1

    def forward(self, input, tape, deterministic=False, dropout_mask=None, **kwargs):
        K = backend.active()
        if deterministic or self.p == 0:
            if tape is not None:
                tape[self] = {'mask': None}
            return input
        if not self.rescale:
            raise NotImplementedError("rescale=False is not used by the reference path")
        pre = tape.pop(('fused_drop', self), None) if tape is not None else None
        if pre is not None and pre[0] is input:
            tape[self] = {'mask': pre[2]}              # the layer below already drew the mask and applied it
            return pre[1]
        mask = dropout_mask
        if mask is None and pre is not None:
            mask = pre[2]        # drawn for this pass already (the stream has moved on), but applied to another tensor: reuse, do not redraw
        if mask is None:
            pos = self.stream_position(input.n, input.F, input.device, dict(kwargs))
            if 'calls_dev' in pos:
                mask = K.dropout_mask_ctr(input.n, input.F, self.p, self._seed, pos['calls_dev'], pos['per_call'], pos['base'],
                                          input.device)
            else:
                mask = K.dropout_mask(input.n, input.F, self.p, self._seed, pos['offset'], input.device)
            self.advance(kwargs.get('device_counters'))
        y = K.dropout_apply(input, mask, self.p)
        if tape is not None:
            tape[self] = {'mask': mask}
        return y

    def backward(self, grad, tape, into, **kwargs):
        K = backend.active()
        mask = tape[self]['mask']
        below = self.input_layer
        if isinstance(grad, PreAct):
            # the consumer's product already applied this layer's mask and the tanh gradient of the layer below in its epilogue
            # (DenseLayer._tanh_layer_below): what comes through is that layer's pre-activation gradient
            assert into[0] is None
            return [grad]
        if (mask is not None and into[0] is None and isinstance(below, DenseLayer) and not isinstance(grad, (PreAct, Masked))
                and below.nonlinearity.act not in (None, 0) and below.nonlinearity.fusable):
            # the layer below applies mask and 1/(1-p) inside its activation-gradient kernel
            return [Masked(grad, mask, 1.0 / (1.0 - self.p))]
        g = grad if mask is None else K.dropout_apply(grad, mask, self.p)
        if into[0] is not None:
            return [_accumulate(into[0], g)]
        return [g]


dropout = DropoutLayer


class NonlinearityLayer(Layer):
    def __init__(self, incoming, nonlinearity=_nl.rectify, name=None, **kwargs):
        super().__init__(incoming, name)
        self.nonlinearity = _nl.resolve(nonlinearity)

    def forward(self, input, tape, **kwargs):
        K = backend.active()
        if self.nonlinearity.act is None:
            raise NotImplementedError("nonlinearity %s has no gfx950 kernel yet" % self.nonlinearity.name)
        y = K.bias_act(input, None, self.nonlinearity.act)
        if tape is not None:
            tape[self] = {'y': y}
        return y

    def backward(self, grad, tape, into, **kwargs):
        K = backend.active()
        g = K.act_bwd(grad, tape[self]['y'], self.nonlinearity.act)
        return [_accumulate(into[0], g) if into[0] is not None else g]


class ElemwiseSumLayer(MergeLayer):
    def __init__(self, incomings, coeffs=1, cropping=None, name=None, **kwargs):
        super().__init__(incomings, name)
        if coeffs != 1 or cropping is not None:
            raise NotImplementedError("only coeffs=1, cropping=None (reference gcnmodel.py:293)")

    def get_output_shape_for(self, input_shapes):
        return input_shapes[0]

    def forward(self, inputs, tape, **kwargs):
        K = backend.active()
        out = inputs[0].like()
        out.t.copy_(inputs[0].t)
        for x in inputs[1:]:
            K.add_inplace(x, out)
        return out

    def backward(self, grad, tape, into, **kwargs):
        outs = []
        for k, b in enumerate(into):
            if b is not None:
                outs.append(_accumulate(b, grad))
            elif k == 0:
                outs.append(grad)
            else:                       # each input needs its own buffer (later accumulations are in place)
                c = grad.like()
                c.t.copy_(grad.t)
                outs.append(c)
        return outs


# --------------------------------------------------------------------------------------------
# graph helpers (lasagne.layers.helper)
# --------------------------------------------------------------------------------------------
def get_all_layers(layer, treat_as_input=None):
    """Topological order, inputs before the layer that consumes them (lasagne helper.py)."""
    try:
        queue = deque(layer)
    except TypeError:
        queue = deque([layer])
    seen, done, result = set(), set(), []
    if treat_as_input is not None:
        seen.update(treat_as_input)
    while queue:
        layer = queue[0]
        if layer is None:
            queue.popleft()
        elif layer not in seen:
            seen.add(layer)
            if hasattr(layer, 'input_layers'):
                queue.extendleft(reversed(layer.input_layers))
            elif hasattr(layer, 'input_layer'):
                queue.appendleft(layer.input_layer)
        else:
            queue.popleft()
            if layer not in done:
                result.append(layer)
                done.add(layer)
    return result


def get_output(layer_or_layers, inputs=None, tape=None, **kwargs):
    """Evaluate the network.  `inputs` maps InputLayers to values (as the reference calls it:
    get_output(l_out, {l_in: X}, A=A, deterministic=...), gcnmodel.py:375,392,400); every kwarg is
    forwarded to every layer's forward / get_output_for, as in Lasagne."""
    all_layers = get_all_layers(layer_or_layers)
    values = {}
    if isinstance(inputs, dict):
        values.update(inputs)
    elif inputs is not None:
        ins = [l for l in all_layers if isinstance(l, InputLayer)]
        if len(ins) != 1:
            raise ValueError("a bare input value needs a network with exactly one InputLayer")
        values[ins[0]] = inputs
    started = set()
    kwargs = dict(kwargs, evaluated_layers=frozenset(all_layers))       # (fusions across layers ask whether their partner takes part)
    for pos, layer in enumerate(all_layers):
        if layer in values:
            continue
        if isinstance(layer, InputLayer):
            if layer.input_var is None:
                raise ValueError("no value for InputLayer %r" % layer.name)
            values[layer] = layer.input_var
            continue
        if hasattr(layer, 'input_layers'):
            x = [values[l] for l in layer.input_layers]
        else:
            x = values[layer.input_layer]
        # partitioned graph: a convolution that comes NEXT in the order and whose input is already there starts
        # its H.W product and the exchange of the result now, so that this layer's work (the highway gate's
        # GEMM: both read the same H) overlaps the collective
        if pos + 1 < len(all_layers):
            nxt = all_layers[pos + 1]
            if (nxt not in started and nxt not in values and hasattr(nxt, 'forward_begin') and nxt.can_split(kwargs)
                    and nxt is not layer and getattr(nxt, 'input_layer', None) in values
                    and getattr(nxt, 'input_layer', None) is not layer):
                nxt.forward_begin(values[nxt.input_layer], tape, **kwargs)
                started.add(nxt)
        values[layer] = layer.forward(x, tape, **kwargs)
    if tape is not None:
        tape['__values__'] = values
    try:
        return [values[l] for l in layer_or_layers]
    except TypeError:
        return values[layer_or_layers]


def backward(layer, grad, tape, **kwargs):
    """Reverse sweep over the layer DAG (what theano.grad did for the reference).  `grad` is the
    gradient at `layer`'s output (or a PreAct token); parameter gradients land in Param.grad."""
    all_layers = get_all_layers(layer)
    grads = {layer: grad}
    # how many inputs of other layers each layer's output is (a fusion that hands a finished gradient through a layer needs to know
    # that nobody else will add to it)
    consumers = {}
    for l in all_layers:
        for i in (l.input_layers if hasattr(l, 'input_layers') else [getattr(l, 'input_layer', None)]):
            if i is not None:
                consumers[i] = consumers.get(i, 0) + 1
    kwargs = dict(kwargs, consumers=consumers)
    # which layers lie on a path from a parameterised/needed layer: all of them need grads except
    # pure inputs
    def dense(g):
        # a highway block's carry gradient that nobody formed in an epilogue (ops.GateCarry): form it now
        return g.dense() if isinstance(g, backend.active().GateCarry) else g

    def run(l, **extra):
        g = dense(grads.pop(l))
        ins = l.input_layers if hasattr(l, 'input_layers') else [l.input_layer]
        into = [grads.get(i) if not isinstance(i, InputLayer) else None for i in ins]
        if not getattr(l, 'takes_gate_carry', False):
            into = [dense(b) for b in into]
        need = [not isinstance(i, InputLayer) for i in ins]
        outs = l.backward(g, tape, into, need_input_grad=any(need), **dict(kwargs, **extra))
        for i, o in zip(ins, outs):
            if o is not None and not isinstance(i, InputLayer):
                grads[i] = o

    rev = list(reversed(all_layers))
    done = set()
    for pos, l in enumerate(rev):
        if isinstance(l, InputLayer) or l not in grads or l in done:
            continue
        # partitioned graph: start the exchange of this convolution's dS, run the NEXT layer of the sweep first if
        # it is independent of this one (the highway gate: its gradient is complete and it is not an ancestor of
        # this layer), then finish -- the gate's two GEMMs overlap the collective.  (Both add into the same dH:
        # the order of the two additions differs from the single-GPU sweep by one fp32 rounding.)
        if hasattr(l, 'backward_begin') and l.can_split(kwargs) and pos + 1 < len(rev):
            nxt = rev[pos + 1]
            if (not isinstance(nxt, InputLayer) and nxt in grads and nxt not in done
                    and nxt not in get_all_layers(l)):
                # exchange of dS  ||  the gate's dW GEMM;   then SpMM;   return exchange  ||  the gate's dH GEMM
                l.backward_begin(grads[l], tape, **kwargs)
                run(nxt, after_dw=lambda l=l: l.backward_mid(tape, **kwargs))
                done.add(nxt)
        run(l)
        done.add(l)
    return grads


def get_all_params(layer, unwrap_shared=True, **tags):
    seen, out = set(), []
    for l in get_all_layers(layer):
        for p in l.get_params(**tags):
            if p not in seen:
                seen.add(p)
                out.append(p)
    return out


def count_params(layer, **tags):
    return int(sum(np.prod(p.shape) for p in get_all_params(layer, **tags)))


def get_all_param_values(layer, **tags):
    return [p.get_value() for p in get_all_params(layer, **tags)]


def set_all_param_values(layer, values, **tags):
    params = get_all_params(layer, **tags)
    if len(params) != len(values):
        raise ValueError("mismatch: got %d values to set %d parameters" % (len(values), len(params)))
    for p, v in zip(params, values):
        p.set_value(v)
