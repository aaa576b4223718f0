"""Nonlinearity tokens (lasagne.nonlinearities names used at gcnmodel.py:286,290,345-348,374).

In Lasagne these are Theano expression builders; here each is a small object naming the fused
epilogue / kernel that implements it on the device (``act`` = GEOGCN_ACT_* code of include/geogcn.h,
or None when it needs its own kernel, i.e. softmax)."""
from __future__ import annotations


class Nonlinearity:
    def __init__(self, name, act, fusable=True):
        # act: GEOGCN_ACT_* code (None: needs its own kernel); fusable: available as a GEMM / SpMM epilogue
        self.name, self.act, self.fusable = name, act, fusable

    def __repr__(self):
        return "<nonlinearity %s>" % self.name


linear = Nonlinearity('linear', 0)
identity = linear
tanh = Nonlinearity('tanh', 1)
sigmoid = Nonlinearity('sigmoid', 2)
softmax = Nonlinearity('softmax', None)        # row softmax: geogcn_softmax_rows_f32
rectify = Nonlinearity('rectify', 4, fusable=False)   # commented out in the reference (gcnmodel.py:345)
selu = Nonlinearity('selu', 3, fusable=False)         # only in the unused residual_dense (gcnmodel.py:290)


def resolve(nl):
    """None means linear, as in lasagne's DenseLayer."""
    if nl is None:
        return linear
    if not isinstance(nl, Nonlinearity):
        raise TypeError("nonlinearity must come from geographconv_amd.nn.nonlinearities, got %r" % (nl,))
    return nl
