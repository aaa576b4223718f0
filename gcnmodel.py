"""Top-level shim keeping the reference's module name: `from gcnmodel import GraphConv` works as it
does against /root/reference/gcnmodel.py, but resolves to the MI355X implementation."""
from geographconv_amd.gcnmodel import *  # noqa: F401,F403
from geographconv_amd.gcnmodel import (ConvolutionDenseLayer2, ConvolutionDenseLayer3,  # noqa: F401
                                       ConvolutionDenseLayer_zero, ConvolutionLayer, DenseLayer2, GraphConv,
                                       MultiplicativeGatingLayer, SparseConvolutionDenseLayer,
                                       SparseConvolutionDenseLayer2, SparseInputDenseLayer, highway_dense,
                                       np_softmax, residual_dense)
