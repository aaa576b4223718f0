"""Lasagne-like layer API for the GCN hot path (the subset of ``lasagne`` that
/root/reference/gcnmodel.py uses: gcnmodel.py:8-14, 270-275, 290-294, 345-414), re-implemented
over the gfx950 HIP kernels of libgeogcn.so.  Submodules mirror lasagne's: ``init``,
``nonlinearities``, ``layers``."""
from . import init, layers, nonlinearities  # noqa: F401
