"""Kernel backend seam.

The product has exactly ONE backend: ``geographconv_amd.ops`` (libgeogcn.so, gfx950).  It is
selected by default and raises if the library or the GPU is missing.  The seam exists so that the
multi-rank *communication logic* (row partition, all-gather placement, gradient all-reduce) can be
exercised on CPU with ``gloo``: ``tests/`` installs a NumPy test double with ``use()`` for the
duration of a test.  Nothing inside this package ever installs anything but ``ops``."""
from __future__ import annotations

_active = None


def active():
    global _active
    if _active is None:
        from . import ops
        _active = ops
    return _active


def use(module):
    """Install a kernel backend (tests only).  Pass None to restore the HIP backend."""
    global _active
    _active = module
