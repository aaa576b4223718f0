"""CPU-side checks of the drop-in boundary: libgeogcn.so builds, loads and exports exactly the
symbols include/geogcn.h declares (no compute calls without a GPU)."""
import ctypes
import os
import subprocess

import pytest

from geographconv_amd import _ffi


@pytest.fixture(scope="module")
def built():
    from geographconv_amd import build
    return build.build_library()


def test_header_and_binding_agree():
    assert set(_ffi.header_symbols()) == set(_ffi.SIGNATURES)


def test_library_exports_every_declared_symbol(built):
    out = subprocess.run(['nm', '-D', '--defined-only', built], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if ' T ' in line}
    missing = set(_ffi.header_symbols()) - exported
    assert not missing, missing
    extra = {s for s in exported if s.startswith('geogcn_')} - set(_ffi.header_symbols())
    assert not extra, extra


def header_abi_version():
    import re
    with open(_ffi.HEADER_PATH) as f:
        return int(re.search(r'^#define GEOGCN_ABI_VERSION (\d+)', f.read(), flags=re.M).group(1))


def test_library_loads_and_reports_version(built):
    lib = _ffi.lib()
    assert lib.geogcn_version() == _ffi.ABI_VERSION == header_abi_version()
    assert isinstance(lib.geogcn_last_error(), bytes)


def test_argument_errors_need_no_gpu(built):
    lib = _ffi.lib()
    # negative size is rejected before any HIP call
    rc = lib.geogcn_gemm_f32(0, 0, -1, 4, 4, None, 4, None, 4, None, 4, None, 0, 0, 0, None, 0, None)
    assert rc == -2 and b'negative' in lib.geogcn_last_error()
    rc = lib.geogcn_spmm_csr_f32(None, 4, 4, 1, None, None, None, None, 4, None, 4, 4, None, 0, None, 0, None)
    assert rc == -1
    assert lib.geogcn_gemm_workspace_bytes(0, 0, 1000, 300, 300, 0) == 0
    assert lib.geogcn_gemm_workspace_bytes(1, 0, 300, 300, 440000, 0) > 0
    assert lib.geogcn_colsum_workspace_bytes(440000, 300) > 0


def test_the_librarys_test_seams_are_read_per_call(built, monkeypatch):
    """csrc/common.h test_seam_i64: GEOGCN_X3_ROWS_MIN_M moves the row threshold of the split-bf16 whole-rows kernel at the NEXT call
    (host-side dispatch only: a workspace query needs no GPU), lifts its padded-column rule, and garbage / non-positive values fall back
    to the default; no debug hook is exported any more."""
    lib = _ffi.lib()
    monkeypatch.delenv('GEOGCN_X3_ROWS_MIN_M', raising=False)

    def takes(M, N, K):          # (the whole-rows kernel's fragment-ordered weights are larger than the staged kernel's three planes)
        return lib.geogcn_gemm_workspace_bytes(0, 0, M, N, K, _ffi.GEMM_BF16X3) > 3 * ((K + 31) // 32 * 32) * N * 2
    assert takes(4096, 300, 300) and not takes(4095, 300, 300) and takes(9475, 300, 300)          # (CMU size: taken since round 6)
    assert takes(440000, 900, 900) and takes(440000, 1024, 1024) and not takes(440000, 1025, 300)
    assert not takes(440000, 129, 300)                                # half a column pass of padding: not taken by default
    monkeypatch.setenv('GEOGCN_X3_ROWS_MIN_M', '1')
    assert takes(700, 300, 300) and takes(96, 12, 12) and takes(440000, 129, 300) and not takes(96, 1025, 12)
    for bad in ('0', '-5', 'abc', '12x', ''):
        monkeypatch.setenv('GEOGCN_X3_ROWS_MIN_M', bad)
        assert not takes(700, 300, 300) and takes(4096, 300, 300), bad
    monkeypatch.delenv('GEOGCN_X3_ROWS_MIN_M')
    assert not takes(700, 300, 300)
    assert not any('debug' in name for name in _ffi.SIGNATURES) and not hasattr(lib, 'geogcn_debug_set_tn_slab_limit')


def test_missing_gpu_fails_loudly():
    import torch
    from geographconv_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_ffi.GeoGcnError):
        ops.require_gpu()


def test_no_product_import_of_oracle():
    """The product package must not import the oracle (it is the checker, not a fallback)."""
    root = os.path.join(os.path.dirname(_ffi.__file__))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


def test_every_entry_point_rejects_null_and_negative_arguments(built):
    """Error behaviour at the boundary (include/geogcn.h conventions): with every pointer NULL and positive sizes, and
    with negative sizes, every int-returning entry point answers < 0 with a message -- before any HIP call, so this runs
    without a GPU -- and never crashes."""
    import ctypes as C
    lib = _ffi.lib()
    skip = {'geogcn_version', 'geogcn_comm_available', 'geogcn_comm_world', 'geogcn_comm_rank', 'geogcn_spmm_hot_capacity'}
    checked = 0
    for name, (res, args) in _ffi.SIGNATURES.items():
        if res is not _ffi.c_i32 or name in skip:
            continue
        for mode in ('null', 'negative'):
            vals = []
            for t in args:
                if t is _ffi.c_ptr or (isinstance(t, type) and issubclass(t, C._Pointer)):
                    vals.append(None)
                elif t in (_ffi.c_i32, _ffi.c_i64, _ffi.c_sz, _ffi.c_u64):
                    vals.append(-3 if (mode == 'negative' and t is not _ffi.c_sz) else 8)
                elif t is _ffi.c_f32:
                    vals.append(0.5)
                else:
                    vals.append(None)
            rc = getattr(lib, name)(*vals)
            assert rc < 0 and lib.geogcn_last_error(), (name, mode, rc)
            checked += 1
    assert checked >= 80
