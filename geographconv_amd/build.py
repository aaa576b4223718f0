"""Build libgeogcn.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m geographconv_amd.build [--force]

The .so stays next to this file (git-ignored, but it travels to the GPU box with gpurun).  hipcc
cross-compiles for gfx950 without a GPU, so this is also the driver's "does it build" check
(__graft_entry__.build)."""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'libgeogcn.so')
SOURCES = ['core.hip', 'spmm.hip', 'spmm_hot.hip', 'xt.hip', 'gemm.hip', 'gemm_x3.hip', 'gemm_bf16.hip', 'elementwise.hip', 'softmax_adam.hip', 'comm.hip']
# (no -munsafe-fp-atomics: the two float atomics of the library -- softmax_adam.hip, exact by construction -- ask for the
#  hardware add themselves with unsafeAtomicAdd; any future atomicAdd gets the safe default)
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
# ablation / experiment builds: GEOGCN_BUILD_DEFINES="GEOGCN_BF16_PROBE_BUILD ..." python -m geographconv_amd.build --force
FLAGS += ['-D' + d for d in os.environ.get('GEOGCN_BUILD_DEFINES', '').split()]


def _hipcc():
    return shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(os.path.dirname(HERE), 'include', 'geogcn.h'))
    return hdrs


def _stale(target, srcs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def _compile(src):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + '.o')
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + _deps()):
        cmd = [_hipcc()] + FLAGS + ['-c', path, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
        if os.path.exists(LIB):
            os.remove(LIB)
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if _stale(LIB, objs):
        cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build_library(force='--force' in sys.argv, verbose=True)
