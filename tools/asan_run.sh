#!/bin/bash
# Host-side AddressSanitizer run of the C-ABI library (aux row 2): libgeogcn.so rebuilt with
# -fsanitize=address on the HOST code (device code unchanged: -fno-gpu-sanitize), the sanitizer runtime preloaded into python,
# then the randomised entry-point sweeps (tests/test_fuzz_gpu.py: 75 cases over every entry point and whole training steps), the kernel tests and
# the ABI argument sweep run against it.  Catches host heap errors in plan builders, workspace arithmetic, argument checks.
#   build here (no GPU needed):   bash tools/asan_run.sh build
#   run on the GPU box:           gpurun -- 'bash tools/asan_run.sh run'      -> gpurun_out/asan_fuzz.log
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
# The sanitizer RUNTIME is GCC's libasan (same ASan ABI as the clang instrumentation): ROCm's own libclang_rt.asan intercepts
# hsa_amd_memory_pool_allocate for device-side ASan and aborts under a stock (non-xnack+) HIP runtime.
RT=$(ls /usr/lib/x86_64-linux-gnu/libasan.so.? | head -1)
cd $ROOT
if [ "${1:-}" = build ]; then
  mkdir -p /tmp/asan_obj
  for f in core spmm spmm_hot xt gemm gemm_x3 gemm_bf16 elementwise softmax_adam comm; do
    hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-gpu-sanitize \
      -fno-omit-frame-pointer -Wno-unused-function -c geographconv_amd/csrc/$f.hip -o /tmp/asan_obj/$f.o || exit 1 &
  done
  wait
  # (linked WITHOUT a sanitizer runtime: the __asan_* symbols resolve against the preloaded libasan)
  hipcc --offload-arch=gfx950 -shared -fPIC -o geographconv_amd/libgeogcn_asan.so /tmp/asan_obj/*.o || exit 1
  ls -la geographconv_amd/libgeogcn_asan.so
  exit 0
fi
mkdir -p gpurun_out
cp geographconv_amd/libgeogcn.so /tmp/libgeogcn_plain.so
cp geographconv_amd/libgeogcn_asan.so geographconv_amd/libgeogcn.so
export LD_PRELOAD="$RT $(ls /usr/lib/x86_64-linux-gnu/libstdc++.so.6)"      # (libstdc++ too: libasan must find __cxa_throw at start-up)
# (libasan's dlopen interceptor becomes the caller of torch's lazy dlopen()s: their $ORIGIN run-paths no longer apply)
export LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib:${LD_LIBRARY_PATH:-}
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=0:detect_odr_violation=0
{ echo "# host ASAN run: $(date -u) runtime $RT"; python -m pytest tests/test_fuzz_gpu.py tests/test_abi.py tests/test_kernels_gpu.py tests/test_x3_gpu.py -q -m "gpu or not gpu" -p no:cacheprovider 2>&1 | tail -25; } > gpurun_out/asan_fuzz.log 2>&1
unset LD_PRELOAD
cp /tmp/libgeogcn_plain.so geographconv_amd/libgeogcn.so
grep -c "ERROR: AddressSanitizer" gpurun_out/asan_fuzz.log
tail -6 gpurun_out/asan_fuzz.log
