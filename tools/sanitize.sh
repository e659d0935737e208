#!/bin/bash
# Host-side sanitizer pass over the C-ABI ("sanitizers").
#
#   tools/sanitize.sh            # AddressSanitizer + UndefinedBehaviorSanitizer (default)
#   tools/sanitize.sh undefined  # UndefinedBehaviorSanitizer alone
#   tools/sanitize.sh thread     # ThreadSanitizer (host threads: the occupancy cache mutex, the CU cache)
# (ROCm's ASan runtime intercepts the HSA allocator and wants XNACK-capable device memory: on a box that does not grant it
# - `out of memory ... hsa_amd_memory_pool_allocate` at the first device allocation - run the ASan pass where no GPU is
# visible (it then covers tests/test_host_abi.py: symbols, argument validation, error paths) and the `undefined` and
# `thread` passes, whose runtimes leave HSA alone, on the GPU box.)
#
# Builds the library with the sanitizer on the HOST side only (-fno-gpu-sanitize: the device code is the shipped code),
# puts it in the library's place for the duration of the run, and runs
#   * tests/test_host_abi.py       - every exported symbol, argument validation, error paths      (no GPU needed)
#   * tests/test_gpu_threads.py    - two host threads x two streams through the per-ply, fused and env-step entry points
#                                    in a fresh process, i.e. with cold caches                    (only where a GPU is visible)
# with the sanitizer runtime preloaded into Python; the shipped library is restored afterwards.  Exit code 0 = clean.
# Leak checking is off (the Python interpreter and the HIP runtime keep their arenas); everything else is fatal.
set -u
KIND=${1:-address}
R=$(cd "$(dirname "$0")/.." && pwd)
LIB=$R/gymgo_amd/libgymgo_amd.so
CLANG_LIB=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
case $KIND in
  address) FLAGS="-fsanitize=address,undefined -fno-sanitize-recover=undefined"; RT=$CLANG_LIB/libclang_rt.asan-x86_64.so
           # (protect_shadow_gap=0: the GPU driver maps device memory into address ranges ASan would otherwise reserve)
           export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 ;;
  undefined) FLAGS="-fsanitize=undefined -fno-sanitize-recover=undefined"; RT=$CLANG_LIB/libclang_rt.ubsan_standalone-x86_64.so
           export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 ;;
  thread)  FLAGS="-fsanitize=thread"; RT=$CLANG_LIB/libclang_rt.tsan-x86_64.so
           # (the suppressions name the uninstrumented GPU runtime: test_host_abi.py's GYMGO_AMD_CUS test initialises HIP in a
           # child process when a GPU is visible, and libhsa-runtime64's own start-up threads race among themselves)
           export TSAN_OPTIONS=suppressions=$R/tools/sanitize/tsan.supp:halt_on_error=1:report_signal_unsafe=0 ;;
  *) echo "usage: $0 [address|undefined|thread]"; exit 2 ;;
esac
[ -f "$RT" ] || { echo "sanitizer runtime $RT not found"; exit 2; }
TMP=$(mktemp -d)
# (SAN_PREBUILT=<file>: a sanitizer build made earlier with the command below - hipcc cross-compiles without a GPU, so the
# minute it takes need not be spent on the GPU box; SAN_KEEP=<file>: keep this run's build there)
if [ -n "${SAN_PREBUILT:-}" ] && [ -f "$SAN_PREBUILT" ]; then
  echo "[sanitize] using the prebuilt $KIND build $SAN_PREBUILT"
  cp "$SAN_PREBUILT" $TMP/libgymgo_amd.so
else
  echo "[sanitize] building the $KIND build (host side only) ..."
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include $FLAGS -fno-gpu-sanitize -shared \
      -o $TMP/libgymgo_amd.so $R/gymgo_amd/csrc/gg_kernels.hip $R/gymgo_amd/csrc/gg_rollout.hip $R/gymgo_amd/csrc/gg_lat.hip $R/gymgo_amd/csrc/gg_r5.hip 2> $TMP/build.log || { tail -20 $TMP/build.log; exit 1; }
  [ -n "${SAN_KEEP:-}" ] && cp $TMP/libgymgo_amd.so "$SAN_KEEP"
fi
[ -f "$LIB" ] && cp -p "$LIB" $TMP/shipped.so
restore() { if [ -f $TMP/shipped.so ]; then cp -p $TMP/shipped.so "$LIB"; else rm -f "$LIB"; fi; rm -rf $TMP; }
trap restore EXIT
cp $TMP/libgymgo_amd.so "$LIB"
touch "$LIB"      # (tests/conftest.py rebuilds a library older than its sources)
cd $R
rc=0
echo "[sanitize] tests/test_host_abi.py"
LD_PRELOAD=$RT python -m pytest tests/test_host_abi.py -x -q -s -p no:cacheprovider; e=$?; [ $e -eq 0 ] || { echo "[sanitize] exit code $e"; rc=1; }
if [ $KIND = thread ]; then
  # torch's GPU initialisation does not survive the TSan runtime: the two-thread test runs as a torch-free C++ driver
  # (tools/sanitize/abi_threads.cpp: the same mix of entry points, two threads x two streams from a start barrier, results
  # compared with sequential calls); the uninstrumented GPU runtime underneath is suppressed (tools/sanitize/tsan.supp)
  echo "[sanitize] tools/sanitize/abi_threads.cpp (two threads x two streams, no Python in the process)"
  DRV=${SAN_DRIVER:-$TMP/abi_threads}
  if [ ! -x "$DRV" ]; then
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fsanitize=thread -I$R/include $R/tools/sanitize/abi_threads.cpp -L$R/gymgo_amd -l:libgymgo_amd.so \
        -Wl,-rpath,$R/gymgo_amd -o $DRV 2> $TMP/drv.log || { tail -5 $TMP/drv.log; rc=1; }
    [ -n "${SAN_KEEP_DRIVER:-}" ] && cp $DRV "$SAN_KEEP_DRIVER"
  fi
  LD_LIBRARY_PATH=$R/gymgo_amd:${LD_LIBRARY_PATH:-} TSAN_OPTIONS=suppressions=$R/tools/sanitize/tsan.supp:halt_on_error=0:report_signal_unsafe=0 $DRV; e=$?
  [ $e -eq 0 ] || { echo "[sanitize] exit code $e"; rc=1; }
elif python -c 'import torch, sys; sys.exit(0 if torch.cuda.is_available() else 1)' 2>/dev/null; then
  echo "[sanitize] tests/test_gpu_threads.py (fresh process: cold caches)"
  LD_PRELOAD=$RT python -m pytest tests/test_gpu_threads.py -m gpu -x -q -s -p no:cacheprovider; e=$?; [ $e -eq 0 ] || { echo "[sanitize] exit code $e"; rc=1; }
else
  echo "[sanitize] no GPU visible: tests/test_gpu_threads.py skipped (run this script on the GPU box: gpurun -- tools/sanitize.sh)"
fi
[ $rc -eq 0 ] && echo "[sanitize] $KIND: clean" || echo "[sanitize] $KIND: FAILED"
exit $rc
