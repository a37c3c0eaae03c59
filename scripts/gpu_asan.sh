#!/bin/bash
# The host side of the library (worker pool, staging ring, cross-stream events, completion word, checkpoint parser) under
# AddressSanitizer + UndefinedBehaviorSanitizer: `make asan` builds the host .cpp files instrumented with g++ (the kernels are the
# same hipcc objects) into lib_asan/libethcnn.so; the CPU suite and the GPU suite then run on it through ETHCNN_LIB with the sanitizer runtime
# preloaded into python.  Any report aborts the process (halt_on_error, -fno-sanitize-recover): a green run = no finding.
# Usage (GPU box): bash scripts/gpu_asan.sh [pytest args]  -> gpurun_out/asan_gpu.log
set -u
mkdir -p gpurun_out
RT=$(g++ -print-file-name=libasan.so)  # gcc's runtime (Makefile, target asan, says why not ROCm's compiler-rt)
LIB=$PWD/hevc-complexity-reduction_amd/lib_asan/libethcnn.so
[ -f "$LIB" ] || make -s -j8 -C hevc-complexity-reduction_amd/csrc asan
export ETHCNN_LIB=$LIB LD_PRELOAD=$RT
# protect_shadow_gap=0: the HIP runtime maps memory inside ASan's shadow gap; leaks: python's own allocations are not ours
# use_sigaltstack=0: gcc 11 libasan fails to unmap the alternate signal stack of exiting python threads (its own CHECK, not a finding)
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:abort_on_error=1:use_sigaltstack=0
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
{
  echo "# sanitizer runtime: $RT"; echo "# library: $LIB ($(nm -D $LIB | grep -c __asan) asan symbols referenced)"
  python -m pytest tests -q -m "not gpu" -x -p no:cacheprovider 2>&1 | tail -4
  # torch cannot be imported under a preloaded sanitizer runtime (it aborts inside its own static initialisers): the bench-contract
  # tests, which import torch / run bench.py, stay out; everything else of the GPU suite runs
  python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider --deselect tests/test_gpu_bench_contract.py ${@:-} 2>&1 | grep -v '^  File "/usr' | tail -25
} > gpurun_out/asan_gpu.log 2>&1
cat gpurun_out/asan_gpu.log
