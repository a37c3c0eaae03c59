#!/bin/bash
# the hand-offs inside a launch (single-launch small pass, k_lstm_frame, gate arrival counters) with LLVM's FULL gfx942
# agent-scope release / acquire sequences (buffer_wbl2 sc1 ... / buffer_inv sc1: lib_fence, scripts/build_variant.sh fence -DETHCNN_FULL_FENCE)
# against the shipped lean form (agent-scope accesses + s_waitcnt vmcnt(0), no cache maintenance): correctness of the fenced build on the
# tests of those paths, then latency A/B/A/B at 96 / 510 / 2040 CTUs (device-resident pictures) and per LDP frame.
set -u
mkdir -p gpurun_out
REPO=$PWD
FENCE=$REPO/hevc-complexity-reduction_amd/lib_fence/libethcnn.so
[ -s $FENCE ] || bash scripts/build_variant.sh fence -DETHCNN_FULL_FENCE
OUT=gpurun_out/handoff_fence_ab.txt
{
echo "# scripts/gpu_fence_ab.sh: lean = shipped library; fence = the same sources with -DETHCNN_FULL_FENCE (buffer_wbl2 sc1 before every announcing"
echo "# atomic's s_waitcnt, buffer_inv sc1 behind every consumer's flag read / completer's ticket)"
echo "== parity of the FENCED build (tests of the single-launch pass, the LSTM frame kernel, the gates):"
ETHCNN_LIB=$FENCE python -m pytest tests/test_gpu_small.py tests/test_gpu_lstm.py tests/test_gates_golden.py tests/test_gpu_switch_points.py -m gpu -x -q --timeout 900 2>&1 | tail -3
for rep in 1 2 3; do
  for v in lean fence; do
    L=""; [ $v = fence ] && L=$FENCE
    echo "== $v, run $rep: device-resident pictures (scripts/latency.py, 200 calls each)"
    ETHCNN_LIB=${L:-$REPO/hevc-complexity-reduction_amd/lib/libethcnn.so} python scripts/latency.py 2>&1 | grep -E "predict|resi" | grep -v 4928
    echo "== $v, run $rep: LDP frames (scripts/latency_ldp.py)"
    ETHCNN_LIB=${L:-$REPO/hevc-complexity-reduction_amd/lib/libethcnn.so} python scripts/latency_ldp.py 2>&1 | grep -E "x[0-9]+ " | head -8
  done
done
} > $OUT 2>&1
tail -60 $OUT
