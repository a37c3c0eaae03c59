#!/bin/bash
# round 6: the side-stream CTU-load stage with ticketed groups + a deadline priority boost (k0_tile_side) against round 5's form (groups by
# blockIdx, priority 0 throughout): parity, then C3 / C2 / C4 step times alternating, then the step gaps of a traced run
set -u
mkdir -p gpurun_out
REPO=$PWD
EXP=$REPO/hevc-complexity-reduction_amd/lib_exp/libethcnn.so
OUT=gpurun_out/tile_side_ab.txt
{
python -m pytest tests/test_gpu_parity.py tests/test_gpu_big_passes.py tests/test_gpu_robustness.py tests/test_gpu_switch_points.py -m gpu -x -q --timeout 900 2>&1 | tail -3
run() {  # label, env...
  local label=$1; shift
  for wl in c3 c2 c4; do
    env ETHCNN_LIB=$EXP "$@" python bench.py --workload $wl --no-cpu-baseline --no-host-scopes --no-other-configs --no-fast-plan --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-34s %s: %.2f M CTU/s  %.4f ms/step  fc1 in region %.4f ms' % ('$label', '$wl', d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms']))"
  done
}
for rep in 1 2; do
  run "round 5 form (TILE_SIDE=0)" ETHCNN_TILE_SIDE=0
  run "tickets, no boost" ETHCNN_TILE_BOOST_PCT=0
  run "tickets + boost at 50 %" ETHCNN_TILE_BOOST_PCT=50
  run "tickets + boost at 70 % [default]" ETHCNN_TILE_BOOST_PCT=70
  run "tickets + boost at 85 %" ETHCNN_TILE_BOOST_PCT=85
done
cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  rm -rf $REPO/gpurun_out/prof_ts$v
  ETHCNN_LIB=$EXP ETHCNN_TILE_SIDE=$v rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_ts$v -o t -- python $REPO/bench.py --workload c3 --no-cpu-baseline --no-host-scopes --no-other-configs --no-fast-plan > /dev/null 2>&1
  python $REPO/scripts/trace_gaps.py $(find $REPO/gpurun_out/prof_ts$v -name "*kernel_trace.csv" | head -1) "c3, ETHCNN_TILE_SIDE=$v"
done
} > $OUT 2>&1
cat $OUT
