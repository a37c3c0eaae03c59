#!/bin/bash
# HBM-side traffic per kernel AND grid size, and of a whole step per plan: separate rocprofv3
# --pmc FETCH_SIZE / WRITE_SIZE passes (kernel trace only) of a short bench run of ONE workload (--no-other-configs: no other config's
# grids in the trace) that visits the exact plan and plans 2 / 3 -> scripts/pmc_traffic.py -> gpurun_out/traffic_by_kernel_grid_<wl>.csv
# + gpurun_out/step_traffic.json (committed as profiles/step_traffic.json; bench.py prints it as roofline.traffic and `hbm`).
#   WLS="c3 c2" (default)   workloads to take
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
rm -f gpurun_out/step_traffic.json
for wl in ${WLS:-c3 c2}; do
  cd /tmp
  for pmc in FETCH_SIZE WRITE_SIZE; do
    rm -rf $REPO/gpurun_out/steptraffic_${wl}_$pmc
    rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $REPO/gpurun_out/steptraffic_${wl}_$pmc -o p -- python $REPO/bench.py --workload $wl --no-cpu-baseline --no-host-scopes --no-other-configs --fast-plans 2,3 --steps 3 --warmup 1 --ramp-ms 30 > $REPO/gpurun_out/steptraffic_${wl}_$pmc.log 2>&1 || tail -3 $REPO/gpurun_out/steptraffic_${wl}_$pmc.log
  done
  cd $REPO
  python scripts/pmc_traffic.py collect $wl gpurun_out/steptraffic_${wl}_FETCH_SIZE gpurun_out/steptraffic_${wl}_WRITE_SIZE | cut -c1-400
done
