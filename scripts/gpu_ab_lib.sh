#!/bin/bash
# A/B of library builds: LIBS="default prio m0" WLS="c2 c3"
set -u
mkdir -p gpurun_out
P=$PWD/hevc-complexity-reduction_amd
for wl in ${WLS:-c3}; do for l in ${LIBS:-default}; do
  if [ $l = default ]; then lib=$P/lib/libethcnn.so; else lib=$P/lib_$l/libethcnn.so; fi
  for rep in 1 2; do
  ETHCNN_LIB=$lib python bench.py --workload $wl --no-cpu-baseline --no-host-scopes --steps ${STEPS:-20} > gpurun_out/abl_${wl}_${l}_$rep.json 2>gpurun_out/abl.err || tail -3 gpurun_out/abl.err
  done
done; done
python scripts/summarize.py "gpurun_out/abl_*.json"
