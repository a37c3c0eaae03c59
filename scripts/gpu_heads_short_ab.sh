#!/bin/bash
# round 6, VERDICT r05 item 7a: the short heads' fetch plan (head_pass_short: h1 of all chunks at block start, four W2 stages, W3 + scalars in one
# round trip) against round 5's (lib_oldheads: scripts/build_variant.sh oldheads -DHEADS_OLD_SHORT): parity, stage time, step time, timeline
set -u
mkdir -p gpurun_out
REPO=$PWD
OLD=$REPO/hevc-complexity-reduction_amd/lib_oldheads/libethcnn.so
[ -s $OLD ] || bash scripts/build_variant.sh oldheads -DHEADS_OLD_SHORT
OUT=gpurun_out/heads_short_ab.txt
{
echo "== parity at the new form"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_big_passes.py tests/test_gates_golden.py tests/test_gpu_robustness.py -m gpu -x -q --timeout 900 2>&1 | tail -3
for rep in 1 2 3; do
  for v in new old; do
    L=$REPO/hevc-complexity-reduction_amd/lib/libethcnn.so; [ $v = old ] && L=$OLD
    ETHCNN_LIB=$L python bench.py --no-cpu-baseline --no-host-scopes --no-other-configs --no-fast-plan --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v run $rep: %.2f M CTU/s  %.4f ms/step  stages alone %s' % (d['value']/1e6, d['ms_per_step'], {k: round(v,4) for k,v in d['stages_ms_per_step'].items()}))"
  done
done
echo "== device timeline of the heads launch, new form (scripts/ubench/heads_probe.hip)"
./scripts/ubench/heads_probe 102000 2>&1 | head -40
echo "== the same, round 5's short heads"
./scripts/ubench/heads_probe_old 102000 2>&1 | head -12
} > $OUT 2>&1
cat $OUT
