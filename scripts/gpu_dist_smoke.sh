#!/bin/bash
# Exercises bench.py's multi-rank control flow (self-launch, barriers, max-over-ranks reduction, sharded host scopes,
# rank-0 JSON) on a ONE-GPU box through the PLAIN command: two ranks share GPU 0 over gloo (RCCL refuses two ranks on
# one device).  The numbers are meaningless; what is checked is that the launch contract works end to end
# (tests/test_gpu_bench_contract.py asserts the same).
set -u
mkdir -p gpurun_out
BENCH_FORCE_DEVICE=0 BENCH_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-seconds 4 > gpurun_out/dist_smoke.json 2> gpurun_out/dist_smoke.err
echo "exit $?"; tail -2 gpurun_out/dist_smoke.err
python bench.py --gpus 2 > /dev/null 2> gpurun_out/dist_refuse.err; echo "plain --gpus 2 on this box: exit $? ($(tail -1 gpurun_out/dist_refuse.err))"
python -c "
import json
lines=[l for l in open('gpurun_out/dist_smoke.json') if l.startswith('{')]
assert len(lines)==1, lines
d=json.loads(lines[0]); print('n_gpus', d['n_gpus'], 'value %.2fM' % (d['value']/1e6), 'ms_per_step %.3f' % d['ms_per_step'], 'keys ok', all(k in d for k in ('metric','unit','steps','warmup','scaling','roofline','config','cpu_baseline','host_scopes')))
print('host_scopes', d['host_scopes']); assert 'wall_ms' in d['host_scopes']['sharded_command'], d['host_scopes']['sharded_command']"
