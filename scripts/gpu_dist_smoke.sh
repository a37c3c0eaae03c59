#!/bin/bash
# Exercises bench.py's multi-rank control flow (barriers, max-over-ranks reduction, rank-0 JSON) on a ONE-GPU
# box: two torchrun ranks share GPU 0 over gloo.  The numbers are meaningless (two ranks share one GPU);
# what is checked is that the launch contract works end to end.
set -u
BENCH_FORCE_DEVICE=0 BENCH_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/dist_smoke.json 2> gpurun_out/dist_smoke.err
echo "exit $?"; tail -2 gpurun_out/dist_smoke.err
python -c "
import json
lines=[l for l in open('gpurun_out/dist_smoke.json') if l.startswith('{')]
assert len(lines)==1, lines
d=json.loads(lines[0]); print('n_gpus', d['n_gpus'], 'value %.2fM' % (d['value']/1e6), 'ms_per_step %.3f' % d['ms_per_step'], 'keys ok', all(k in d for k in ('metric','unit','steps','warmup','scaling','roofline','config')))"
