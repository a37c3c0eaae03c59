set -u
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 20 > gpurun_out/b_c3.json 2>gpurun_out/b.err || tail -3 gpurun_out/b.err
python -c "
import json; d=json.load(open('gpurun_out/b_c3.json')); print('c3', d['value'], d['host_scopes'])"
python bench.py --workload c2 --no-cpu-baseline --steps 20 > gpurun_out/b_c2.json 2>gpurun_out/b.err || tail -3 gpurun_out/b.err
python -c "
import json; d=json.load(open('gpurun_out/b_c2.json')); print('c2', d['value'], d['host_scopes'])"
for t in 8 16 32; do ETHCNN_HOST_THREADS=$t python bench.py --no-cpu-baseline --steps 5 > gpurun_out/b_t$t.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/b_t$t.json')); print('threads $t', d['host_scopes'])"; done
