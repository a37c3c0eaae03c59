set -u
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-host-scopes --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['roofline']['frac'])"
ETHCNN_OVERLAP=0 python bench.py --no-cpu-baseline --no-host-scopes --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c3 serial', d['value'], d['roofline']['frac'])"
python scripts/fuzz_many.py 2>&1 | tail -1
