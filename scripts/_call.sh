set -u
mkdir -p gpurun_out
rm -f gpurun_out/ho_*.json
for rep in 1 2; do for o in 0 1; do for wl in c3 c2; do
ETHCNN_HEADS_ORDER=$o python bench.py --workload $wl --no-cpu-baseline --no-host-scopes --steps 30 > gpurun_out/ho_${wl}_o${o}_r${rep}.json 2>gpurun_out/ho.err || tail -3 gpurun_out/ho.err
done; done; done
python scripts/summarize.py "gpurun_out/ho_*.json"
ETHCNN_HEADS_ORDER=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stages or frames or gates" 2>&1 | tail -2
