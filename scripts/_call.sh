set -u
mkdir -p gpurun_out
rm -f gpurun_out/fb_*.json
python -m pytest tests/test_gpu_parity.py tests/test_gpu_robustness.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do for v in 30 -1; do for wl in c3 c2 c4; do
if [ $v = -1 ]; then env -u ETHCNN_FC1_VARIANT python bench.py --workload $wl --no-cpu-baseline --no-host-scopes --steps 30 > gpurun_out/fb_${wl}_fused_r$rep.json 2>gpurun_out/fb.err || tail -3 gpurun_out/fb.err
else ETHCNN_FC1_VARIANT=$v python bench.py --workload $wl --no-cpu-baseline --no-host-scopes --steps 30 > gpurun_out/fb_${wl}_two_r$rep.json 2>gpurun_out/fb.err || tail -3 gpurun_out/fb.err; fi
done; done; done
python scripts/summarize.py "gpurun_out/fb_*.json"
