set -u
python -m pytest tests/test_gpu_robustness.py -x -q -m gpu 2>&1 | tail -5
python scripts/fuzz_many.py 2>&1 | tail -3
python scripts/soak.py 2>&1 | tail -3
