#!/usr/bin/env python
"""GPU box: writes gpurun_out/hm/{seq.yuv, cu_depth_gpu.dat, cu_depth_oracle.dat} for
scripts/hm_bitstream_check.py (which runs where the HM binary lives).  cu_depth_gpu.dat is
produced by the real drop-in launcher exactly as HM's hook would run it."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
import ethcnn_np as oracle  # noqa: E402

w, h, frames, qp, seed, gain = 416, 240, 4, 32, 9, 8.0
out = os.path.join(ROOT, "gpurun_out", "hm")
os.makedirs(out, exist_ok=True)
luma = bench.synth_luma(w, h, frames, 1)
yuv = np.concatenate([np.concatenate([luma[f].reshape(-1), np.full(w * h // 2, 128, np.uint8)]) for f in range(frames)])
yuv.tofile(os.path.join(out, "seq.yuv"))
open(os.path.join(out, "Thr_info.txt"), "w").write("0.5 0.5 0.5 0.5 0.5 0.5\n")
env = dict(os.environ, ETHCNN_SYNTHETIC_SEED=str(seed), ETHCNN_HEAD_GAIN=str(gain))
r = subprocess.run([sys.executable, os.path.join(ROOT, "video_to_cu_depth.py"), "seq.yuv", str(w), str(h), str(qp)], cwd=out, env=env)
assert r.returncode == 0
os.replace(os.path.join(out, "cu_depth.dat"), os.path.join(out, "cu_depth_gpu.dat"))
P = oracle.predict_frames(oracle.synth_blob(seed, gain), yuv, w, h, frames, qp, 0.5, 0.5, frame_stride=w * h * 3 // 2)
P.astype("<f4").tofile(os.path.join(out, "cu_depth_oracle.dat"))
print("hm case written:", w, h, frames, qp, "files identical:",
      open(os.path.join(out, "cu_depth_gpu.dat"), "rb").read() == open(os.path.join(out, "cu_depth_oracle.dat"), "rb").read())
