"""-m gpu: the label / accuracy tooling (tools/score_cu_depth.py) over a cu_depth.dat made on the
GPU: a label file in the reference's `Info_*_CUDepth.dat` format (one byte per 16x16 block) + the predictor's file through
the command line of the tool.  The trained ETH-CNN weights are not in the reference repository, so the accuracy itself
means nothing here (seeded weights); what is checked is the whole chain: GPU file == oracle file, hence identical
confusion matrices, and the matrices count exactly the CTUs / 32x32 / 16x16 blocks the hierarchy defines."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "score_cu_depth.py")


def _labels_from_texture(luma):
    """a plausible ground truth: deeper splits where the 16x16 block is busier (quantised local range)"""
    f, h, w = luma.shape
    b = luma.reshape(f, h // 16, 16, w // 16, 16).astype(np.int32)
    rng_ = b.max(axis=(2, 4)) - b.min(axis=(2, 4))
    return np.digitize(rng_, [24, 96, 200]).astype(np.uint8)  # 0..3


def _run_tool(labels, dat, w, h):
    r = subprocess.run([sys.executable, TOOL, labels, dat, str(w), str(h), "0.5", "0.5", "0.5"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return [[int(x) for x in re.findall(r"\d+", line.split("=")[1].split("accuracy")[0])] for line in r.stdout.strip().splitlines()], r.stdout


def test_score_tool_over_a_gpu_made_file(pkg, oracle, tmp_path):
    import bench
    w, h, frames, qp = 768, 512, 3, 32
    luma = bench.synth_luma(w, h, frames, seed=21)
    yuv = str(tmp_path / "seq.yuv")
    with open(yuv, "wb") as f:
        for k in range(frames):
            f.write(luma[k].tobytes())
            f.write(bytes([128]) * (w * h // 2))
    labels = str(tmp_path / "Info_test_768x512_qp32_nf3_CUDepth.dat")
    lab = _labels_from_texture(luma)
    lab.tofile(labels)
    blob = oracle.synth_blob(1, 8.0)
    c = pkg.EthCnn(device=0)
    c.load_blob(blob)
    c.set_thresholds(0.5, 0.5)
    gpu_dat = str(tmp_path / "cu_depth.dat")
    assert c.predict_yuv_file(yuv, w, h, qp, gpu_dat) == frames
    c.close()
    cpu_dat = str(tmp_path / "cu_depth_oracle.dat")
    oracle.predict_frames(blob, luma, w, h, frames, qp, 0.5, 0.5, mode=0).tofile(cpu_dat)
    assert open(gpu_dat, "rb").read() == open(cpu_dat, "rb").read()
    (m64, m32, m16), text = _run_tool(labels, gpu_dat, w, h)
    assert _run_tool(labels, cpu_dat, w, h)[0] == [m64, m32, m16]
    nctu = frames * 12 * 8
    assert sum(m64) == nctu                                     # every CTU is scored at level 64
    truly64 = m64[2] + m64[3]
    assert sum(m32) == 4 * truly64                              # 32x32 blocks of the truly split CTUs only
    assert sum(m16) == 4 * (m32[2] + m32[3])                    # 16x16 blocks of the truly split 32x32 only
    assert truly64 > 0 and (m32[2] + m32[3]) > 0, "the synthetic labels must exercise all three levels"
    print(text)
