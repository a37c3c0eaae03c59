"""GPU: the builder-run claims under the driver's eyes, sized to seconds:
  * a 300-case slice of scripts/fuzz_many.py (random geometries / pitches / strides / thresholds / QPs; AI + resi + LDP);
  * a 400-step slice of scripts/ldp_stress.py (ETH-LSTM steps with changing frame sizes and thresholds: the gate ticket);
  * the FULL C4 job geometry -- 4928x3264 x 425 frames = 1,668,975 CTUs -- streamed from an in-memory generator (no 10 GB
    file): 8 frame-range shards vs 5 differently-cut shards byte-identical, sampled frames bit-exact vs the oracle."""
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _script(name, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", name)], capture_output=True, text=True, env=e, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    return r.stdout


def test_fuzz_slice():
    out = _script("fuzz_many.py", CASES=300, SEED=20260928)
    assert "300 cases, 0 mismatches" in out, out[-500:]


def test_ldp_stress_slice():
    out = _script("ldp_stress.py", STEPS=400, SEED=5)
    assert "400 steps" in out and " 0 mismatches" in out, out[-500:]
    closed = int(out.split("steps,")[1].split("with a closed gate")[0])
    assert closed > 50  # the ticket / zero-fill path really ran


class C4Frames(object):
    """frame k of the synthetic C4 sequence, regenerable on demand: 6 seeded base frames, shifted by a per-frame offset"""
    W, H, N = 4928, 3264, 425

    def __init__(self):
        rng = np.random.default_rng(0xE7C00004)
        yy, xx = np.arange(self.H, dtype=np.int32)[:, None], np.arange(self.W, dtype=np.int32)[None, :]
        kind = ((yy // 256) + (xx // 256)) % 4
        self.base = []
        for b in range(6):
            noise = rng.integers(0, 256, size=(self.H, self.W), dtype=np.uint8)
            grad = ((yy * (2 + b) + xx * 3) // 8) % 256
            flat = ((yy // 16) * 31 + (xx // 16) * (17 + b)) % 200 + 20
            self.base.append(np.where(kind == 0, grad, np.where(kind == 1, noise // 4 + 96, np.where(kind == 2, flat, noise))).astype(np.uint8))

    def frame(self, k):
        return self.base[k % 6] + np.uint8((k * 37) % 256)  # uint8 wrap-around

    def frames(self, f0, f1):
        out = np.empty((f1 - f0, self.H, self.W), dtype=np.uint8)
        for k in range(f0, f1):
            np.add(self.base[k % 6], np.uint8((k * 37) % 256), out=out[k - f0])
        return out


def test_c4_full_job_streamed_in_shards(pkg, oracle):
    seq = C4Frames()
    nctu = pkg.ethcnn.ctus_per_frame(seq.W, seq.H)
    assert nctu == 3927 and nctu * seq.N == 1668975
    blob = oracle.synth_blob(1, 8.0)
    c = pkg.EthCnn(device=0)
    c.load_blob(blob)
    c.set_thresholds(0.5, 0.5)
    try:
        def run(world):
            crc, keep = 0, {}
            for g in range(world):
                f0, f1 = pkg.sharding.frame_range(seq.N, world, g)
                probs = c.predict_luma(seq.frames(f0, f1), seq.W, seq.H, f1 - f0, 27)
                assert probs.shape == ((f1 - f0) * nctu, 21)
                crc = zlib.crc32(probs.tobytes(), crc)
                for k in (0, 53, 54, 211, 371, 424):  # frames on both sides of shard boundaries + the ends
                    if f0 <= k < f1:
                        keep[k] = probs[(k - f0) * nctu:(k - f0 + 1) * nctu].copy()
            return crc, keep
        crc8, keep8 = run(8)     # the job's own partition: 53-54 frames per GPU
        crc5, keep5 = run(5)     # other cut points, other pass sizes
        assert crc8 == crc5      # 140,193,900 bytes of cu_depth.dat payload, independent of the sharding
        for k, got in keep8.items():
            assert np.array_equal(got.view(np.uint32), keep5[k].view(np.uint32))
            want = oracle.predict_frames(blob, seq.frame(k), seq.W, seq.H, 1, 27, 0.5, 0.5, mode=0)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), k
    finally:
        c.close()


def test_handoff_stress_slice():
    """a 6000-launch slice of scripts/handoff_stress.py (the round's full run: 100,000 launches, profiles/r06_handoff_stress.txt): the
    single-launch pass and the LDP frame launches on ONE workspace with different data every launch, bit-compared, beside a context
    streaming C3-sized passes and a thread of copies on the same GPU -- the test of the lean hand-off form (include csrc/ethcnn_kernels.h:
    agent-scope accesses + s_waitcnt, without the two cache-maintenance instructions the full release / acquire sequence would add
    at 20 us per 1080p call)"""
    out = _script("handoff_stress.py", LAUNCHES=6000)
    assert "handoff stress:" in out and out.strip().endswith(": 0 mismatches"), out[-800:]
