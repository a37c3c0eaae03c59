"""CPU: the TF-CPU proxy (oracle/tf_cpu_proxy.py, the "B2" CPU baseline of bench.py) computes the real
thing: its cu_depth.dat agrees with the C oracle within the north star's 1e-4, gates included."""
import numpy as np


def test_proxy_matches_oracle(oracle, tmp_path):
    import bench
    import tf_cpu_proxy as proxy
    w, h, frames, qp = 200, 136, 2, 32
    luma = bench.synth_luma(w, h, frames, seed=5)
    yuv = tmp_path / "s.yuv"
    with open(str(yuv), "wb") as f:
        for k in range(frames):
            f.write(luma[k].tobytes())
            f.write(bytes([128]) * (w * h // 2))
    for gain, thr in ((8.0, (0.5, 0.5)), (1.0, (0.999, 0.5))):
        blob = oracle.synth_blob(4, gain)
        done, ctus, _ = proxy.predict_file(blob, str(yuv), w, h, qp, str(tmp_path / "p.dat"), thr[0], thr[1], threads=2)
        assert (done, ctus) == (frames, frames * 12)
        got = np.fromfile(str(tmp_path / "p.dat"), dtype="<f4").reshape(-1, 21)
        want = oracle.predict_frames(blob, luma, w, h, frames, qp, thr[0], thr[1], mode=0)
        assert np.abs(got - want).max() <= 1e-4
        assert np.array_equal(got == 0.0, want == 0.0)  # same gate outcome
    # bounded sample: stops after max_frames
    done, ctus, _ = proxy.predict_file(blob, str(yuv), w, h, qp, str(tmp_path / "p.dat"), max_frames=1, threads=2)
    assert (done, ctus) == (1, 12) and (tmp_path / "p.dat").stat().st_size == 12 * 84
