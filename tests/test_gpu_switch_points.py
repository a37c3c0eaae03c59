"""-m gpu: the HOST entry (ethcnn_predict_luma: host pointers in, host probabilities out) exactly AT the sizes where the library changes
its plan (ADVICE r03 #1 / VERDICT r04 item 5), bit-exact against the oracle, pass pipeline on:

  total CTUs <= 2304 (kSmallPassMaxCtus, ethcnn_kernels.h)     one launch for the whole picture (ethcnn_small.hip)
  2305 .. 8191                                                  the latency path with five launches on the main stream
  >= 8192 (kPipelineMinCtus, ethcnn_ctx.h)                      the staging ring + pass pipeline (tile stage on the side stream beside FC1)

A pass of EXACTLY 8192 CTUs once took the latency path while its tile stage ran on the side stream, unordered with the latency path's
H2D copy (fixed in round 3 by the strict `<` in ethcnn_predict_luma); these are the sizes that would show such a slip again.
Reference scope: the <= 1024-CTU sub-batching inside a frame (video_to_cu_depth.py:61-73) must not depend on any of this.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (CTUs per row, CTU rows, frames): total = the switch points and their neighbours, as one picture and as several
CASES = [
    (48, 48, 1),     # 2304: the largest single-launch picture
    (48, 24, 2),     # 2304 as two pictures
    (461, 5, 1),     # 2305: first five-launch size
    (461, 1, 5),     # 2305 as five one-row pictures
    (8191, 1, 1),    # 8191: last latency-path size (a prime: one row of CTUs)
    (128, 64, 1),    # 8192: first ring / pipeline size, one picture
    (16, 16, 32),    # 8192 as 32 pictures of 1024x1024
    (8, 8, 128),     # 8192 as 128 pictures of 512x512
    (2731, 3, 1),    # 8193
    (2731, 1, 3),    # 8193 as three pictures
]


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("pinned", [False, True], ids=["pageable", "page-locked"])
@pytest.mark.parametrize("cw,chh,frames", CASES, ids=["%dx%dx%d=%d" % (a, b, f, a * b * f) for a, b, f in CASES])
def test_host_entry_at_the_plan_switch_points(pkg, oracle, cw, chh, frames, pinned):
    w, h = 64 * cw, 64 * chh
    rng = np.random.default_rng(cw * 131 + chh * 7 + frames)
    luma = rng.integers(0, 256, size=(frames, h, w), dtype=np.uint8)
    luma[:, : h // 2, : w // 3] = (luma[:, : h // 2, : w // 3] // 32 + 90).astype(np.uint8)  # a smoother region: both gate states occur
    blob = oracle.synth_blob(17, 8.0)
    want = oracle.predict_frames(blob, luma, w, h, frames, 32, 0.5, 0.5, mode=0)
    assert want.shape[0] == cw * chh * frames
    with pkg.EthCnn(device=0) as c:
        c.load_blob(blob)
        c.set_thresholds(0.5, 0.5)
        c.set_pass_pipeline(True)
        src = luma
        if pinned:
            src = c.host_buffer(luma.nbytes).reshape(luma.shape)
            src[...] = luma
        for rep in range(2):  # twice: the second call meets the state the first one left (events, parity, completion word)
            got = c.predict_luma(src, w, h, frames, 32)
            assert np.array_equal(_bits(got), _bits(want)), "%d CTUs, call %d" % (want.shape[0], rep)
