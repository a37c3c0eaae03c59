"""Deterministic CTU generator for the large golden sets (tests + tests/golden/gen_meta_exec_golden.py).

Version-independent (a splitmix64 counter stream, no numpy Generator), so a golden file can hold just
(seed, n, expected outputs) instead of 4 KiB of pixels per CTU.  Classes, cycling by index:
  0 full-range noise            1 smooth gradient + small noise      2 flat (one value)
  3 saturated / extreme         4 right-edge zero padding            5 bottom-edge zero padding
  6 corner (both) zero padding  7 low-contrast texture around a mid level
(4-6 are what video_to_cu_depth.py:54-57 produces for partial CTUs: zeros beyond the frame edge.)
"""
import numpy as np


def _splitmix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _bytes(seed, idx, count):
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(idx))
        h = _splitmix64(base + np.arange((count + 7) // 8, dtype=np.uint64))
    return h.view(np.uint8)[:count]


def make_ctus(seed, n, residual=False):
    """-> uint8 [n,64,64].  residual=True: values centred on 128 (resi.yuv-like) for the LDP front-end."""
    out = np.empty((n, 64, 64), dtype=np.uint8)
    yy, xx = np.mgrid[0:64, 0:64]
    for i in range(n):
        r = _bytes(seed, i, 4096 + 16)
        noise, p = r[:4096].reshape(64, 64).astype(np.int32), r[4096:].astype(np.int32)
        k = i % 8
        if k == 0:
            c = noise
        elif k == 1:
            c = (yy * (p[0] % 5) + xx * (p[1] % 5)) // 2 + p[2] % 64 + noise % 8
        elif k == 2:
            c = np.full((64, 64), p[0])
        elif k == 3:
            c = [np.zeros((64, 64), np.int32), np.full((64, 64), 255), ((yy // 8 + xx // 8) % 2) * 255,
                 ((yy // 16 + xx // 16) % 2) * 255][p[0] % 4]
        elif k in (4, 5, 6):
            c = noise if p[3] % 2 else (yy * 2 + xx + p[2]) % 256
            c = c.copy()
            if k in (4, 6):
                c[:, 1 + p[0] % 63:] = 0
            if k in (5, 6):
                c[1 + p[1] % 63:, :] = 0
        else:
            c = 96 + p[0] % 64 + noise % (2 + p[1] % 14)
        if residual:
            c = 128 + (c.astype(np.int32) - 128) // (2 + p[4] % 6)
            if k == 3:
                c = [np.full((64, 64), 128), np.zeros((64, 64), np.int32), np.full((64, 64), 255), 128 + ((yy + xx) % 2) * 3][p[0] % 4]
        out[i] = np.clip(c, 0, 255).astype(np.uint8)
    return out


def crc(ctus):
    import zlib
    return zlib.crc32(np.ascontiguousarray(ctus).tobytes())


def make_frame(seed, w, h, residual=False, flat_from_ctu=None):
    """-> uint8 luma [h, w]: the CTUs of make_ctus(seed, nctu) laid out in raster order, cropped to the frame.
    flat_from_ctu = k: every CTU with raster index >= k is flat (value 128): after the 16x16 mean removal such CTUs are
    all-zero inputs, so every one of them yields the same probabilities (used to steer the <=1024-CTU batch gates)."""
    wc, hc = (w + 63) // 64, (h + 63) // 64
    ctus = make_ctus(seed, wc * hc, residual=residual)
    if flat_from_ctu is not None:
        ctus[flat_from_ctu:] = 128
    full = ctus.reshape(hc, wc, 64, 64).transpose(0, 2, 1, 3).reshape(hc * 64, wc * 64)
    return np.ascontiguousarray(full[:h, :w])


def yuv420_bytes(luma_frames):
    """planar 4:2:0 file content for a list of luma planes (chroma = 128), as video_to_cu_depth.py:47-48 reads it"""
    out = bytearray()
    for y in luma_frames:
        h, w = y.shape
        out += y.tobytes() + bytes([128]) * (w * h // 2)
    return bytes(out)
