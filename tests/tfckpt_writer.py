"""Test helper: writes a minimal, valid TensorFlow V2 checkpoint bundle (<prefix>.index +
<prefix>.data-00000-of-00001) without TensorFlow, to exercise the product's reader.
Pure-Python crc32c here is deliberately independent of the library's implementation."""
import struct

import numpy as np

_T = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _T.append(_c)


def crc32c(data):
    c = 0xFFFFFFFF
    for b in bytes(data):
        c = _T[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _entry_proto(shape, offset, size, crc_masked):
    dims = b"".join(b"\x12" + _varint(len(d)) + d for d in (b"\x08" + _varint(s) for s in shape))
    return (b"\x08\x01" + b"\x12" + _varint(len(dims)) + dims + b"\x20" + _varint(offset) +
            b"\x28" + _varint(size) + b"\x35" + struct.pack("<I", crc_masked))


def _block(entries, restart_interval=16, prefix_compress=True):
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        elif prefix_compress:
            while shared < min(len(k), len(prev)) and k[shared] == prev[shared]:
                shared += 1
        out += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        prev = k
    for r in restarts or [0]:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts) or 1)
    return bytes(out)


def write_bundle(prefix, tensors, data_crc=None, corrupt=None):
    """tensors: list of (name, float32 ndarray) -- written in sorted-key order, back to back.
    data_crc: callable(bytes)->masked crc (defaults to the pure-Python one).
    corrupt: None | 'data' (flip a payload byte after the crc is taken) | 'index_crc'."""
    data_crc = data_crc or (lambda b: mask(crc32c(b)))
    tensors = sorted(tensors, key=lambda t: t[0])
    payload, entries = bytearray(), [(b"", b"\x08\x01\x10\x00\x1a\x02\x08\x01")]  # BundleHeaderProto
    for name, arr in tensors:
        raw = np.ascontiguousarray(arr, dtype="<f4").tobytes()
        entries.append((name.encode(), _entry_proto(arr.shape, len(payload), len(raw), data_crc(raw))))
        payload += raw
    if corrupt == "data":
        payload[len(payload) // 2] ^= 0x40
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(payload)
    blk = _block(entries)
    trailer = b"\x00" + struct.pack("<I", mask(crc32c(blk + b"\x00")) ^ (1 if corrupt == "index_crc" else 0))
    file = bytearray(blk + trailer)
    meta = _block([])
    meta_off = len(file)
    file += meta + b"\x00" + struct.pack("<I", mask(crc32c(meta + b"\x00")))
    idx = _block([(tensors[-1][0].encode() + b"\xff", _varint(0) + _varint(len(blk)))])
    idx_off = len(file)
    file += idx + b"\x00" + struct.pack("<I", mask(crc32c(idx + b"\x00")))
    footer = _varint(meta_off) + _varint(len(meta)) + _varint(idx_off) + _varint(len(idx))
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    file += footer
    with open(prefix + ".index", "wb") as f:
        f.write(file)
