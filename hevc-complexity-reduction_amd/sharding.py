"""Frame-range sharding of the predictor across the GPUs of one node.

Frames are independent (the gate predicate's scope is a <=1024-CTU sub-batch INSIDE one
frame, video_to_cu_depth.py:64-72), so G workers split [0, F) into contiguous ranges and
each pwrites its slice of cu_depth.dat at a deterministic offset.  There is no exchange
step and therefore no collective; torch.distributed is used by callers only to agree on
timing (bench.py) -- never on the data path.
"""
import os


def frame_range(nframes, world, rank):
    """GPU g gets frames [floor(g F / G), floor((g+1) F / G))."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    return (rank * nframes) // world, ((rank + 1) * nframes) // world


def output_bytes(nframes, width, height):
    nctu = ((width + 63) // 64) * ((height + 63) // 64)
    return nframes * nctu * 21 * 4


def presize_output(out_path, nframes, width, height):
    """Rank 0 (or the parent) creates cu_depth.dat with its final size before workers write."""
    with open(out_path, "wb") as f:
        f.truncate(output_bytes(nframes, width, height))


def run_shard(predict_shard, yuv_path, width, height, qp, out_path, rank, world):
    """One worker's share.  `predict_shard(yuv, w, h, qp, out, f0, f1)` is
    EthCnn.predict_yuv_shard on the GPU box (tests inject a CPU checker)."""
    frame_bytes = width * height * 3 // 2
    size = os.path.getsize(yuv_path)
    if frame_bytes == 0 or size % frame_bytes:
        raise ValueError("%s: size %d is not a multiple of the frame size %d" % (yuv_path, size, frame_bytes))
    f0, f1 = frame_range(size // frame_bytes, world, rank)
    if f1 > f0:
        predict_shard(yuv_path, width, height, qp, out_path, f0, f1)
    return f0, f1


def max_over_ranks(value, dist=None, device=None):
    """bench.py's timing reduction: MAX over ranks (identity when not distributed)."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
