"""Host mirror of the reference's predictor driver
(/root/reference/HM-16.5_Test_AI/bin/video_to_cu_depth.py): same command line, same files,
same exit-status contract, so the unchanged HM hook
(TAppEncCfg.cpp:2317-2321: system("python video_to_cu_depth.py <yuv> <w> <h> <qp>"))
drives it.  Everything below the argument handling happens in libethcnn.so.

Differences from the reference, all outside the numbers it produces:
  * cu_depth.dat is written to a temp file and renamed (never a partial file);
  * the YUV is streamed (luma only) instead of being tiled in Python;
  * when the trained checkpoint blobs are absent (they are not in the reference repo),
    ETHCNN_SYNTHETIC_SEED=<n> [ETHCNN_HEAD_GAIN=<g>] opts into seeded synthetic weights;
    without it a missing checkpoint is an error (non-zero exit, HM aborts).
"""
from __future__ import print_function

import os
import sys
import time

from . import ethcnn as _e
from . import net_CNN as nt

NUM_CHANNELS = nt.NUM_CHANNELS
NUM_EXT_FEATURES = nt.NUM_EXT_FEATURES
NUM_LABEL_BYTES = nt.NUM_LABEL_BYTES
IMAGE_SIZE = nt.IMAGE_SIZE
SAVE_FILE = 'cu_depth.dat'   # video_to_cu_depth.py:20
THR_FILE = 'Thr_info.txt'    # net_CNN.py:47


def get_file_size(path):
    return os.path.getsize(path)


def get_y_conv_on_large_data(ctx, input_image, qp_seq):
    """video_to_cu_depth.py:61-73: [n,64,64,1] -> [n,21], gates per <=1024-CTU sub-batch."""
    return ctx.predict_ctus(input_image, qp_seq)


def get_prob(ctx, yuv_name, image_size, save_file, qp_seq, n_frames_start, n_frames_end, frame_width, frame_height):
    """video_to_cu_depth.py:75-118: frames [n_frames_start, n_frames_end) of the file -> save_file (which then holds exactly
    those frames; the reference reads and discards the first n_frames_start, :86-87).  Its own call passes 0 and the frame
    count (:139-140)."""
    assert image_size == IMAGE_SIZE
    frame_bytes = frame_width * frame_height * 3 // 2
    total = get_file_size(yuv_name) // frame_bytes
    if n_frames_start == 0 and n_frames_end == total:
        return ctx.predict_yuv_file(yuv_name, frame_width, frame_height, qp_seq, save_file)
    if not 0 <= n_frames_start <= n_frames_end <= total:
        raise ValueError("get_prob: frame range [%d, %d) outside the file's %d frames" % (n_frames_start, n_frames_end, total))
    return ctx.predict_yuv_range(yuv_name, frame_width, frame_height, qp_seq, save_file, n_frames_start, n_frames_end)


def restore_model(ctx, qp_seq, model_dir='.'):
    """video_to_cu_depth.py:126-133: QP band -> checkpoint prefix -> saver.restore."""
    prefix = os.path.join(model_dir, _e.model_name_for_qp(qp_seq))
    seed = os.environ.get('ETHCNN_SYNTHETIC_SEED')
    if os.path.exists(prefix + '.data-00000-of-00001') or seed is None:
        ctx.load_checkpoint(prefix)
        return prefix
    ctx.load_synthetic(int(seed), float(os.environ.get('ETHCNN_HEAD_GAIN', '1.0')))
    return 'synthetic(seed=%s)' % seed


def guard_fast_plan(ctx):
    """ETHCNN_FC1_PLAN=2|3 is an opt-in; the encoder asserts a zero exit status (TAppEncCfg.cpp:2321).  The library REFUSES a plan
    whose load-time accuracy guard fails for the restored checkpoint (include/ethcnn.h, ethcnn_check_fc1_plan); the launcher then says
    so on stderr and runs the exact plan -- always correct, only slower -- instead of failing the encode."""
    plan = ctx.fc1_plan()
    if plan:
        g = ctx.check_fc1_plan(plan)
        if not g['accepted']:
            sys.stderr.write('video_to_cu_depth: %s\nvideo_to_cu_depth: continuing with the exact plan (0)\n' % g['message'])
            ctx.set_fc1_plan(0)


def _shard_worker(device, yuv_file, width, height, qp_seq, out_path, f0, f1, thr, nworkers=1):
    """One process per GPU: own context, own frame range, pwrite into out_path.
    The node's host-CPU budget is shared: every worker starts budget / nworkers staging-fill threads
    (ethcnn_host_thread_budget), not a full pool each."""
    os.environ['ETHCNN_LOCAL_WORKERS'] = str(max(1, int(nworkers)))
    ctx = _e.EthCnn(device=device)
    ctx.set_thresholds(*thr)
    restore_model(ctx, qp_seq)
    guard_fast_plan(ctx)
    ctx.predict_yuv_shard(yuv_file, width, height, qp_seq, out_path, f0, f1)
    ctx.close()


def predict_sharded(yuv_file, width, height, qp_seq, save_file, devices):
    """Frame-range sharding over `devices` (list of HIP ordinals).  No collective: ranges are disjoint and the output offsets
    deterministic.  Default: ONE process, a worker thread per device inside the library (ethcnn_predict_yuv_file_sharded: one
    interpreter, one checkpoint parse, peers cloned from the first context -- the encoder blocks on this command, TAppEncCfg.cpp:2317-2321).
    ETHCNN_SHARD_PROCESSES=1: a worker PROCESS per device (the form the torchrun bench uses), each with its own context."""
    if os.environ.get('ETHCNN_SHARD_PROCESSES', '0') in ('', '0'):
        ctx = _e.EthCnn(device=devices[0])
        ctx.load_thresholds(THR_FILE)
        restore_model(ctx, qp_seq)
        guard_fast_plan(ctx)
        n = ctx.predict_yuv_file_sharded(devices, yuv_file, width, height, qp_seq, save_file)
        ctx.close()
        return n
    import multiprocessing as mp
    from . import sharding
    n_frames = get_file_size(yuv_file) // (width * height * 3 // 2)
    thr = nt.get_thresholds(THR_FILE)
    tmp = '%s.tmp.%d' % (save_file, os.getpid())
    sharding.presize_output(tmp, n_frames, width, height)
    mpctx = mp.get_context('spawn')
    procs = []
    ranges = [(dev,) + sharding.frame_range(n_frames, len(devices), g) for g, dev in enumerate(devices)]
    ranges = [r for r in ranges if r[2] > r[1]]
    for dev, f0, f1 in ranges:
        p = mpctx.Process(target=_shard_worker, args=(dev, yuv_file, width, height, qp_seq, tmp, f0, f1, thr, len(ranges)))
        p.start()
        procs.append(p)
    ok = True
    for p in procs:
        p.join()
        ok = ok and p.exitcode == 0
    if not ok:
        os.remove(tmp)
        raise RuntimeError('a shard worker failed')
    os.replace(tmp, save_file)
    return n_frames


def main(argv=None):
    argv = sys.argv if argv is None else argv
    assert len(argv) == 5                      # :120
    yuv_file = argv[1]
    width, height, qp_seq = int(argv[2]), int(argv[3]), int(argv[4])
    frame_bytes = width * height * 3 // 2
    assert frame_bytes > 0 and get_file_size(yuv_file) % frame_bytes == 0   # :137
    # ETHCNN_DEVICES="0,1,2,3" shards frames over several GPUs; default: one GPU (ETHCNN_DEVICE)
    devices = [int(d) for d in os.environ.get('ETHCNN_DEVICES', os.environ.get('ETHCNN_DEVICE', '0')).split(',')]
    timing = os.environ.get('ETHCNN_TIMING', '0') not in ('', '0')
    stamps = [('imports done', time.perf_counter())]
    t1 = time.time()
    if len(devices) > 1:
        n_frames = predict_sharded(yuv_file, width, height, qp_seq, SAVE_FILE, devices)
        stamps.append(('create + weights + predict (%d workers)' % len(devices), time.perf_counter()))
    else:
        ctx = _e.EthCnn(device=devices[0])
        stamps.append(('create (HIP runtime init %.1f, whole call %.1f)' % ctx.startup_times(), time.perf_counter()))
        ctx.load_thresholds(THR_FILE)          # net_CNN.py:47 (cwd-relative; at import time there)
        restore_model(ctx, qp_seq)
        guard_fast_plan(ctx)
        stamps.append(('thresholds + weights + plan guard', time.perf_counter()))
        t1 = time.time()                       # the reference times get_prob only (:142-145)
        n_frames = get_prob(ctx, yuv_file, IMAGE_SIZE, SAVE_FILE, qp_seq, 0,
                            get_file_size(yuv_file) // frame_bytes, width, height)
        stamps.append(('predict', time.perf_counter()))
        ctx.close()
        stamps.append(('destroy', time.perf_counter()))
    t2 = time.time()
    if timing:  # (perf_counter is CLOCK_MONOTONIC: comparable with the spawning process's stamp, ETHCNN_T0_MS)
        t0 = float(os.environ.get('ETHCNN_T0_MS', '0')) * 1e-3
        up = float(os.environ.get('ETHCNN_T_UP_MS', '0')) * 1e-3
        prev, parts = (up or stamps[0][1]), []
        if t0 and up:
            parts.append('spawn -> interpreter up %.1f' % ((up - t0) * 1e3))
        for name, t in stamps:
            parts.append('%s %.1f' % (name, (t - prev) * 1e3))
            prev = t
        sys.stderr.write('video_to_cu_depth.py timing (ms): ' + ' | '.join(parts) + '\n')
    print('%s  frame %d/%d  %dx%d' % (yuv_file, n_frames, n_frames, width, height))
    print('--------\n\nPredicting Time: %.3f sec.\n\n--------' % float(t2 - t1))  # :145
    return 0
