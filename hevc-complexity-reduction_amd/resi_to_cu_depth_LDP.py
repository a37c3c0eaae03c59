"""Host mirror of the reference's LDP predictor daemon
(/root/reference/HM-16.5_Test_LDP/bin/resi_to_cu_depth_LDP.py): same files, same handshake, so
the unchanged HM-LDP encoder (TEncGOP.cpp:1463-1503) drives it:

    HM:     pre-encode -> resi.yuv ; command.dat = "<POC> <w> <h> <qp> [end]" ; touch pred_start.sig
    daemon: read command, remove pred_start.sig, (re)load the LSTM model when the QP changed,
            predict, write state.dat, cu_depth.dat, then touch pred_end.sig
    HM:     spin on pred_end.sig, remove it, fread cu_depth.dat (nctu x 21 float32)

Everything numeric happens in libethcnn.so (ethcnn_ldp_predict_frame: resi_cnn + one ETH-LSTM
step + heads + gates on the GPU).  Differences from the reference, none in the numbers:
  * cu_depth.dat / state.dat are written to temp files and renamed before pred_end.sig appears;
  * the poll loop sleeps 200 us between checks instead of spinning, and `serve` can stop after
    `max_frames` or an idle timeout (tests; the reference loops forever);
  * the recurrent state is also kept in memory (state.dat is still written and is still the
    source after a restart, as in get_state_in_from_one_file);
  * missing trained CNN blob (model_LDP_2000000_qp22~37.dat.data is not in the reference repo):
    ETHCNN_SYNTHETIC_SEED=<n> opts into seeded synthetic CNN weights, otherwise it is an error.
"""
from __future__ import print_function

import os
import sys
import time

import numpy as np

from . import ethcnn as _e
from . import net_CNN as nt

IMAGE_SIZE = nt.IMAGE_SIZE
NUM_CHANNELS = nt.NUM_CHANNELS
NUM_EXT_FEATURES = 2      # QP and POC (resi_to_cu_depth_LDP.py:19)
VECTOR_LENGTH = _e.NVEC   # config.py VECTOR_LENGTH = 64 + 128 + 256
LSTM_MAX_LENGTH = 1
LSTM_DEPTH = 1
MINI_BATCH_SIZE = 1024    # :118 (gate scope; applied inside the library)

COMPLETE_FILE = 'complete.dat'
YUV_FILE = 'resi.yuv'
STATE_FILE = 'state.dat'
SAVE_FILE = 'cu_depth.dat'
COMMAND_FILE = 'command.dat'
START_FILE = 'pred_start.sig'
END_FILE = 'pred_end.sig'
THR_FILE = 'Thr_info.txt'
MODEL_CNN_FILE = 'model_LDP_2000000_qp22~37.dat'  # :159


def send_init_signal(init_file):
    with open(init_file, 'w+') as f:
        f.write('1')
    print('Python: predictor initialized.')


def send_complete_signal(complete_file):
    with open(complete_file, 'w+') as f:
        f.write('1')
    print('Python: Operation completed.')


def get_command(command_file):
    """:56-72: '<i_frame> <w> <h> <qp> [end]' -> ints, or four -1 while the line is incomplete."""
    try:
        with open(command_file, 'r') as f:
            str_arr = f.readline().split(' ')
    except IOError:
        return -1, -1, -1, -1
    if len(str_arr) == 5 and str_arr[4] == '[end]':
        try:
            return int(str_arr[0]), int(str_arr[1]), int(str_arr[2]), int(str_arr[3])
        except ValueError:
            pass
    return -1, -1, -1, -1


def get_images_from_one_file(yuv_file, frame_width, frame_height, CUwidth=IMAGE_SIZE):
    """:74-101 reads the first frame's luma; the zero-padded 64x64 tiling happens on the GPU.
    Returns (luma [h, w] uint8, num_vectors)."""
    assert CUwidth == IMAGE_SIZE
    with open(yuv_file, 'rb') as f:
        y_buf = f.read(frame_width * frame_height)
    if len(y_buf) != frame_width * frame_height:
        raise IOError('%s: short read (%d of %d luma bytes)' % (yuv_file, len(y_buf), frame_width * frame_height))
    luma = np.frombuffer(y_buf, dtype=np.uint8).reshape(frame_height, frame_width)
    return luma, _e.ctus_per_frame(frame_width, frame_height)


def get_state_in_from_one_file(state_file, num_vectors, i_frame):
    """:103-112: zeros for i_frame <= 1 (returned as None = zeros inside the library)."""
    if i_frame > 1:
        want = num_vectors * LSTM_DEPTH * 2 * VECTOR_LENGTH
        state_in = np.fromfile(state_file, dtype=np.float32, count=want)
        if state_in.size != want:
            raise IOError('%s: holds %d floats, need %d' % (state_file, state_in.size, want))
        return state_in.reshape(num_vectors, LSTM_DEPTH, 2, VECTOR_LENGTH)
    return None


def predict_cu_depth(ctx, luma, frame_width, frame_height, state_in, qp_seq, i_frame):
    """:114-129 -> (depth_out [n, 21], state_out [n, 1, 2, 448])"""
    n = _e.ctus_per_frame(frame_width, frame_height)
    sin = None if state_in is None else np.asarray(state_in, dtype=np.float32).reshape(n, 2, VECTOR_LENGTH)
    depth_out, state_out = ctx.ldp_predict_frame(luma, frame_width, frame_height, qp_seq, i_frame, sin)
    return depth_out, state_out.reshape(n, LSTM_DEPTH, 2, VECTOR_LENGTH)


def _write_atomic(path, arr):
    tmp = '%s.tmp.%d' % (path, os.getpid())
    with open(tmp, 'wb') as f:
        f.write(np.ascontiguousarray(arr, dtype=np.float32).tobytes())
    os.rename(tmp, path)


def save_cu_depth_and_state(depth_out, state_out, save_file, state_file, end_file, num_vectors):
    """:131-145: state.dat, cu_depth.dat, then the (empty) ending signal."""
    assert depth_out.size == num_vectors * (1 + 4 + 16)
    _write_atomic(state_file, state_out)
    _write_atomic(save_file, depth_out)
    open(end_file, 'wb').close()


def restore_cnn(ctx, model_dir='.'):
    prefix = os.path.join(model_dir, MODEL_CNN_FILE)
    seed = os.environ.get('ETHCNN_SYNTHETIC_SEED')
    if os.path.exists(prefix + '.data-00000-of-00001') or seed is None:
        ctx.load_checkpoint(prefix)
        return prefix
    ctx.load_synthetic(int(seed), float(os.environ.get('ETHCNN_HEAD_GAIN', '1.0')))
    return 'synthetic(seed=%s)' % seed


def restore_lstm(ctx, qp_seq, model_dir='.'):
    """:166-179: QP band -> LSTM checkpoint."""
    prefix = os.path.join(model_dir, _e.lstm_model_name_for_qp(qp_seq))
    seed = os.environ.get('ETHCNN_SYNTHETIC_SEED')
    if os.path.exists(prefix + '.data-00000-of-00001') or seed is None:
        ctx.load_lstm_checkpoint(prefix)
        return prefix
    ctx.load_lstm_synthetic(int(seed), float(os.environ.get('ETHCNN_HEAD_GAIN', '1.0')))
    return 'synthetic(seed=%s)' % seed


def serve(workdir='.', max_frames=None, idle_timeout=None, poll_s=2e-4, device=0, verbose=True):
    """The daemon loop (:148-190).  Returns the number of frames predicted."""
    p = lambda name: os.path.join(workdir, name)
    ctx = _e.EthCnn(device=device)
    try:
        ctx.load_thresholds(p(THR_FILE))
        restore_cnn(ctx, workdir)
        if verbose:
            print('Python: predictor initialized on %s.' % ctx.device_name)
        n_frame_total, qp_seq = 0, 0
        last_state, last_key = None, None
        idle_since = time.time()
        while max_frames is None or n_frame_total < max_frames:
            if not os.path.isfile(p(START_FILE)):
                if idle_timeout is not None and time.time() - idle_since > idle_timeout:
                    break
                time.sleep(poll_s)
                continue
            i_frame, frame_width, frame_height, qp_seq_temp = get_command(p(COMMAND_FILE))
            if i_frame < 0:
                time.sleep(poll_s)
                continue
            qp_seq_last, qp_seq = qp_seq, qp_seq_temp
            os.remove(p(START_FILE))
            if qp_seq != qp_seq_last:
                name = restore_lstm(ctx, qp_seq, workdir)
                if verbose:
                    print('Set QP = %d' % qp_seq)
                    print('LSTM model loaded (%s).' % name)
            luma, num_vectors = get_images_from_one_file(p(YUV_FILE), frame_width, frame_height, IMAGE_SIZE)
            key = (frame_width, frame_height, i_frame - 1)
            if i_frame > 1 and last_key == key and last_state is not None:
                state_in = last_state  # == what state.dat holds (written below on the previous frame)
            else:
                state_in = get_state_in_from_one_file(p(STATE_FILE), num_vectors, i_frame)
            depth_out, state_out = predict_cu_depth(ctx, luma, frame_width, frame_height, state_in, qp_seq, i_frame)
            save_cu_depth_and_state(depth_out, state_out, p(SAVE_FILE), p(STATE_FILE), p(END_FILE), num_vectors)
            last_state, last_key = state_out, (frame_width, frame_height, i_frame)
            n_frame_total += 1
            idle_since = time.time()
            if verbose:
                print('%d frames predicted.' % n_frame_total)
        return n_frame_total
    finally:
        ctx.close()


def main(argv):
    """`python resi_to_cu_depth_LDP.py` in HM-LDP's bin/ (no arguments, like the reference);
    optional: --max-frames N, --idle-timeout SECONDS."""
    max_frames = idle = None
    args = list(argv[1:])
    while args:
        a = args.pop(0)
        if a == '--max-frames':
            max_frames = int(args.pop(0))
        elif a == '--idle-timeout':
            idle = float(args.pop(0))
        else:
            sys.stderr.write('usage: resi_to_cu_depth_LDP.py [--max-frames N] [--idle-timeout S]\n')
            return 2
    serve('.', max_frames=max_frames, idle_timeout=idle)
    return 0
