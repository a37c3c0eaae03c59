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
  * the recurrent state stays RESIDENT IN HBM between frames (ethcnn_ldp_step): only the daemon itself
    ever reads state.dat back (:103-106), so the per-frame 2 x nctu x 3.5 KB PCIe round trip is not on the
    encoder's critical path.  state.dat is still written every frame for protocol compatibility -- AFTER
    cu_depth.dat and pred_end.sig, while HM is already encoding -- and it is still the source whenever
    the resident state cannot be the right one (daemon restart, a frame out of sequence, a state.dat
    somebody else replaced: inode / size / mtime are checked).  A sidecar `state.dat.idx` brackets the late write:
    "pending <i_frame> <w> <h>" appears BEFORE pred_end.sig, "<i_frame> <w> <h>" replaces it once state.dat holds that
    frame's state.  A restarted daemon that finds "pending" (the previous daemon died between the ending signal and the
    state write, so state.dat still belongs to an earlier frame) refuses the file instead of silently feeding a stale
    state into the recurrence; a state.dat without sidecar (the reference daemon's) is taken as it is, whatever frame
    it is from -- as the reference does (HM may skip frames: intra pictures are not predicted);
  * resi.yuv is read straight into pinned host memory (DMA-able without a staging copy);
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
STATE_INDEX_SUFFIX = '.idx'   # ours: "[pending] <i_frame> <w> <h>": the frame whose output state state.dat holds / is about to hold
SAVE_FILE = 'cu_depth.dat'
COMMAND_FILE = 'command.dat'
START_FILE = 'pred_start.sig'
END_FILE = 'pred_end.sig'
THR_FILE = 'Thr_info.txt'
MODEL_CNN_FILE = 'model_LDP_2000000_qp22~37.dat'  # :159


class StaleStateError(IOError):
    """state.dat cannot be the state of the previous frame (see get_state_in_from_one_file)."""


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


def get_images_from_one_file(yuv_file, frame_width, frame_height, CUwidth=IMAGE_SIZE, into=None):
    """:74-101 reads the first frame's luma; the zero-padded 64x64 tiling happens on the GPU.
    Returns (luma [h, w] uint8, num_vectors).  `into`: a (pinned) uint8 buffer to read into."""
    assert CUwidth == IMAGE_SIZE
    want = frame_width * frame_height
    with open(yuv_file, 'rb') as f:
        if into is not None:
            got = f.readinto(memoryview(into[:want]))
            luma = into[:want]
        else:
            y_buf = f.read(want)
            got = len(y_buf)
            luma = np.frombuffer(y_buf, dtype=np.uint8)
    if got != want:
        raise IOError('%s: short read (%d of %d luma bytes)' % (yuv_file, got, want))
    return luma.reshape(frame_height, frame_width), _e.ctus_per_frame(frame_width, frame_height)


def get_state_in_from_one_file(state_file, num_vectors, i_frame, geometry=None):
    """:103-112: zeros for i_frame <= 1 (returned as None = zeros inside the library).  The sidecar this daemon writes
    must not say "pending" (the state write behind an ending signal never completed: the file is an EARLIER frame's) and
    must name `geometry` = (w, h) when given: a stale state is an error, never a silently wrong recurrence."""
    if i_frame > 1:
        try:
            with open(state_file + STATE_INDEX_SUFFIX, 'r') as f:
                tag = f.read().split()
        except (IOError, OSError):
            tag = None  # no sidecar: somebody else's state.dat (the reference daemon writes none) -> trusted as there
        if tag is not None:
            if tag[:1] == ['pending']:
                raise StaleStateError('%s is stale: the state of frame %s was never written (the daemon that served it stopped after '
                                      'its ending signal); delete %s%s to accept %s as it is, or restart the encode'
                                      % (state_file, tag[1] if len(tag) > 1 else '?', state_file, STATE_INDEX_SUFFIX, state_file))
            if len(tag) != 3 or (geometry is not None and [str(int(g)) for g in geometry] != tag[1:]):
                raise IOError('%s belongs to another sequence (%s), frame %d is %s' % (state_file, ' '.join(tag), i_frame, geometry))
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


def _file_sig(path):
    try:
        st = os.stat(path)
        return st.st_ino, st.st_size, st.st_mtime_ns  # inode: a same-size replacement inside one timestamp tick is still seen
    except OSError:
        return None


def _write_atomic(path, arr):
    tmp = '%s.tmp.%d' % (path, os.getpid())
    with open(tmp, 'wb') as f:
        f.write(np.ascontiguousarray(arr, dtype=np.float32).tobytes())
    os.rename(tmp, path)


def save_cu_depth_and_state(depth_out, state_out, save_file, state_file, end_file, num_vectors, tag=None):
    """:131-145 writes state.dat, cu_depth.dat, then the (empty) ending signal.  Here cu_depth.dat and the ending
    signal come FIRST (they are what HM waits for) and state.dat is refreshed afterwards, while HM is already
    encoding: `state_out` may be a callable that fetches the state from the GPU at that point.  Returns the state."""
    assert depth_out.size == num_vectors * (1 + 4 + 16)

    def sidecar(text):  # tag = (i_frame, w, h)
        tmp = '%s%s.tmp.%d' % (state_file, STATE_INDEX_SUFFIX, os.getpid())
        with open(tmp, 'w') as f:
            f.write(text + '\n')
        os.rename(tmp, state_file + STATE_INDEX_SUFFIX)
    if tag is not None:
        sidecar('pending %d %d %d' % tuple(tag))  # from here until the state write below, state.dat is an earlier frame's
    _write_atomic(save_file, depth_out)
    open(end_file, 'wb').close()
    if callable(state_out):
        state_out = state_out()
    _write_atomic(state_file, state_out)
    if tag is not None:
        sidecar('%d %d %d' % tuple(tag))
    return state_out


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


def serve(workdir='.', max_frames=None, idle_timeout=None, poll_s=2e-4, device=0, verbose=True, accept_stale=False):
    """The daemon loop (:148-190).  Returns the number of frames predicted.
    A stale state.dat (sidecar says "pending": the daemon that served that frame stopped between its ending signal and its
    state write) stops the daemon with StaleStateError and a message that names the recovery -- delete state.dat.idx to accept
    state.dat as it is, or restart the encode -- unless accept_stale is set, in which case the file is used as it is after
    a warning.  (HM then keeps spinning on pred_end.sig, exactly as it does when the reference's daemon dies.)"""
    p = lambda name: os.path.join(workdir, name)
    ctx = _e.EthCnn(device=device)
    try:
        ctx.load_thresholds(p(THR_FILE))
        restore_cnn(ctx, workdir)
        if verbose:
            print('Python: predictor initialized on %s.' % ctx.device_name)
        n_frame_total, qp_seq = 0, 0
        last_key, state_sig = None, None   # (w, h, i_frame) of the resident state; (size, mtime_ns) of the state.dat we wrote
        pinned, pinned_probs = None, None  # capacities tracked separately: 65x65 has fewer pixels but more CTUs than 128x64
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
            need_px, need_pr = frame_width * frame_height, _e.ctus_per_frame(frame_width, frame_height) * 21
            if pinned is None or pinned.size < need_px or pinned_probs.size < need_pr:
                need_px = max(need_px, 0 if pinned is None else pinned.size)
                need_pr = max(need_pr, 0 if pinned_probs is None else pinned_probs.size)
                ctx.free_host_buffers()
                pinned = ctx.host_buffer(need_px)
                pinned_probs = ctx.host_buffer(need_pr * 4).view(np.float32)
            luma, num_vectors = get_images_from_one_file(p(YUV_FILE), frame_width, frame_height, IMAGE_SIZE, into=pinned)
            # the state of frame i_frame - 1 is resident in HBM when this daemon produced it for this geometry and the
            # state.dat it wrote then is still the one on disk; anything else goes through the file, as in the reference
            resident = i_frame > 1 and last_key == (frame_width, frame_height, i_frame - 1) and state_sig == _file_sig(p(STATE_FILE))
            try:
                state_in = None if (resident or i_frame <= 1) else get_state_in_from_one_file(p(STATE_FILE), num_vectors, i_frame,
                                                                                               (frame_width, frame_height))
            except StaleStateError as exc:
                sys.stderr.write('resi_to_cu_depth_LDP: %s\n' % exc)
                if not accept_stale:
                    raise
                os.remove(p(STATE_FILE) + STATE_INDEX_SUFFIX)  # accepted: state.dat is taken as it is, as the reference would
                state_in = get_state_in_from_one_file(p(STATE_FILE), num_vectors, i_frame, (frame_width, frame_height))
            if state_in is not None:
                state_in = np.asarray(state_in, dtype=np.float32).reshape(num_vectors, 2, VECTOR_LENGTH)
            depth_out = ctx.ldp_step(luma, frame_width, frame_height, qp_seq, i_frame, state_in,
                                     probs_out=pinned_probs[:num_vectors * 21].reshape(num_vectors, 21))
            save_cu_depth_and_state(depth_out, lambda: ctx.ldp_get_state(frame_width, frame_height), p(SAVE_FILE),
                                    p(STATE_FILE), p(END_FILE), num_vectors, tag=(i_frame, frame_width, frame_height))
            last_key, state_sig = (frame_width, frame_height, i_frame), _file_sig(p(STATE_FILE))
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
    accept_stale = False
    args = list(argv[1:])
    while args:
        a = args.pop(0)
        if a == '--max-frames':
            max_frames = int(args.pop(0))
        elif a == '--idle-timeout':
            idle = float(args.pop(0))
        elif a == '--accept-stale':
            accept_stale = True
        else:
            sys.stderr.write('usage: resi_to_cu_depth_LDP.py [--max-frames N] [--idle-timeout S] [--accept-stale]\n')
            return 2
    try:
        serve('.', max_frames=max_frames, idle_timeout=idle, accept_stale=accept_stale)
    except StaleStateError:
        return 1  # (the message, with the recovery step, is already on stderr)
    return 0
