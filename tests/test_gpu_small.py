"""GPU: the single-launch small pass (csrc/ethcnn_small.hip: CTU load + trunk -> FC1 -> heads -> gates as a dataflow inside one
grid; register-fed FC1 / heads up to 576 CTUs, LDS-staged above) against the oracle and against the five-launch path
(ethcnn_set_small_pass_launch off; its FC1 is one of the register-fed k_fc1_regs shapes at these sizes): bit-identical probabilities,
features, FC1 outputs, LDP vectors and LDP recurrences over aligned geometries of every FC1 shape, zero-padded edges, pitched
planes, several frames per pass, closed / mixed gates, and long call sequences (the launch's last block must leave its sync
area zero for the next one)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _luma(rng, frames, h, pitch, w):
    a = rng.integers(0, 256, size=(frames, h, pitch), dtype=np.uint8)
    a[:, : h // 3] = a[:, : h // 3] // 16 + 100          # a smooth band: low split probabilities
    a[:, :, w:] = 0xEE                                    # the pitch gap must never be read as pixels
    return a


GEOMS = [  # (width, height, frames, pitch): every FC1 shape of the launch (<= 576, <= 2304, more rows), ragged right / bottom CTUs
    (768, 512, 1, 768), (1920, 1080, 1, 1920), (416, 240, 1, 416), (400, 136, 3, 448), (832, 480, 2, 832),
    (3840, 2160, 1, 3840), (4928, 3264, 1, 4928), (1920, 1080, 3, 2048), (64, 64, 1, 64), (16, 16, 5, 16), (1280, 720, 8, 1280),
    (2560, 1440, 1, 2560)]  # (920 CTUs: the 64 x 32 register-fed FC1 kernel of the multi-launch path)


@pytest.mark.parametrize("geom", GEOMS)
def test_single_launch_pass_matches_oracle_and_five_launches(pkg, oracle, geom):
    w, h, frames, pitch = geom
    rng = np.random.default_rng(w * 31 + h)
    blob = oracle.synth_blob(6, 4.0)
    luma = _luma(rng, frames, h, pitch, w)
    nctu = pkg.ethcnn.ctus_per_frame(w, h)
    c = pkg.EthCnn(device=0)
    c.load_blob(blob)
    try:
        d_in, d_out = c.alloc(luma.nbytes), c.alloc(frames * nctu * 84)
        d_in.upload(luma)

        def run(on):
            c.set_small_pass_launch(on)
            c.predict_luma_device(d_in, w, h, frames, 30, d_out, pitch=pitch)
            c.synchronize()
            return d_out.download(np.float32, frames * nctu * 21).reshape(-1, 21)
        c.set_thresholds(-1.0, -1.0)
        raw = run(True)
        m64 = float(raw[: min(1024, nctu), 0].max())
        for t1, t2 in ((0.5, 0.5), (m64, 0.5), (float(np.median(raw[:, 0])), float(np.median(raw[:, 1:5]))), (2.0, -0.5)):
            c.set_thresholds(t1, t2)
            want = oracle.predict_frames(blob, luma, w, h, frames, 30, t1, t2, mode=0, pitch=pitch)
            got1 = run(True)
            st = c.stage_times()
            assert np.array_equal(_bits(got1), _bits(want)), (geom, t1, t2)
            assert np.array_equal(_bits(run(False)), _bits(want)), (geom, t1, t2)
        # the launch really was ONE kernel; intermediates are where debug_fetch expects them
        c.set_profiling(2)
        c.reset_stage_times()
        run(True)
        st = c.stage_times()["launches"]
        n = frames * nctu
        if n <= 2304:   # up to one 3840x2160 picture; beyond that five full launches win and the library keeps them
            assert (st["tile"], st["trunk"], st["fc1"], st["heads"], st["gate"]) == (0, 0, 1, 0, 0), st
        else:
            assert st["tile"] == 1 and st["heads"] == 1
        f1, h1 = c.debug_fetch(pkg.ethcnn.DBG_FEATURES, n), c.debug_fetch(pkg.ethcnn.DBG_FC1, n)
        c.reset_stage_times()
        run(False)
        st = c.stage_times()["launches"]
        assert st["tile"] == 1 and st["trunk"] == 1 and st["fc1"] >= 1 and st["heads"] == 1
        assert np.array_equal(_bits(f1), _bits(c.debug_fetch(pkg.ethcnn.DBG_FEATURES, n)))
        assert np.array_equal(_bits(h1), _bits(c.debug_fetch(pkg.ethcnn.DBG_FC1, n)))
        c.set_profiling(0)
        d_in.free()
        d_out.free()
    finally:
        c.close()


def test_ldp_front_end_and_recurrence_single_launch(pkg, oracle):
    import ethcnn_lstm_np as ol
    rng = np.random.default_rng(12)
    blob, lblob = oracle.synth_blob(8, 1.0), ol.synth_lstm_blob(4, 3.0)
    c = pkg.EthCnn(device=0)
    c.load_blob(blob)
    c.load_lstm_blob(lblob)
    c.set_thresholds(0.6, 0.7)
    try:
        for (w, h) in ((1920, 1080), (416, 240), (3840, 2160), (64, 16)):
            frames = [np.clip(np.rint(128 + rng.laplace(0, 7, size=(h, w))), 0, 255).astype(np.uint8) for _ in range(3)]
            want_vec = oracle.resi_vectors(blob, frames[0], w, h, mode=0)
            for on in (True, False):
                c.set_small_pass_launch(on)
                assert np.array_equal(_bits(c.resi_vectors(frames[0], w, h)), _bits(want_vec)), (w, h, on)
                gs = os_ = None
                for i, fr in enumerate(frames, 1):
                    gp, gs = c.ldp_predict_frame(fr, w, h, 32, i, gs)
                    op, os_ = ol.lstm_step(lblob, oracle.resi_vectors(blob, fr, w, h), os_, 32, i, 0.6, 0.7, mode=0)
                    assert np.array_equal(_bits(gp), _bits(op)) and np.array_equal(_bits(gs), _bits(os_)), (w, h, on, i)
    finally:
        c.close()


def test_long_sequences_of_single_launch_passes(pkg, oracle):
    """600 back-to-back calls over changing geometries, thresholds and both entry points, asynchronous runs in between: every
    launch finds the sync area its predecessor's last block cleared"""
    rng = np.random.default_rng(99)
    blob = oracle.synth_blob(2, 8.0)
    c = pkg.EthCnn(device=0)
    c.load_blob(blob)
    try:
        cases = []
        for (w, h, frames) in ((768, 512, 1), (1920, 1088, 1), (416, 240, 4), (128, 64, 1), (2560, 1440, 1)):
            luma = _luma(rng, frames, h, w, w)
            nctu = pkg.ethcnn.ctus_per_frame(w, h)
            d_in, d_out = c.alloc(luma.nbytes), c.alloc(frames * nctu * 84)
            d_in.upload(luma)
            want = {}
            for thr in ((0.5, 0.5), (0.97, 0.5), (2.0, 0.0)):
                want[thr] = oracle.predict_frames(blob, luma, w, h, frames, 37, thr[0], thr[1], mode=0)
            cases.append((w, h, frames, nctu, d_in, d_out, want))
        thrs = list(cases[0][6].keys())
        for k in range(600):
            w, h, frames, nctu, d_in, d_out, want = cases[int(rng.integers(len(cases)))]
            thr = thrs[int(rng.integers(len(thrs)))]
            c.set_thresholds(*thr)
            reps = 1 + int(rng.integers(3))
            for _ in range(reps):  # unsynchronised repeats: launch i+1 is enqueued while launch i runs
                c.predict_luma_device(d_in, w, h, frames, 37, d_out)
            c.synchronize()
            got = d_out.download(np.float32, frames * nctu * 21).reshape(-1, 21)
            assert np.array_equal(_bits(got), _bits(want[thr])), (k, w, h, thr)
        for case in cases:
            case[4].free()
            case[5].free()
    finally:
        c.close()


def test_pull_form_page_locked_pictures(pkg, oracle):
    """One picture in page-locked host memory (the encoder hook's buffer): the single-launch pass PULLS it over PCIe itself
    (ethcnn_small.hip, "PULL form": pull blocks -> pixel records -> trunk group by group) -- one block per group up to 1080p, 16
    blocks walking the groups above, 64 x 16 FC1 tiles at every size.  Bit-exact against the oracle, repeatedly (the sync area
    cleans itself), alternating with the direct-gather form of the same launch (a device-resident picture) on the same context,
    and with a pageable source (small: staged, then pulled; above 1024 CTUs: the banded copy)."""
    rng = np.random.default_rng(44)
    blob = oracle.synth_blob(9, 4.0)
    c = pkg.EthCnn(0)
    try:
        c.load_blob(blob)
        for (w, h) in ((64, 64), (416, 240), (1280, 720), (1920, 1080), (2560, 1600), (2576, 1600), (3840, 2160), (1024, 2304), (48, 4096)):
            luma = rng.integers(0, 256, size=(1, h, w), dtype=np.uint8)
            nctu = pkg.ethcnn.ctus_per_frame(w, h)
            pin = c.host_buffer(w * h)
            pin[:] = luma.reshape(-1)
            d_in, d_out = c.alloc(luma.nbytes), c.alloc(nctu * 84)
            d_in.upload(luma)
            for thr in ((0.5, 0.5), (0.9, 0.2)):
                c.set_thresholds(*thr)
                want = oracle.predict_frames(blob, luma, w, h, 1, 27, thr[0], thr[1], mode=0)
                for rep in range(3):
                    got = c.predict_luma(pin.reshape(1, h, w), w, h, 1, 27)
                    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (w, h, thr, rep, "page-locked")
                    c.predict_luma_device(d_in, w, h, 1, 27, d_out)
                    c.synchronize()
                    got = d_out.download(np.float32, nctu * 21).reshape(-1, 21)
                    assert np.array_equal(_bits(got), _bits(want.reshape(-1, 21))), (w, h, thr, rep, "device")
                got = c.predict_luma(luma, w, h, 1, 27)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (w, h, thr, "pageable")
            d_in.free()
            d_out.free()
            c.free_host_buffers()
    finally:
        c.close()


def test_predict_luma_streamed_input(pkg, oracle):
    """ethcnn_predict_luma_begin / ethcnn_rows_ready / ethcnn_predict_luma_end: one picture's pass is queued on a page-locked buffer a
    filling thread is still writing (the in-process hook's conversion loop) -- CTU rows in scrambled order, some before begin -- and
    is bit-identical to the oracle: single-launch PULL form (<= 2304 CTUs, 16-byte rows), the five-launch path with a waiting tile
    stage (odd width; 3927 CTUs), fast FC1 plan on the latter.  Misuse and the never-reported row as for the LDP entry."""
    import threading
    import time
    e = pkg.ethcnn
    rng = np.random.default_rng(78)
    blob = oracle.synth_blob(10, 4.0)
    c = pkg.EthCnn(0)
    try:
        c.load_blob(blob)
        c.set_thresholds(0.6, 0.4)
        for (w, h) in ((1920, 1080), (416, 240), (200, 136), (3840, 2160), (4928, 3264)):
            nctu, nrows = e.ctus_per_frame(w, h), (h + 63) // 64
            pin = c.host_buffer(w * h)
            pprobs = c.host_buffer(nctu * 84).view(np.float32)
            for rep in range(3):
                luma = rng.integers(0, 256, size=(1, h, w), dtype=np.uint8)
                want = oracle.predict_frames(blob, luma, w, h, 1, 30, 0.6, 0.4, mode=0)
                order = rng.permutation(nrows)
                early = order[:rep]
                pin[:] = 0x55
                def put(cy):
                    pin[cy * 64 * w:min(h, cy * 64 + 64) * w] = luma[0, cy * 64:cy * 64 + 64].reshape(-1)
                    c.rows_ready(cy, cy + 1)
                for cy in early:
                    put(int(cy))
                def filler():
                    time.sleep(0.001)
                    for cy in order[len(early):]:
                        put(int(cy))
                t = threading.Thread(target=filler)
                t.start()
                c.predict_luma_begin(pin, w, h, 30, pprobs)
                t.join()
                c.predict_luma_end()
                assert np.array_equal(_bits(pprobs.reshape(-1, 21)), _bits(want.reshape(-1, 21))), (w, h, rep)
                # the plain call on the same context in between
                assert np.array_equal(_bits(c.predict_luma(luma, w, h, 1, 30).reshape(-1, 21)), _bits(want.reshape(-1, 21))), (w, h, rep)
            c.free_host_buffers()
        # misuse
        w, h = 416, 240
        nctu, nrows = e.ctus_per_frame(w, h), 4
        pin = c.host_buffer(w * h)
        pprobs = c.host_buffer(nctu * 84).view(np.float32)
        with pytest.raises(e.EthCnnError):
            c.predict_luma_begin(np.zeros(w * h, np.uint8), w, h, 30, pprobs)   # pageable
        with pytest.raises(e.EthCnnError):
            c.predict_luma_end()                                                 # nothing begun
        big = c.host_buffer(64 * 64 * 8192)
        with pytest.raises(e.EthCnnError):
            c.predict_luma_begin(big, 64 * 8192, 64, 30, np.zeros(8192 * 21, np.float32))  # 8192 CTUs: not one pass
        # rows reported AHEAD of a begin that then fails are forgotten: the next streamed picture waits for its own rows
        luma = rng.integers(0, 256, size=(1, h, w), dtype=np.uint8)
        want = oracle.predict_frames(blob, luma, w, h, 1, 30, 0.6, 0.4, mode=0)
        c.rows_ready(0, nrows)
        with pytest.raises(e.EthCnnError):
            c.predict_luma_begin(np.zeros(w * h, np.uint8), w, h, 30, pprobs)
        pin[:] = 0
        def late():
            time.sleep(0.003)
            pin[:] = luma.reshape(-1)
            c.rows_ready(0, nrows)
        t = threading.Thread(target=late)
        t.start()
        c.predict_luma_begin(pin, w, h, 30, pprobs)
        t.join()
        c.predict_luma_end()
        assert np.array_equal(_bits(pprobs.reshape(-1, 21)), _bits(want.reshape(-1, 21)))
        c.predict_luma_begin(pin, w, h, 30, pprobs)
        with pytest.raises(e.EthCnnError):
            c.predict_luma_begin(pin, w, h, 30, pprobs)                          # still open
        c.rows_ready(0, nrows)
        c.predict_luma_end()
        # a row that never comes: the kernels give up after ~1 s, the context keeps working
        luma = rng.integers(0, 256, size=(1, h, w), dtype=np.uint8)
        pin[:] = luma.reshape(-1)
        c.predict_luma_begin(pin, w, h, 30, pprobs)
        c.rows_ready(1, nrows)
        t0 = time.time()
        with pytest.raises(e.EthCnnError, match="never reported"):
            c.predict_luma_end()
        assert 0.5 < time.time() - t0 < 10.0
        want = oracle.predict_frames(blob, luma, w, h, 1, 30, 0.6, 0.4, mode=0)
        c.rows_ready(0, nrows)
        c.predict_luma_begin(pin, w, h, 30, pprobs)
        c.predict_luma_end()
        assert np.array_equal(_bits(pprobs.reshape(-1, 21)), _bits(want.reshape(-1, 21)))
    finally:
        c.close()


def test_claim_or_execute_when_producer_blocks_never_run(pkg, oracle):
    """FORWARD PROGRESS of the dataflow launch when the GPU is shared: a consumer that has waited too long executes
    the unclaimed work items it depends on itself.  ETHCNN_SMALL_STEAL_TEST=k makes every k-th producer block (trunk and FC1
    alike) leave WITHOUT claiming its item -- as if it had never been given a slot -- and gives consumers no patience, so FC1
    blocks must run trunk items and heads blocks must run FC1 items (which run trunk items).  Results stay bit-exact.  Run in
    a subprocess (the knob is read once per process)."""
    import os
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import importlib, os, sys
        import numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
        import ethcnn_np as oracle
        pkg = importlib.import_module("hevc-complexity-reduction_amd")
        rng = np.random.default_rng(5)
        blob = oracle.synth_blob(6, 4.0)
        c = pkg.EthCnn(0)
        c.load_blob(blob)
        for (w, h, frames) in ((1920, 1080, 1), (768, 512, 1), (416, 240, 3), (3840, 2160, 1), (64, 64, 1)):
            luma = rng.integers(0, 256, size=(frames, h, w), dtype=np.uint8)
            for thr in ((0.5, 0.5), (0.9, 0.5), (2.0, -0.5)):
                c.set_thresholds(*thr)
                want = oracle.predict_frames(blob, luma, w, h, frames, 30, thr[0], thr[1], mode=0)
                for rep in range(3):
                    got = c.predict_luma(luma, w, h, frames, 30)
                    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (w, h, thr, rep)
                if frames == 1:  # page-locked picture: the PULL form at every size (its pull blocks are producers too)
                    pin = c.host_buffer(w * h)
                    pin[:] = luma.reshape(-1)
                    got = c.predict_luma(pin.reshape(1, h, w), w, h, 1, 30)
                    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (w, h, thr, "pull")
                    c.free_host_buffers()
            want_vec = oracle.resi_vectors(blob, luma[0], w, h, mode=0)
            assert np.array_equal(c.resi_vectors(luma[0], w, h).view(np.uint32), want_vec.view(np.uint32)), (w, h)
        print("steal ok")
    """ % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    for k in ("2", "3", "7"):
        from conftest import exp_env  # the knob exists in the experiments build only
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                           env=exp_env(ETHCNN_SMALL_STEAL_TEST=k))
        assert r.returncode == 0 and "steal ok" in r.stdout, (k, r.stdout[-800:], r.stderr[-1500:])


def test_four_processes_share_the_gpu(pkg, oracle, tmp_path):
    """The scenario that broke the first form of the dataflow launch: several PROCESSES issuing single-picture passes on one GPU
    at the same time (their launches' waiting blocks can starve each other's producers; claim-or-execute keeps them moving).
    Four processes that start together (file barrier) and each issue single-picture passes of three geometries back to back for
    3 s (plus a 12,240-CTU pass now and then, whose long-running blocks skew the XCDs' dispatch progress), every 8th result
    checked bit for bit; none may trap, hang, stall or fall silent.  Two of the four alternate with the PULL form of the launch
    (page-locked pictures, host to host: pull blocks -> trunk items that wait for them), every result checked."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import importlib, os, sys, time
        import numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
        import ethcnn_np as oracle
        pkg = importlib.import_module("hevc-complexity-reduction_amd")
        seed, gate = int(sys.argv[1]), sys.argv[2]
        rng = np.random.default_rng(seed)
        blob = oracle.synth_blob(6, 4.0)
        c = pkg.EthCnn(0)
        c.load_blob(blob)
        cases = []
        for (w, h) in ((1920, 1080), (768, 512), (3840, 2160)):
            luma = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
            nctu = pkg.ethcnn.ctus_per_frame(w, h)
            d_in, d_out = c.alloc(luma.nbytes), c.alloc(nctu * 84)
            d_in.upload(luma)
            pin = c.host_buffer(w * h)   # ... and page-locked: the PULL form of the same launch (odd processes use it every other round)
            pin[:] = luma.reshape(-1)
            cases.append((w, h, nctu, d_in, d_out, oracle.predict_frames(blob, luma, w, h, 1, 30, 0.5, 0.5, mode=0), pin))
        open(os.path.join(gate, "ready%%d" %% seed), "w").close()
        while len(os.listdir(gate)) < 4:
            time.sleep(0.001)
        t0, k = time.time(), 0
        big_in, big_out = c.alloc(24 * 1080 * 1920), c.alloc(24 * 510 * 84)   # a many-CTU pass now and then: its long-running
        big_in.upload(rng.integers(0, 256, size=24 * 1080 * 1920, dtype=np.uint8))  # blocks skew the XCDs' dispatch progress
        while time.time() - t0 < float(os.environ.get('ETHCNN_SHARED_SECONDS', '3')) or k %% 8:
            if k %% 32 == 8 * seed:
                c.predict_luma_device(big_in, 1920, 1080, 24, 30, big_out)
            w, h, nctu, d_in, d_out, want, pin = cases[(k // 8 + seed) %% 3]
            if seed %% 2 == 1 and (k // 8) %% 2 == 1:  # host -> host through the pull blocks (synchronous call)
                got = c.predict_luma(pin.reshape(1, h, w), w, h, 1, 30).reshape(-1, 21)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (seed, k, "pull")
                k += 1
                continue
            c.predict_luma_device(d_in, w, h, 1, 30, d_out)
            if k %% 8 == 7:
                c.synchronize()
                got = d_out.download(np.float32, nctu * 21).reshape(-1, 21)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (seed, k)
            k += 1
        c.synchronize()
        print("ok %%d %%d calls %%.2f s" %% (seed, k, time.time() - t0))
    """ % (root, root))
    gate = tmp_path / "gate"
    gate.mkdir()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(s), str(gate)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for s in range(4)]
    for s, p in enumerate(procs):
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0 and ("ok %d" % s) in out, (s, out[-500:], err[-1500:])
        print(out.strip())
        assert int(out.split()[2]) >= 400, out      # sharing the GPU four ways still leaves > 250 calls a second each
        # the launch without claim-or-execute (A/B build -DSMALL_NO_STEAL) passes the value checks but in 4 of 6 runs STALLS here for
        # 4-17 s (until the driver's queue preemption unties the processes) or traps (profiles/r03_shared_gpu_ab.txt)
        assert float(out.split()[4]) < 2.0 + float(os.environ.get('ETHCNN_SHARED_SECONDS', '3')), out


def test_completion_word_only_covers_the_last_launch(pkg, oracle):
    """ethcnn_synchronize returns through the completion word of a single-launch pass (a word in page-locked memory stored by the
    launch's last block, ~5 us sooner than hipStreamSynchronize) ONLY when that pass is the last thing enqueued: a long
    pipelined call or an LDP front-end enqueued behind it must be waited for the ordinary way.  Outputs are poisoned before
    every round, so a synchronize that returned early would be seen."""
    rng = np.random.default_rng(31)
    blob = oracle.synth_blob(4, 4.0)
    c = pkg.EthCnn(device=0)
    c.load_blob(blob)
    try:
        w, h = 1920, 1080
        small = _luma(rng, 1, h, w, w)
        big = _luma(rng, 24, h, w, w)
        n1 = pkg.ethcnn.ctus_per_frame(w, h)
        want_small = oracle.predict_frames(blob, small, w, h, 1, 30, 0.5, 0.5, mode=0)
        want_big = oracle.predict_frames(blob, big, w, h, 24, 30, 0.5, 0.5, mode=0)
        want_vec = oracle.resi_vectors(blob, small[0], w, h, mode=0)
        d_s, d_b = c.alloc(small.nbytes), c.alloc(big.nbytes)
        o_s, o_b, o_v = c.alloc(want_small.nbytes), c.alloc(want_big.nbytes), c.alloc(want_vec.nbytes)
        d_s.upload(small)
        d_b.upload(big)
        poison_b = np.full(want_big.size, 7.0, np.float32)
        poison_v = np.full(want_vec.size, 7.0, np.float32)
        for rep in range(6):
            o_b.upload(poison_b)
            o_v.upload(poison_v)
            c.predict_luma_device(d_s, w, h, 1, 30, o_s)            # carries the completion word ...
            if rep % 2 == 0:
                c.predict_luma_device(d_b, w, h, 24, 30, o_b)       # ... but 12,240 CTUs of five-launch passes follow
            else:
                c.resi_vectors_device(d_s, w, h, o_v)               # ... but an LDP front-end (no word) follows
            c.synchronize()
            if rep % 2 == 0:
                got = o_b.download(np.float32, want_big.size).reshape(want_big.shape)
                assert np.array_equal(_bits(got), _bits(want_big)), rep
            else:
                got = o_v.download(np.float32, want_vec.size).reshape(want_vec.shape)
                assert np.array_equal(_bits(got), _bits(want_vec)), rep
            got = o_s.download(np.float32, want_small.size).reshape(want_small.shape)
            assert np.array_equal(_bits(got), _bits(want_small)), rep
            # and the plain case: the pass alone, many times (the word's sequence number moves on every launch)
            for _ in range(20):
                c.predict_luma_device(d_s, w, h, 1, 30, o_s)
                c.synchronize()
            got = o_s.download(np.float32, want_small.size).reshape(want_small.shape)
            assert np.array_equal(_bits(got), _bits(want_small)), rep
        for b in (d_s, d_b, o_s, o_b, o_v):
            b.free()
    finally:
        c.close()


def test_streamed_staging_copy_that_comes_too_late_is_rerun_not_returned(pkg, oracle):
    """a pageable picture takes the streamed-staging form of ethcnn_predict_luma -- the single-launch pass is
    queued on the page-locked staging buffer FIRST, then the rows are copied in and reported.  If the calling thread is held for
    more than ~1 s between the two (SIGSTOP, a VM pause), the kernels give up waiting and compute on stale staging contents; the
    call must notice (the gave-up word) and run the pass again on the now complete buffer instead of returning ETHCNN_OK with wrong
    numbers.  ETHCNN_TEST_STAGE_STALL_MS (experiments build) holds the thread there."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import importlib, os, sys, time
        import numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
        import ethcnn_np as oracle
        pkg = importlib.import_module("hevc-complexity-reduction_amd")
        rng = np.random.default_rng(15)
        blob = oracle.synth_blob(6, 8.0)
        c = pkg.EthCnn(0)
        c.load_blob(blob)
        c.set_thresholds(0.5, 0.5)
        w, h = 832, 480
        for k in range(2):  # the second picture meets a staging buffer that holds the FIRST one: stale data would be a plausible-looking wrong answer
            luma = rng.integers(0, 256, size=(1, h, w), dtype=np.uint8)
            want = oracle.predict_frames(blob, luma, w, h, 1, 32, 0.5, 0.5, mode=0)
            t0 = time.time()
            got = c.predict_luma(luma, w, h, 1, 32)
            dt = time.time() - t0
            assert dt > 1.2, dt  # the stall really happened on this path
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), k
        print("rerun ok")
    """ % (root, root))
    from conftest import exp_env
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=exp_env(ETHCNN_TEST_STAGE_STALL_MS=1300))
    assert r.returncode == 0 and "rerun ok" in r.stdout, (r.stdout[-800:], r.stderr[-1500:])
