// ethcnn_pass.cpp -- one pass over CTUs resident in HBM: the kernel pipeline of a pass, pass planning, ethcnn_predict_luma_device
#include "ethcnn_ctx.h"

// ------------------------------------------------------------------- pipeline -------
int make_geom(ethcnn_ctx* c, int w, int h, ptrdiff_t pitch, ptrdiff_t fstride, FrameGeom* g) {
    if (w <= 0 || h <= 0) return set_err(c, ETHCNN_ERR_ARG, "bad frame size %dx%d", w, h);
    if (pitch < w) return set_err(c, ETHCNN_ERR_ARG, "pitch %td < width %d", pitch, w);
    g->width = w;
    g->height = h;
    g->pitch = (long)pitch;
    g->frame_stride = (long)fstride;
    g->cw = (w + 63) / 64;
    g->ch = (h + 63) / 64;
    g->nctu = g->cw * g->ch;
    return 0;
}

// Everything a pass leaves in flight ends on the main stream (its tile stage is always followed by its own trunk there), so
// "after all passes enqueued so far" is simply main-stream order.  Main-stream users of the workspace outside run_pass (LDP
// front-end, LSTM step) only have to tell the NEXT pipelined tile stage, which runs on the side stream, to wait for them:
// ---- completion word.  done_arm: number for a launch whose last block will store it; the caller sets c->done_armed once the
// launch is enqueued.  Every other enqueue on the main stream clears done_armed first (the word would not cover it).
unsigned done_arm(ethcnn_ctx* c) {
    if (!c->done_sync || !c->h_done) return 0;
    if (++c->done_seq == 0) ++c->done_seq;
    return c->done_seq;
}
// wait for everything enqueued on the main stream: through the completion word when the last enqueued launch carries one
// (bounded: a launch that never reports -- a device fault -- falls through to hipStreamSynchronize, which returns the error)
hipError_t stream_sync(ethcnn_ctx* c) {
    const unsigned seq = c->done_armed;
    c->done_armed = 0;
    if (seq) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 1;; ++spins) {
            if (__atomic_load_n(c->h_done, __ATOMIC_ACQUIRE) == seq) return hipSuccess;
#if defined(__SSE2__)
            _mm_pause();
#endif
            if ((spins & 4095u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
        }
    }
    return hipStreamSynchronize(c->stream);
}

int serial_end(ethcnn_ctx* c) {
    c->main_dirty = true;  // the next pipelined tile stage records e_main behind this work and waits for it
    return 0;
}

// the single-launch form of a small pass (ethcnn_small.hip); fc1_out: ws.h1 (All-Intra) or the caller's vectors (resi).
// Asynchronous on the main stream.
int run_small_pass(ethcnn_ctx* c, const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, bool resi, const Workspace& w,
                   float* fc1_out, float qn, float* d_probs, int nchunks, bool pull, const unsigned* wait_rows) {
    const int words = small_pass_sync_words(n, nchunks);
    c->done_armed = 0;
    if (c->small_epoch >= (1 << 30)) c->ssync_clean = false;  // tags start over on a freshly zeroed area
    if (words > c->ssync_cap || !c->ssync_clean) {
        if (words > c->ssync_cap) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->d_ssync) (void)hipFree(c->d_ssync);
            c->d_ssync = nullptr;
            c->ssync_cap = 0;
            const int cap = words;  // fixed size (~0.5 MB), once
            HIPCHK(c, hipMalloc((void**)&c->d_ssync, (size_t)cap * sizeof(int)));
            c->ssync_cap = cap;
        }
        HIPCHK(c, hipMemsetAsync(c->d_ssync, 0, (size_t)c->ssync_cap * sizeof(int), c->stream));  // stream-ordered
        c->small_epoch = 0;
    }
    ++c->small_epoch;
    c->ssync_clean = false;  // until this launch has been enqueued without an error
    (void)hipGetLastError();
    const unsigned seq = resi ? 0u : done_arm(c);
    { StageTimer t(c, ETHCNN_STAGE_FC1, n); launch_small_pass(d_luma, g, ctu0, n, resi, w, c->dw, fc1_out, qn, c->thr1, c->thr2, d_probs, nchunks, c->d_ssync, c->small_epoch, seq ? c->h_done : nullptr, seq, c->stream, pull, wait_rows, c->rows_seq, c->h_done + 1, (seq && !resi) ? c->host_probs : nullptr); }
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return set_err(c, ETHCNN_ERR_DEVICE, "launch of the single-launch small pass failed: %s", hipGetErrorString(le));
    c->ssync_clean = true;
    c->done_armed = seq;
    c->host_probs_used = seq && !resi && c->host_probs != nullptr && reinterpret_cast<uintptr_t>(c->host_probs) % 16 == 0;
    return 0;
}

// one pass over CTUs [ctu0, ctu0+n) of the sequence; ctu0 is sub-batch aligned.  input_ready: event after which d_luma
// may be read (nullptr: the caller ordered it before the call).  Asynchronous; the pass ends on the main stream.
int run_pass(ethcnn_ctx* c, const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, int qp, float* d_probs_pass,
             hipEvent_t input_ready) {
    const int cpf = chunks_per_frame(g.nctu);
    const long nchunks = (ctu0 + n - 1) / g.nctu * cpf + ((ctu0 + n - 1) % g.nctu) / kSubBatch + 1 -
                         (ctu0 / g.nctu * cpf + (ctu0 % g.nctu) / kSubBatch);
    c->done_armed = 0;
    // the 16-bit plans' weight images + their accuracy guard: before anything of THIS pass is enqueued (the guard's measured stage runs
    // two passes of its own through this workspace).  Not for a pass that will run as the single-launch small pass: that one always
    // computes exactly, whatever the plan (and may be a STREAMED picture whose rows are not in memory yet)
    const bool takes_small = c->small_launch && n < kPipelineMinCtus && small_pass_ok(d_luma, g, n);
    int rc = (c->fc1_plan && !takes_small) ? ensure_fast_weights(c, c->fc1_plan) : 0;
    if (rc) return rc;
    rc = ensure_workspace(c, n, (int)nchunks);
    if (rc) return rc;
    const float qn = (float)qp * (1.0f / 51.0f);  // net_CNN.py:106
    // Small passes (a frame or a few: the in-process encoder hook, the LDP-sized calls) stay on one stream: there is no FC1 of
    // a previous pass long enough to hide anything under, and the cross-stream event costs ~10 us of a 75 us call
    // Plan 3 (round 5): the CTU-load stage is folded into the trunk's S branch (k1_trunk_f16_fold) -- no tile launch, nothing for a
    // side stream to run.  (Experiments build: ETHCNN_PLAN3_FOLD=0 keeps round 4's tile stage beside FC1 for the A/B.)
    // 2 (default): the whole trunk behind one pass over the frames (k1_trunk_f16_foldall); 1: S branch folded, M / L as a second launch
    static const int fold3_knob = [] { const char* e = dev_env("ETHCNN_PLAN3_FOLD"); return e ? std::atoi(e) : 2; }();
    const bool fold3 = c->fc1_plan == 3 && fold3_knob != 0 && c->tile_wait_rows == nullptr;
    const bool side_tile = c->overlap != 0 && n >= kPipelineMinCtus && !fold3;
    const int p = side_tile ? (int)(c->pass_idx++ & 1) : 0;
    const Workspace w = ws_view(c, p);
    if (side_tile && !c->s_tile && (rc = ensure_side_streams(c)) != 0) return rc;
    hipStream_t s_tile = side_tile ? c->s_tile : c->stream;
    if (input_ready) HIPCHK(c, hipStreamWaitEvent(s_tile, input_ready, 0));
    if (!side_tile && c->small_launch && small_pass_ok(d_luma, g, n)) {
        // one picture (the in-process encoder hook, the reference's own 768x512 case): CTU load + trunk -> FC1 -> heads -> gates
        // as ONE launch instead of five dependent ones
        Workspace wv = w;
        if (!c->debug_capture) wv.h2 = wv.logits = wv.raw = nullptr;
        rc = run_small_pass(c, d_luma, g, ctu0, n, false, wv, w.h1, qn, d_probs_pass, (int)nchunks, c->luma_over_pcie,
                            c->luma_over_pcie ? c->tile_wait_rows : nullptr);
        if (rc) return rc;
        c->main_dirty = true;  // a later pipelined tile stage must wait for this pass
        c->times.ctus += n;
        c->last_n = n;
        c->last_parity = p;
        c->last_fast = 0;  // (the single-launch pass always computes FC1 exactly)
        return 0;
    }
    if (side_tile) {
        // tile(i) overwrites the tile outputs and gate flags of buffer set p: last read by trunk(i-2) / gate(i-2).  Both are
        // ordered before trunk(i-1) on the main stream, so the wait for e_trunk[p ^ 1] below covers them; a main-stream
        // pass in between (small pass, LDP call) records e_main behind its last kernel instead.  Every event RECORD on the
        // main stream is a barrier packet between two kernels (~7 us of idle GPU, rocprofv3 kernel trace): there is exactly
        // one per pipelined pass (e_trunk).  A never-recorded event is a no-op.
        HIPCHK(c, hipStreamWaitEvent(s_tile, c->e_trunk[p], 0));
        if (c->main_dirty) {  // main-stream users of the workspace since the last pipelined pass (small passes, LDP steps)
            HIPCHK(c, hipEventRecord(c->e_main, c->stream));
            c->main_dirty = false;
        }
        HIPCHK(c, hipStreamWaitEvent(s_tile, c->e_main, 0));
        // ... and it should run beside FC1(i-1), not beside trunk(i-1): with the trunk it competes for VALU issue and HBM
        // (measured: trunk 556 -> 819 us, tile 180 -> 511 us, step period 2.60 -> 2.73 ms; profiles/r02_overlap_trace.txt)
        const bool behind_fc1 = c->tile_after_fc1 != 0 && c->fc1_plan != 0;
        HIPCHK(c, hipStreamWaitEvent(s_tile, behind_fc1 ? c->e_fc1[p ^ 1] : c->e_trunk[p ^ 1], 0));
    }
    // A/B knob (experiments build): CTU-load stage folded into the trunk for big exact passes (profiles/r04_tile_fold.txt)
    static const bool fold_knob = [] { const char* e = dev_env("ETHCNN_TILE_FOLD"); return e && std::atoi(e) != 0; }();
    const bool fold = fold_knob && c->fc1_plan == 0 && (g.width % 16 == 0) && (g.pitch % 16 == 0) && (g.frame_stride % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(d_luma) % 16 == 0);
    (void)hipGetLastError();  // launch errors below are reported per stage; drop anything stale first
#define LAUNCH_OK(name)                                                                                            \
    do {                                                                                                           \
        const hipError_t le_ = hipGetLastError();                                                                  \
        if (le_ != hipSuccess) return set_err(c, ETHCNN_ERR_DEVICE, "launch of the %s stage failed: %s", name, hipGetErrorString(le_)); \
    } while (0)
    // experiments build only (scripts/plan3_power.py): ONE stage of the pass, launched alone over and over (outputs meaningless) -- socket
    // power and shader clock of each stage at steady state.  1 trunk (+ CTU-load stage), 2 FC1, 3 heads + gate
    // (the first three passes of the process run whole: the stage then loops on the REAL features / h1 they left -- FC1's power depends on its data)
    static const int only_knob = [] { const char* e = dev_env("ETHCNN_STAGE_ONLY"); return e ? std::atoi(e) : 0; }();
    static int passes_seen = 0;
    const int only = (only_knob != 0 && ++passes_seen > 3) ? only_knob : 0;
    // the tile stage also zeroes the pass's gate predicates
    if (fold3 || (only != 0 && only != 1)) {
        // (no tile launch; the folded trunk below clears the sync area itself)
    } else if (fold) {  // no tile launch: only the pass's sync area is cleared (what the tile stage does on the way)
        HIPCHK(c, hipMemsetAsync(w.flags, 0, (size_t)sync_words((int)nchunks) * sizeof(int), s_tile));
    } else {
        StageTimer t(c, ETHCNN_STAGE_TILE, n, s_tile);
        launch_tile(d_luma, g, ctu0, n, w, sync_words((int)nchunks), s_tile, side_tile ? c->tile_blocks : 0,
                    side_tile ? nullptr : c->tile_wait_rows, c->rows_seq, c->h_done + 1);  // (streamed input: a single main-stream pass)
    }
    LAUNCH_OK("tile");
    if (side_tile) {
        HIPCHK(c, hipEventRecord(c->e_tile[p], s_tile));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->e_tile[p], 0));
    }
    const int fast = c->fc1_plan;  // plans 2 / 3: trunk -> 16-bit feature pieces -> FC1 on the 16-bit matrix pipe
    if (only == 0 || only == 1)
    { StageTimer t(c, ETHCNN_STAGE_TRUNK);
      if (fold) launch_trunk_direct(d_luma, g, ctu0, w, c->dw, n, c->stream);
      else if (fold3 && fold3_knob == 1) {
          launch_trunk_f16_fold(d_luma, g, ctu0, n, w, c->dw, sync_words((int)nchunks), c->stream);
          launch_trunk_f16(w, c->dw, n, c->stream, /*ml_only=*/true);
      } else if (fold3) {
          static const int bpc = [] { const char* e = dev_env("ETHCNN_PLAN3_FOLD_BLOCKS"); return e ? std::atoi(e) : 2; }();
          launch_trunk_f16_foldall(d_luma, g, ctu0, n, w, c->dw, sync_words((int)nchunks), c->stream, bpc);
      } else if (fast == 3) launch_trunk_f16(w, c->dw, n, c->stream);
      else launch_trunk(w, c->dw, n, false, c->stream, fast); }
    LAUNCH_OK("trunk");
    if (side_tile) HIPCHK(c, hipEventRecord(c->e_trunk[p], c->stream));
    Workspace wv = w;
    if (!c->debug_capture) wv.h2 = wv.logits = wv.raw = nullptr;
    if (fast) {
        if (only == 0 || only == 2) { StageTimer t(c, ETHCNN_STAGE_FC1, n); launch_fc1_fast(w, c->dw, n, w.h1, fast == 3 ? 2 : fast, c->stream, c->cus); }
        LAUNCH_OK("FC1 (16-bit pipe)");
        if (side_tile && c->tile_after_fc1) HIPCHK(c, hipEventRecord(c->e_fc1[p], c->stream));
        // plan 3: the heads on the 16-bit pipe as well (experiments build: ETHCNN_PLAN3_HEADS=0 keeps the exact heads for the A/B)
        static const bool heads16_knob = [] { const char* e = dev_env("ETHCNN_PLAN3_HEADS"); return !e || std::atoi(e) != 0; }();
        if (only == 0 || only == 3) { StageTimer t(c, ETHCNN_STAGE_HEADS);
          if (fast == 3 && heads16_knob && c->dw.heads16_w)
              launch_heads_f16(wv, c->dw, n, qn, g.nctu, ctu0, c->thr1, c->thr2, d_probs_pass, c->stream);
          else launch_heads(wv, c->dw, n, qn, g.nctu, ctu0, c->thr1, c->thr2, d_probs_pass, c->stream); }
    } else {
        if (only == 0 || only == 2) { StageTimer t(c, ETHCNN_STAGE_FC1, n); launch_fc1(w, c->dw, n, w.h1, c->stream); }
        LAUNCH_OK("FC1");
        if (only == 0 || only == 3) { StageTimer t(c, ETHCNN_STAGE_HEADS); launch_heads(wv, c->dw, n, qn, g.nctu, ctu0, c->thr1, c->thr2, d_probs_pass, c->stream); }
    }
    if (only == 0 || only == 3) { StageTimer t(c, ETHCNN_STAGE_GATE); launch_gate(w, n, g.nctu, ctu0, c->thr2, d_probs_pass, c->stream); }
    LAUNCH_OK("heads / gate");
#undef LAUNCH_OK
    if (!side_tile) c->main_dirty = true;  // a later pipelined tile stage must wait for this pass
    c->times.ctus += n;
    c->last_n = n;
    c->last_parity = p;
    c->last_fast = fast;
    return 0;
}

// split `total` CTUs (nframes * nctu) into passes: whole frames when a frame fits the
// workspace, otherwise sub-batch-aligned pieces of one frame (gate scope stays intact).
std::vector<Pass> plan_passes(int nctu, int nframes, int max_ctus) {
    std::vector<Pass> out;
    if (nctu <= max_ctus) {
        const int fpp = std::max(1, max_ctus / nctu);
        for (int f = 0; f < nframes; f += fpp) {
            const int nf = std::min(fpp, nframes - f);
            out.push_back({(long)f * nctu, nf * nctu});
        }
    } else {
        for (int f = 0; f < nframes; ++f)
            for (int o = 0; o < nctu; o += max_ctus) out.push_back({(long)f * nctu + o, std::min(max_ctus, nctu - o)});
    }
    return out;
}

// ---- the measured stage of the 16-bit plans' accuracy guard (ethcnn_model.cpp::check_fast_plan).  A seeded picture of 2048 x 1280
// (32 x 20 CTUs) made of 128 x 128 macro tiles of six kinds, so that large and tiny activations both occur (the floors of the split
// pieces are ABSOLUTE: they show where values are small); both plans through the multi-launch path at QP 22 and at QP 37, gates open
// (every output compared).
static void calibration_picture(std::vector<uint8_t>& luma, int w, int h) {
    luma.resize((size_t)w * h);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 32); };
    for (int ty = 0; ty < h / 128; ++ty)
        for (int tx = 0; tx < w / 128; ++tx) {
            const int kind = (tx + 5 * ty) % 6, base = 16 + (int)(rnd() % 208), amp = 1 + (int)(rnd() % 6), per = 4 << (rnd() % 4);
            for (int y = 0; y < 128; ++y)
                for (int x = 0; x < 128; ++x) {
                    int v;
                    switch (kind) {
                        case 0: v = base; break;                                                         // flat
                        case 1: v = base + (int)(rnd() % (2 * amp + 1)) - amp; break;                    // low contrast: a few grey levels of noise
                        case 2: v = (x * 2 + y) % 256; break;                                            // gradient
                        case 3: v = (int)(rnd() & 255); break;                                           // full-range noise
                        case 4: v = ((x / per) & 1) ? 235 : 20; break;                                   // edges
                        default: { const int a = x % (2 * per), b = y % (2 * per);                       // smooth texture (triangle waves)
                                   v = 128 + (a < per ? a : 2 * per - a) * 96 / per - 48 + (b < per ? b : 2 * per - b) * 32 / per - 16; }
                    }
                    luma[(size_t)(ty * 128 + y) * w + tx * 128 + x] = (uint8_t)std::min(255, std::max(0, v));
                }
        }
}
int calibrate_fast_plan(ethcnn_ctx* c, int plan, double* max_abs) {
    constexpr int W = 2048, H = 1280;
    constexpr int kQps[2] = {22, 37};  // both ends of the QP range the four checkpoints cover (the QP column of FC2 / FC3 is part of the plan-3 heads)
    FrameGeom g;
    int rc = make_geom(c, W, H, W, (ptrdiff_t)W * H, &g);
    if (rc) return rc;
    std::vector<uint8_t> luma;
    calibration_picture(luma, W, H);
    const size_t pf = (size_t)g.nctu * kNOut;
    uint8_t* d_luma = nullptr;
    float* d_p = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_luma, luma.size()));
    if (hipMalloc((void**)&d_p, 4 * pf * 4) != hipSuccess) { (void)hipFree(d_luma); return set_err(c, ETHCNN_ERR_NOMEM, "calibration: out of device memory"); }
    std::vector<float> p(4 * pf);  // [qp][exact | plan]
    const float t1 = c->thr1, t2 = c->thr2;
    const int plan0 = c->fc1_plan, small0 = c->small_launch;
    const bool cap0 = c->debug_capture, pcie0 = c->luma_over_pcie;
    const unsigned* const wait0 = c->tile_wait_rows;  // (the caller may be a streamed picture's pass: the calibration picture is complete)
    float* const hp0 = c->host_probs;
    c->thr1 = c->thr2 = -1.0f;  // gates open: all 21 outputs of every CTU are compared
    c->small_launch = 0;        // the plans are forms of the multi-launch path
    c->debug_capture = false;
    c->tile_wait_rows = nullptr;
    c->luma_over_pcie = false;
    c->host_probs = nullptr;
    hipError_t e = hipMemcpyAsync(d_luma, luma.data(), luma.size(), hipMemcpyHostToDevice, c->stream);
    for (int k = 0; k < 4 && rc == 0 && e == hipSuccess; ++k) {
        c->fc1_plan = (k & 1) == 0 ? 0 : plan;
        rc = run_pass(c, d_luma, g, 0, g.nctu, kQps[k >> 1], d_p + k * pf);
    }
    c->thr1 = t1; c->thr2 = t2; c->fc1_plan = plan0; c->small_launch = small0; c->debug_capture = cap0;
    c->tile_wait_rows = wait0; c->luma_over_pcie = pcie0; c->host_probs = hp0;
    if (rc == 0 && e == hipSuccess) e = hipMemcpyAsync(p.data(), d_p, 4 * pf * 4, hipMemcpyDeviceToHost, c->stream);
    const hipError_t e2 = hipStreamSynchronize(c->stream);
    (void)hipFree(d_luma);
    (void)hipFree(d_p);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return set_err(c, ETHCNN_ERR_DEVICE, "calibration of plan %d failed: %s", plan, hipGetErrorString(e != hipSuccess ? e : e2));
    double worst = 0.0;
    for (int q = 0; q < 2; ++q)
        for (size_t i = 0; i < pf; ++i) {
            const double d = std::fabs((double)p[(2 * q) * pf + i] - (double)p[(2 * q + 1) * pf + i]);
            if (!(d <= worst)) worst = d;  // (a NaN sticks)
        }
    *max_abs = worst;
    return ETHCNN_OK;
}

extern "C" int ethcnn_predict_luma_device(ethcnn_ctx* c, const uint8_t* d_luma, int w, int h, ptrdiff_t pitch,
                                          ptrdiff_t fstride, int nframes, int qp, float* d_probs) {
    if (!c || !d_luma || !d_probs || nframes < 0) return c ? set_err(c, ETHCNN_ERR_ARG, "null pointer / negative frame count") : ETHCNN_ERR_ARG;
    if (!c->have_weights) return set_err(c, ETHCNN_ERR_NOWEIGHTS, "no weights loaded");
    FrameGeom g;
    int rc = make_geom(c, w, h, pitch, fstride, &g);
    if (rc) return rc;
    if (nframes == 0) return ETHCNN_OK;
    HIPCHK(c, hipSetDevice(c->device));
    for (const Pass& p : plan_passes(g.nctu, nframes, c->max_ctus)) {
        rc = run_pass(c, d_luma, g, p.ctu0, p.n, qp, d_probs + (size_t)p.ctu0 * kNOut);
        if (rc) return rc;
    }
    return ETHCNN_OK;
}
