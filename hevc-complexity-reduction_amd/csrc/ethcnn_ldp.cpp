// ethcnn_ldp.cpp -- config #5: resi_cnn vectors, one ETH-LSTM step, the per-frame Low-Delay-P calls (resident state, streamed input)
#include "ethcnn_ctx.h"

// -------------------------------------------------------------- config #5 -----------
extern "C" int ethcnn_resi_vectors_device(ethcnn_ctx* c, const uint8_t* d_luma, int w, int h, ptrdiff_t pitch, float* d_vec) {
    if (c) c->done_armed = 0;
    if (!c || !d_luma || !d_vec) return c ? set_err(c, ETHCNN_ERR_ARG, "null pointer") : ETHCNN_ERR_ARG;
    if (!c->have_weights) return set_err(c, ETHCNN_ERR_NOWEIGHTS, "no weights loaded");
    FrameGeom g;
    int rc = make_geom(c, w, h, pitch, (ptrdiff_t)pitch * h, &g);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    for (int o = 0; o < g.nctu; o += c->max_ctus) {
        const int n = std::min(c->max_ctus, g.nctu - o);
        rc = ensure_workspace(c, n, 1);
        if (rc) return rc;
        // one LDP frame: CTU load + trunk -> FC1 as one launch -- for a picture in HBM, and (PULL form) for a page-locked one the
        // caller is still filling (streamed input); a complete page-locked picture keeps the tile-stage launch (launch_small_pass)
        const bool streamed = c->tile_wait_rows != nullptr && c->luma_over_pcie;
        if (c->small_launch && (!c->luma_over_pcie || (streamed && c->pull)) && small_pass_ok(d_luma, g, n)) {
            // (the sync area is laid out before anything is queued: run_small_pass may wait for the stream when it has to be re-zeroed)
            rc = run_small_pass(c, d_luma, g, o, n, true, c->ws, d_vec + (size_t)o * kNVec, 0.0f, nullptr, 1, streamed, streamed ? c->tile_wait_rows : nullptr);
            if (rc) return rc;
            c->times.ctus += n;
            c->last_n = n;
            c->last_parity = 0;
            continue;
        }
        { StageTimer t(c, ETHCNN_STAGE_TILE, n); launch_tile(d_luma, g, o, n, c->ws, 0, c->stream, 0, c->tile_wait_rows, c->rows_seq, c->h_done + 1); }
        { StageTimer t(c, ETHCNN_STAGE_TRUNK); launch_trunk(c->ws, c->dw, n, true, c->stream); }
        { StageTimer t(c, ETHCNN_STAGE_FC1, n); launch_fc1(c->ws, c->dw, n, d_vec + (size_t)o * kNVec, c->stream); }
        HIPCHK(c, hipGetLastError());
        c->times.ctus += n;
        c->last_n = n;
        c->last_parity = 0;
    }
    return serial_end(c);
}

extern "C" int ethcnn_resi_vectors(ethcnn_ctx* c, const uint8_t* luma, int w, int h, ptrdiff_t pitch, float* vec) {
    if (!c || !luma || !vec) return c ? set_err(c, ETHCNN_ERR_ARG, "null pointer") : ETHCNN_ERR_ARG;
    if (w <= 0 || h <= 0 || pitch < w) return set_err(c, ETHCNN_ERR_ARG, "bad geometry");
    const int nctu = ((w + 63) / 64) * ((h + 63) / 64);
    const size_t lbytes = (size_t)(h - 1) * pitch + w;  // the meaningful bytes of a pitched plane: the last row ends at w
    int rc = ensure_staging(c, lbytes, (size_t)nctu * kNVec * 4);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_in[0], luma, lbytes, hipMemcpyHostToDevice, c->stream));
    rc = ethcnn_resi_vectors_device(c, c->d_in[0], w, h, pitch, c->d_out[0]);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(vec, c->d_out[0], (size_t)nctu * kNVec * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ETHCNN_OK;
}

static int ensure_lstm_buffers(ethcnn_ctx* c, int n) {
    if (n <= c->lstm_cap) return ETHCNN_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    void* ptrs[] = {c->d_vec, c->d_state[0], c->d_state[1], c->d_lprobs};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    c->d_vec = c->d_state[0] = c->d_state[1] = c->d_lprobs = nullptr;
    c->lstm_cap = 0;
    const int cap = (n + 15) / 16 * 16;
    HIPCHK(c, hipMalloc((void**)&c->d_vec, (size_t)cap * kNVec * 4));
    HIPCHK(c, hipMalloc((void**)&c->d_state[0], (size_t)cap * 2 * kNVec * 4));
    HIPCHK(c, hipMalloc((void**)&c->d_state[1], (size_t)cap * 2 * kNVec * 4));
    HIPCHK(c, hipMalloc((void**)&c->d_lprobs, (size_t)cap * kNOut * 4));
    c->lstm_cap = cap;
    return ETHCNN_OK;
}

// lstm() x3 + heads + gates on resident vectors: the part of sess.run after resi_cnn
// the sync area of the LSTM launch for frames of n CTUs, zeroed where it has to be (see ethcnn_lstm_step_device).  May WAIT for the
// stream (allocation, re-zeroing): a streamed step calls it before it queues kernels that wait for the caller.
static int ensure_lgate(ethcnn_ctx* c, int n) {
    const int gwords = lstm_frame_words(n);
    if (c->lgate_n != n || c->lstm_epoch >= (1 << 30)) c->lgate_clean = false;
    if (gwords > c->lgate_chunks || !c->lgate_clean) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (gwords > c->lgate_chunks) {
            if (c->d_lgate) (void)hipFree(c->d_lgate);
            c->d_lgate = nullptr;
            c->lgate_chunks = 0;
            HIPCHK(c, hipMalloc((void**)&c->d_lgate, (size_t)gwords * sizeof(int)));
            c->lgate_chunks = gwords;
        }
        HIPCHK(c, hipMemsetAsync(c->d_lgate, 0, (size_t)c->lgate_chunks * sizeof(int), c->stream));  // stream-ordered
        c->lstm_epoch = 0;
        c->lgate_n = n;
        c->lgate_clean = true;  // (zero and laid out for n: the launch below marks it dirty until it has been enqueued)
    }
    return 0;
}

extern "C" int ethcnn_lstm_step_device(ethcnn_ctx* c, const float* d_vec, const float* d_state_in, int n, int qp,
                                       int i_frame, float* d_state_out, float* d_probs) {
    if (!c || !d_vec || !d_state_out || !d_probs) return c ? set_err(c, ETHCNN_ERR_ARG, "null pointer") : ETHCNN_ERR_ARG;
    c->done_armed = 0;
    if (n <= 0) return set_err(c, ETHCNN_ERR_ARG, "n must be positive");
    if (!c->have_lstm) return set_err(c, ETHCNN_ERR_NOWEIGHTS, "no LSTM weights loaded");
    HIPCHK(c, hipSetDevice(c->device));
    const int chunks = (n + kSubBatch - 1) / kSubBatch;
    int rc = ensure_workspace(c, std::min(n, c->max_ctus), chunks);
    if (rc) return rc;
    if (n > c->ws.cap) return set_err(c, ETHCNN_ERR_ARG, "frame of %d CTUs exceeds max_ctus_per_pass", n);
    // gate predicates + ticket tree of the LSTM heads launch: zero between launches by construction (every word is reset by its
    // last user); (re)established here after an allocation or after any failure on this path
    // (the one-launch frame kernel keeps its counters, flags and claim words behind them; a claim word holds the tag of the last
    // launch that claimed it, so the area is zeroed again whenever the frame size -- and with it the layout -- changes, and
    // before the tags wrap)
    rc = ensure_lgate(c, n);
    if (rc) return rc;
    ++c->lstm_epoch;
    c->lgate_clean = false;  // until this launch has been enqueued without an error
    const unsigned seq = done_arm(c);
    {
        StageTimer t(c, ETHCNN_STAGE_HEADS);
        launch_lstm(d_vec, d_state_in, d_state_out, c->d_lstm, n, qp, i_frame, c->thr1, c->thr2, c->debug_capture ? c->ws.raw : nullptr,
                    d_probs, c->d_lgate, seq ? c->h_done : nullptr, seq, (c->lstm_one_launch && n <= kLstmOneLaunchMaxCtus) ? 1 : 0, c->lstm_epoch, c->stream);
    }
    HIPCHK(c, hipGetLastError());
    c->lgate_clean = true;
    c->done_armed = seq;
    c->last_n = n;
    return serial_end(c);
}

// predict_cu_depth() of resi_to_cu_depth_LDP.py:108-129 for one frame; the new state stays in HBM.
// state source: host `state_in` when given, else zeros (resident == false) or the previous step's state in HBM.
// Two halves: ldp_step_begin enqueues everything, ldp_step_end waits and finishes the bookkeeping.  streamed: the caller is still
// FILLING the page-locked luma buffer (ethcnn_rows_ready reports its CTU rows); the tile stage waits for them row by row.
static int ldp_step_begin(ethcnn_ctx* c, const uint8_t* luma, int w, int h, ptrdiff_t pitch, int qp, int i_frame,
                          const float* state_in, bool resident, float* probs, bool streamed) {
    if (!c || !luma || !probs) return c ? set_err(c, ETHCNN_ERR_ARG, "null pointer") : ETHCNN_ERR_ARG;
    if (c->ldp.open || c->ai.open) return set_err(c, ETHCNN_ERR_ARG, "ethcnn_ldp_step_begin: the previous streamed call has not been ended (ethcnn_ldp_step_end / ethcnn_predict_luma_end)");
    if (w <= 0 || h <= 0 || pitch < w) return set_err(c, ETHCNN_ERR_ARG, "bad geometry");
    if (!c->have_weights) return set_err(c, ETHCNN_ERR_NOWEIGHTS, "no CNN weights loaded");
    if (!c->have_lstm) return set_err(c, ETHCNN_ERR_NOWEIGHTS, "no LSTM weights loaded");
    const int nctu = ((w + 63) / 64) * ((h + 63) / 64);
    const size_t lbytes = (size_t)(h - 1) * pitch + w;  // the meaningful bytes of a pitched plane
    if (streamed) {
        if (!c->h_rows) return set_err(c, ETHCNN_ERR_DEVICE, "ethcnn_ldp_step_begin: no page-locked memory for the row words");
        if ((h + 63) / 64 > kStreamCtuRows) return set_err(c, ETHCNN_ERR_ARG, "ethcnn_ldp_step_begin: more than %d CTU rows", kStreamCtuRows);
        if (!in_pinned(c, luma, lbytes))
            return set_err(c, ETHCNN_ERR_ARG, "ethcnn_ldp_step_begin: the luma buffer must come from ethcnn_host_alloc (the kernels read it in place while it is filled)");
    }
    int rc = ensure_staging(c, lbytes, (size_t)nctu * kNVec * 4);
    if (rc) return rc;
    if (nctu > c->lstm_cap) c->state_cur = -1;  // the buffers are about to be reallocated
    rc = ensure_lstm_buffers(c, nctu);
    if (rc) return rc;
    const size_t sbytes = (size_t)nctu * 2 * kNVec * 4;
    int in = -1;  // index of the input state buffer, -1 = zeros
    if (state_in) {
        in = 0;
        HIPCHK(c, hipMemcpyAsync(c->d_state[in], state_in, sbytes, hipMemcpyHostToDevice, c->stream));
    } else if (resident) {
        if (c->state_cur < 0 || c->state_nctu != nctu)
            return set_err(c, ETHCNN_ERR_ARG, "ethcnn_ldp_step: frame %d needs the previous frame's state, but none is resident for %d CTUs",
                           i_frame, nctu);
        in = c->state_cur;
    }
    const int out = (in == 0) ? 1 : 0;
    // Latency path (one frame, lock-step with the encoder): buffers from ethcnn_host_alloc are used IN PLACE -- the tile stage
    // reads the luma over PCIe while it runs, the heads / gate stages write the 84 B per CTU straight into the caller's
    // memory -- instead of two copy launches around the kernels
    const uint8_t* d_luma = c->d_in[0];
    // page-locked luma is read in place over PCIe by the tile stage (one coalesced pass while it runs): measured 123.8 us per
    // 1080p call against 128.3 us for "DMA it into HBM first, then the single-launch pass" (profiles/r03_latency_ldp.txt;
    // ETHCNN_LDP_INPLACE=0 selects the latter for A/B runs)
    static const bool copy_first = [] { const char* e = dev_env("ETHCNN_LDP_INPLACE"); return e && std::atoi(e) == 0; }();
    const bool in_place = in_pinned(c, luma, lbytes) && (streamed || !copy_first);
    if (in_place) d_luma = luma;
    else HIPCHK(c, hipMemcpyAsync(c->d_in[0], luma, lbytes, hipMemcpyHostToDevice, c->stream));
    const size_t pbytes = (size_t)nctu * kNOut * 4;
    // probabilities: straight into page-locked host memory (the caller's, else the staging buffer + one memcpy), the launch's
    // last block reports through the completion word
    float* d_probs = in_pinned(c, probs, pbytes) ? probs : (c->done_sync ? c->h_out[0] : c->d_lprobs);
    if (streamed) {
        // everything that may wait for the stream (allocations, re-zeroing of sync areas) happens BEFORE kernels are queued that wait
        // for the caller -- who may be this very thread, about to fill the buffer when the call returns
        HIPCHK(c, hipSetDevice(c->device));
        rc = ensure_workspace(c, std::min(nctu, c->max_ctus), (nctu + kSubBatch - 1) / kSubBatch);
        if (rc == 0) rc = ensure_lgate(c, nctu);
        if (rc) return rc;
    }
    c->luma_over_pcie = (d_luma == luma);
    c->tile_wait_rows = streamed ? c->h_rows : nullptr;
    rc = ethcnn_resi_vectors_device(c, d_luma, w, h, pitch, c->d_vec);
    c->luma_over_pcie = false;
    c->tile_wait_rows = nullptr;
    if (rc == 0) rc = ethcnn_lstm_step_device(c, c->d_vec, in >= 0 ? c->d_state[in] : nullptr, nctu, qp, i_frame, c->d_state[out], d_probs);
    if (rc) {
        // (streamed: kernels already queued may be waiting for rows the caller will now never report: release them -- the result is
        // discarded -- so that the stream drains)
        if (streamed) {
            for (int cy = 0; cy < (h + 63) / 64; ++cy) __atomic_store_n(c->h_rows + cy, c->rows_seq, __ATOMIC_RELEASE);
            (void)hipStreamSynchronize(c->stream);  // (nothing may still be reading the caller's buffer when the error is returned)
            if (++c->rows_seq == 0) c->rows_seq = 1;
        }
        return rc;
    }
    c->ldp.open = true;
    c->ldp.streamed = streamed;
    c->ldp.probs = probs;
    c->ldp.d_probs = d_probs;
    c->ldp.pbytes = pbytes;
    c->ldp.out = out;
    c->ldp.in = in;
    c->ldp.nctu = nctu;
    c->ldp.in_from_host = state_in != nullptr;
    c->ldp.prev_cur = c->state_cur;    // (after ensure_lstm_buffers: -1 when the buffers were reallocated)
    c->ldp.prev_nctu = c->state_nctu;
    return ETHCNN_OK;
}

static int ldp_step_end(ethcnn_ctx* c) {
    if (!c) return ETHCNN_ERR_ARG;
    if (!c->ldp.open) return set_err(c, ETHCNN_ERR_ARG, "ethcnn_ldp_step_end: no step has been begun");
    c->ldp.open = false;
    const unsigned seq = c->rows_seq;
    if (c->ldp.streamed) {  // the next streamed picture's number is fixed from here on (ethcnn_rows_ready may run before its begin)
        ++c->rows_seq;
        if (c->rows_seq == 0) c->rows_seq = 1;
    }
    if (c->ldp.d_probs == c->d_lprobs) {
        c->done_armed = 0;
        HIPCHK(c, hipMemcpyAsync(c->ldp.probs, c->d_lprobs, c->ldp.pbytes, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, stream_sync(c));
    if (c->ldp.streamed && __atomic_load_n(c->h_done + 1, __ATOMIC_ACQUIRE) == seq) {
        // computed on rows that never arrived: the output state is garbage, the INPUT state (the other buffer) is untouched, so the caller
        // may run the frame again (ethcnn_ldp_step on the by now complete buffer) with the same arguments.  What is resident afterwards
        // (the size must always belong to the buffer):
        //   resident input        that state, as before the step;
        //   caller's state_in     the copy of it in buffer 0, for THIS frame's CTU count;
        //   zeros (i_frame <= 1)  whatever was resident before, unless the failed step wrote over it (it wrote buffer 0).
        if (c->ldp.in_from_host) {
            c->state_cur = 0;
            c->state_nctu = c->ldp.nctu;
        } else if (c->ldp.in >= 0) {
            c->state_cur = c->ldp.in;
        } else {
            c->state_cur = (c->ldp.prev_cur == c->ldp.out) ? -1 : c->ldp.prev_cur;
            c->state_nctu = c->ldp.prev_nctu;
        }
        return set_err(c, ETHCNN_ERR_ROWS_TIMEOUT, "ethcnn_ldp_step_end: the kernels waited 1 s for luma rows that were never reported (ethcnn_rows_ready)");
    }
    if (c->ldp.d_probs == c->h_out[0]) std::memcpy(c->ldp.probs, c->ldp.d_probs, c->ldp.pbytes);
    c->state_cur = c->ldp.out;
    c->state_nctu = c->ldp.nctu;
    return ETHCNN_OK;
}

static int ldp_step_impl(ethcnn_ctx* c, const uint8_t* luma, int w, int h, ptrdiff_t pitch, int qp, int i_frame,
                         const float* state_in, bool resident, float* probs) {
    const int rc = ldp_step_begin(c, luma, w, h, pitch, qp, i_frame, state_in, resident, probs, false);
    return rc ? rc : ldp_step_end(c);
}

extern "C" int ethcnn_ldp_step(ethcnn_ctx* c, const uint8_t* luma, int w, int h, ptrdiff_t pitch, int qp, int i_frame,
                               const float* state_in, float* probs) {
    return ldp_step_impl(c, luma, w, h, pitch, qp, i_frame, state_in, /*resident=*/!state_in && i_frame > 1, probs);
}

// ---- streamed input: begin (kernels queued, waiting for rows) | rows_ready (any thread, as the buffer fills) | end
extern "C" int ethcnn_ldp_step_begin(ethcnn_ctx* c, const uint8_t* luma, int w, int h, ptrdiff_t pitch, int qp, int i_frame,
                                     const float* state_in, float* probs) {
    if (!c) return ETHCNN_ERR_ARG;
    const bool was_open = c->ai.open || c->ldp.open;
    const unsigned seq = c->rows_seq;
    const int rc = ldp_step_begin(c, luma, w, h, pitch, qp, i_frame, state_in, /*resident=*/!state_in && i_frame > 1, probs, true);
    // (a failed begin consumes the picture's number: see ethcnn_predict_luma_begin)
    if (rc != ETHCNN_OK && !was_open && c->rows_seq == seq && ++c->rows_seq == 0) c->rows_seq = 1;
    return rc;
}

extern "C" int ethcnn_rows_ready(ethcnn_ctx* c, int ctu_row_begin, int ctu_row_end) {
    // (thread-safe: touches nothing but the row words; no error text -- another thread may be inside a call on this context)
    if (!c || !c->h_rows || ctu_row_begin < 0 || ctu_row_end > kStreamCtuRows || ctu_row_begin > ctu_row_end) return ETHCNN_ERR_ARG;
    const unsigned seq = __atomic_load_n(&c->rows_seq, __ATOMIC_RELAXED);
#if defined(__SSE2__)
    _mm_sfence();  // rows written with non-temporal stores (big memcpy calls, streaming converters) are not ordered by a release store alone
#endif
    for (int cy = ctu_row_begin; cy < ctu_row_end; ++cy) __atomic_store_n(c->h_rows + cy, seq, __ATOMIC_RELEASE);
    return ETHCNN_OK;
}

extern "C" int ethcnn_ldp_step_end(ethcnn_ctx* c) { return ldp_step_end(c); }

extern "C" int ethcnn_ldp_get_state(ethcnn_ctx* c, float* state_out, size_t nfloats) {
    if (!c || !state_out) return c ? set_err(c, ETHCNN_ERR_ARG, "null pointer") : ETHCNN_ERR_ARG;
    if (c->state_cur < 0) return set_err(c, ETHCNN_ERR_ARG, "ethcnn_ldp_get_state: no resident state (call ethcnn_ldp_step first)");
    if (nfloats != (size_t)c->state_nctu * 2 * kNVec)
        return set_err(c, ETHCNN_ERR_ARG, "ethcnn_ldp_get_state: the resident state holds %zu floats, not %zu", (size_t)c->state_nctu * 2 * kNVec, nfloats);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(state_out, c->d_state[c->state_cur], nfloats * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ETHCNN_OK;
}

// the reference's per-frame call as one synchronous function: state in and out through host memory
extern "C" int ethcnn_ldp_predict_frame(ethcnn_ctx* c, const uint8_t* luma, int w, int h, ptrdiff_t pitch, int qp,
                                        int i_frame, const float* state_in, float* state_out, float* probs) {
    if (!c || !luma || !state_out || !probs) return c ? set_err(c, ETHCNN_ERR_ARG, "null pointer") : ETHCNN_ERR_ARG;
    int rc = ldp_step_impl(c, luma, w, h, pitch, qp, i_frame, state_in, /*resident=*/false, probs);  // NULL = zeros here
    if (rc) return rc;
    const int nctu = ((w + 63) / 64) * ((h + 63) / 64);
    return ethcnn_ldp_get_state(c, state_out, (size_t)nctu * 2 * kNVec);
}
