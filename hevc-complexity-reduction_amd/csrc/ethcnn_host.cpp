// ethcnn_host.cpp -- host and file entry points: staging ring, worker pool, latency path, streamed pictures, the YUV-file driver
#include "ethcnn_ctx.h"

void free_staging(ethcnn_ctx* c) {
    for (int i = 0; i < kStageBufs; ++i) {
        if (c->h_in[i]) (void)hipHostFree(c->h_in[i]);
        if (c->h_out[i]) (void)hipHostFree(c->h_out[i]);
        if (c->d_in[i]) (void)hipFree(c->d_in[i]);
        if (c->d_out[i]) (void)hipFree(c->d_out[i]);
        c->h_in[i] = nullptr; c->h_out[i] = nullptr; c->d_in[i] = nullptr; c->d_out[i] = nullptr;
        hipEvent_t* evs[3] = {&c->ev_in[i], &c->ev_comp[i], &c->ev_out[i]};
        for (hipEvent_t* e : evs) {
            if (*e) (void)hipEventDestroy(*e);
            *e = nullptr;
        }
    }
    c->in_cap = c->out_cap = 0;
}

// `nbufs` of the ring are needed by the caller (the single-frame LDP / resi entry points use one)
int ensure_staging(ethcnn_ctx* c, size_t in_bytes, size_t out_bytes, int nbufs) {
    bool have = in_bytes <= c->in_cap && out_bytes <= c->out_cap;
    for (int i = 0; i < nbufs && have; ++i) have = c->h_in[i] != nullptr;
    if (!have) {
        HIPCHK(c, hipDeviceSynchronize());
        const size_t ic = std::max(in_bytes, c->in_cap), oc = std::max(out_bytes, c->out_cap);
        int keep = nbufs;
        for (int i = 0; i < kStageBufs; ++i)
            if (c->h_in[i]) keep = std::max(keep, i + 1);
        free_staging(c);  // on any failure below the partial ring is released by ethcnn_destroy / the next call
        AffinityScope on_gpu_node(c->numa);  // page-locked memory is allocated where the calling thread runs
        for (int i = 0; i < keep; ++i) {
            HIPCHK(c, hipHostMalloc((void**)&c->h_in[i], ic, hipHostMallocDefault));
            HIPCHK(c, hipHostMalloc((void**)&c->h_out[i], oc, hipHostMallocDefault));
            HIPCHK(c, hipMalloc((void**)&c->d_in[i], ic));
            HIPCHK(c, hipMalloc((void**)&c->d_out[i], oc));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_comp[i], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming));
        }
        c->in_cap = ic;
        c->out_cap = oc;
    }
    return 0;
}

// A staging group is filled in units of (frame, band of rows) of ~512 KiB so that the units divide
// evenly over the pool whatever the frame count of the group: fn(frame, row0, rows).
// CPUs this process may actually use: the logical count capped by the cgroup CPU quota (cpu.max "1600000 100000" = 16)
static int usable_cpus() {
    int n = std::max(1, (int)std::thread::hardware_concurrency());
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32];
        long per = 0;
        if (std::fscanf(f, "%31s %ld", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0)
            n = std::max(1, std::min(n, (int)((std::atol(q) + per / 2) / per)));
        std::fclose(f);
    }
    return n;
}

// Fill threads of ONE context when `local_workers` contexts (one process per GPU) share the node's CPU budget
// (`usable`: the cgroup quota / logical count; <= 0 = probe it): the budget is divided, never multiplied -- 8 workers under
// a 16-core quota get 2 threads each, not 8 x 16 runnable threads on 16 cores (the oversubscription that collapses any
// OpenMP-style pool under CFS throttling: 256 threads ran 5x slower than 16 on the GPU boxes).  A single worker takes
// min(16, usable, logical / 2): more than 16 fill threads measured slower (scripts/s3_threads.py).
extern "C" int ethcnn_host_thread_budget(int local_workers, int usable) {
    if (usable <= 0) usable = std::min(usable_cpus(), std::max(1, (int)std::thread::hardware_concurrency() / 2));
    const int w = std::max(1, local_workers);
    return std::max(1, std::min(16, usable / w));
}

// how many predictor processes share this node: the launcher says (ETHCNN_LOCAL_WORKERS; predict_sharded sets it for its
// workers), else torchrun's LOCAL_WORLD_SIZE, else one
static int local_workers() {
    for (const char* name : {"ETHCNN_LOCAL_WORKERS", "LOCAL_WORLD_SIZE"})
        if (const char* e = std::getenv(name))
            if (std::atoi(e) > 0) return std::atoi(e);
    return 1;
}

static HostPool* host_pool(ethcnn_ctx* c) {
    if (!c->pool) {
        // (shard_workers: this context is one of several workers of ONE process -- ethcnn_predict_yuv_file_sharded -- on top of
        // whatever other predictor processes share the node)
        int nt = ethcnn_host_thread_budget(local_workers() * std::max(1, c->shard_workers), 0);
        if (c->host_threads_opt > 0) nt = std::min(32, c->host_threads_opt);
        if (const char* e = std::getenv("ETHCNN_HOST_THREADS")) nt = std::max(1, std::min(32, std::atoi(e)));  // explicit override
        c->pool = new HostPool(nt, c->numa);
    }
    return c->pool;
}

extern "C" int ethcnn_host_threads(ethcnn_ctx* c) {  // the pool size this context uses (creates the pool)
    return c ? host_pool(c)->size() : ETHCNN_ERR_ARG;
}

// Copy into page-locked staging memory with non-temporal stores: no read-for-ownership of the destination lines and no
// cache pollution, so the fill threads take a third less DRAM bandwidth away from the DMA engine that is draining the
// previous group at the same time (profiles/r02_host_copy.txt: 56 GB/s through the fill | H2D pipeline against 49 GB/s
// with memcpy; the DMA engine alone moves 57.5).
static void nt_copy(uint8_t* dst, const uint8_t* src, size_t n) {
#if !defined(__SSE2__)
    std::memcpy(dst, src, n);  // no streaming stores on this host ISA: plain copy, same result
    return;
#else
    const size_t head = std::min(n, (size_t)(-(uintptr_t)dst & 15));
    if (head) std::memcpy(dst, src, head);
    dst += head; src += head; n -= head;
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i*)(src + i)), b = _mm_loadu_si128((const __m128i*)(src + i + 16));
        const __m128i c2 = _mm_loadu_si128((const __m128i*)(src + i + 32)), d = _mm_loadu_si128((const __m128i*)(src + i + 48));
        _mm_stream_si128((__m128i*)(dst + i), a);
        _mm_stream_si128((__m128i*)(dst + i + 16), b);
        _mm_stream_si128((__m128i*)(dst + i + 32), c2);
        _mm_stream_si128((__m128i*)(dst + i + 48), d);
    }
    if (i < n) std::memcpy(dst + i, src + i, n - i);
    _mm_sfence();
#endif
}

template <typename Fn>
static int parallel_bands(ethcnn_ctx* c, int nframes, int w, int h, Fn fn) {
    const size_t plane = (size_t)w * h;
    const int bands = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::min(h, 32), plane / (512u << 10)));
    const std::function<int(int)> unit = [&](int u) -> int {
        const int f = u / bands, b = u % bands;
        const int r0 = (int)((long)h * b / bands), r1 = (int)((long)h * (b + 1) / bands);
        return fn(f, r0, r1 - r0);
    };
    return host_pool(c)->run(nframes * bands, unit);
}

// Host pipeline over a ring of kStageBufs pinned + device buffer pairs: for each group of frames, `fill(buf, f0, nf)` packs
// luma planes tightly (pitch = width) into pinned memory on the worker pool, then H2D -> kernels -> D2H run on three
// streams, and `drain(buf, f0, nf)` consumes the pinned probabilities -- fill of group i+2, H2D of group i+1, kernels +
// D2H of group i and the drain of group i-1 all overlap.
template <typename Fill, typename Drain>
static int host_pipeline(ethcnn_ctx* c, int w, int h, int nframes, int qp, Fill fill, Drain drain) {
    FrameGeom g;
    int rc = make_geom(c, w, h, w, (ptrdiff_t)w * h, &g);
    if (rc) return rc;
    if (!c->have_weights) return set_err(c, ETHCNN_ERR_NOWEIGHTS, "no weights loaded");
    if (nframes == 0) return ETHCNN_OK;
    HIPCHK(c, hipSetDevice(c->device));
    // Group size: whole frames, >= ~4096 CTUs (kernel efficiency), ~16 groups per call so that the pipeline's ramp (first
    // fill, last kernels + D2H + drain) is a small part of it, never larger than the workspace.  A frame larger than the
    // workspace is still one group; run_pass splits it.
    const int fpg = std::max(1, std::min(std::min(nframes, c->max_ctus / g.nctu),
                                         std::max((4096 + g.nctu - 1) / g.nctu, (nframes + 15) / 16)));
    const size_t plane = (size_t)w * h;
    rc = ensure_staging(c, plane * fpg, (size_t)fpg * g.nctu * kNOut * 4, kStageBufs);
    if (rc == 0) rc = ensure_side_streams(c);
    if (rc) return rc;
    struct Group { int f0, nf; };
    std::vector<Group> groups;
    {   // short groups at both ends: the DMA engine (the bottleneck stage) starts after the FIRST fill and everything behind
        // the LAST H2D (kernels, D2H, drain) is exposed -- 1, 2, then fpg frames per group, and 2, 1 at the end
        std::vector<int> head, tail;
        int left = nframes;
        for (int sz = 1; sz < fpg && left > 4 * fpg; sz *= 2) {
            head.push_back(sz);
            tail.push_back(sz);
            left -= 2 * sz;
        }
        int f = 0;
        for (int sz : head) { groups.push_back({f, sz}); f += sz; }
        int tail_sum = 0;
        for (int sz : tail) tail_sum += sz;
        for (; f < nframes - tail_sum; ) { const int nf = std::min(fpg, nframes - tail_sum - f); groups.push_back({f, nf}); f += nf; }
        for (size_t i = tail.size(); i-- > 0;) { groups.push_back({f, tail[i]}); f += tail[i]; }
    }
    const size_t ng = groups.size();
    auto retire = [&](size_t gi) -> int {  // group gi's probabilities are in pinned memory: hand them to the caller
        const int b = (int)(gi % kStageBufs);
        HIPCHK(c, hipEventSynchronize(c->ev_out[b]));
        return drain(c->h_out[b], groups[gi].f0, groups[gi].nf);
    };
    auto body = [&]() -> int {
        for (size_t gi = 0; gi < ng; ++gi) {
            const int b = (int)(gi % kStageBufs);
            const Group& G = groups[gi];
            if (gi >= (size_t)kStageBufs) {  // ring slot b was last used by group gi - kStageBufs: retire it first
                int r = retire(gi - kStageBufs);
                if (r) return r;
            }
            int r = fill(c->h_in[b], G.f0, G.nf);
            if (r) return r;
            HIPCHK(c, hipMemcpyAsync(c->d_in[b], c->h_in[b], plane * G.nf, hipMemcpyHostToDevice, c->copy_in));
            HIPCHK(c, hipEventRecord(c->ev_in[b], c->copy_in));
            for (const Pass& p : plan_passes(g.nctu, G.nf, c->max_ctus)) {
                r = run_pass(c, c->d_in[b], g, p.ctu0, p.n, qp, c->d_out[b] + (size_t)p.ctu0 * kNOut, c->ev_in[b]);
                if (r) return r;
            }
            HIPCHK(c, hipEventRecord(c->ev_comp[b], c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->copy_out, c->ev_comp[b], 0));
            HIPCHK(c, hipMemcpyAsync(c->h_out[b], c->d_out[b], (size_t)G.nf * g.nctu * kNOut * 4, hipMemcpyDeviceToHost, c->copy_out));
            HIPCHK(c, hipEventRecord(c->ev_out[b], c->copy_out));
            // slot reuse needs no further stream waits: before group gi + kStageBufs touches slot b the host has
            // synchronised on ev_out[b] (retire), which orders after this group's H2D, kernels and D2H
        }
        for (size_t gi = (ng >= (size_t)kStageBufs ? ng - kStageBufs : 0); gi < ng; ++gi) {  // the groups still in flight, in order
            int r = retire(gi);
            if (r) return r;
        }
        return ETHCNN_OK;
    };
    const int result = body();
    (void)hipStreamSynchronize(c->copy_in);  // on an error path nothing may still be reading / writing the ring
    if (c->s_tile) (void)hipStreamSynchronize(c->s_tile);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->copy_out);
    return result;
}

// One picture (or a few small ones): a single pass of < 8192 CTUs (kPipelineMinCtus; <= 2304 of them as ONE launch, above that five).  The staging ring above is built for throughput -- a pool
// wake-up, three streams and two events per group -- which is most of the time of a one-frame call.  Here: (copy into pinned
// staging unless the caller's buffer IS pinned) -> H2D -> the pass -> D2H, all on the main stream, one synchronisation.
static int predict_luma_latency(ethcnn_ctx* c, const uint8_t* luma, int w, int h, ptrdiff_t pitch, ptrdiff_t fstride, int nframes,
                                int qp, float* probs) {
    FrameGeom g;
    int rc = make_geom(c, w, h, w, (ptrdiff_t)w * h, &g);
    if (rc) return rc;
    if (!c->have_weights) return set_err(c, ETHCNN_ERR_NOWEIGHTS, "no weights loaded");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t plane = (size_t)w * h, in_bytes = plane * nframes, out_bytes = (size_t)nframes * g.nctu * kNOut * 4;
    rc = ensure_staging(c, in_bytes, out_bytes, 1);
    if (rc) return rc;
    const bool packed = pitch == w && fstride == (ptrdiff_t)plane;
    const uint8_t* src = luma;
    // One picture, single-launch pass: the launch PULLS the picture from page-locked memory itself (the caller's, or the staging
    // buffer) -- no copy-engine launch in front of the kernel, and the trunk / FC1 / heads of the first CTU rows run while the last
    // rows are still on the bus (ethcnn_small.hip, "PULL form"; profiles/r04_latency_host.txt)
    const bool stage_rows = !(packed && in_pinned(c, luma, in_bytes));
    const bool pull = nframes == 1 && c->pull && c->small_launch && w % 16 == 0 && g.nctu <= kSmallPassMaxCtus;
    // A picture in PAGEABLE (or pitched) memory has to be copied into the page-locked staging buffer first: that copy is STREAMED into
    // the pass (the mechanism of ethcnn_predict_luma_begin, applied to the library's own staging) -- the launch is queued on the
    // staging buffer, then the rows are copied CTU row by CTU row, each reported as it lands
    const bool stream_stage = pull && stage_rows && c->h_rows != nullptr && g.ch <= kStreamCtuRows;
    const bool banded = !pull && nframes == 1 && g.nctu > kSubBatch && c->small_launch && w % 16 == 0;  // (below)
    if (stage_rows && !banded && !stream_stage) {  // tight planes into the pinned staging buffer
        for (int f = 0; f < nframes; ++f) {
            const uint8_t* s = luma + (size_t)f * fstride;
            uint8_t* d = c->h_in[0] + (size_t)f * plane;
            if (pitch == w) std::memcpy(d, s, plane);
            else for (int y = 0; y < h; ++y) std::memcpy(d + (size_t)y * w, s + (size_t)y * pitch, (size_t)w);
        }
        src = c->h_in[0];
    }
    // (single-launch forms: the launch's last block hands the probabilities to the host itself and reports through the completion
    // word -- no copy launch behind the kernel)
    float* const dst = in_pinned(c, probs, out_bytes) ? probs : c->h_out[0];
    bool direct = false;
    unsigned streamed_seq = 0;  // != 0: the pass was queued on a staging buffer that was still being filled (checked after the wait below)
    if (stream_stage) {
        rc = ensure_workspace(c, g.nctu, chunks_per_frame(g.nctu));  // (whatever may wait for the stream: before the waiting kernels are queued)
        if (rc) return rc;
        c->host_probs = dst;
        c->host_probs_used = false;
        c->luma_over_pcie = true;
        c->tile_wait_rows = c->h_rows;
        rc = run_pass(c, c->h_in[0], g, 0, g.nctu, qp, c->d_out[0]);
        c->luma_over_pcie = false;
        c->tile_wait_rows = nullptr;
        c->host_probs = nullptr;
        direct = rc == 0 && c->host_probs_used;
        const unsigned seq = c->rows_seq;
        {   // experiments build only: hold this thread between the launch and the copy (tests/test_gpu_small.py: the give-up path)
            static const int stall_ms = [] { const char* e = dev_env("ETHCNN_TEST_STAGE_STALL_MS"); return e ? std::atoi(e) : 0; }();
            if (stall_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(stall_ms));
        }
        const bool copy = rc == 0;  // (rows are reported also when the launch failed: whatever is queued must drain)
        const std::function<int(int)> ctu_row = [&](int cy) -> int {
            if (copy)
                for (int y = cy * kCtu; y < std::min(h, cy * kCtu + kCtu); ++y) std::memcpy(c->h_in[0] + (size_t)y * w, luma + (size_t)y * pitch, (size_t)w);
#if defined(__SSE2__)
            _mm_sfence();  // (memcpy may use non-temporal stores)
#endif
            __atomic_store_n(c->h_rows + cy, seq, __ATOMIC_RELEASE);
            return 0;
        };
        // (4 MB and more -- a 2160p plane is 150 us of single-threaded memcpy, as long as its transfer -- on the worker pool)
        if (copy && plane >= (4u << 20)) (void)host_pool(c)->run(g.ch, ctu_row);
        else for (int cy = 0; cy < g.ch; ++cy) (void)ctu_row(cy);
        streamed_seq = seq;
        if (++c->rows_seq == 0) c->rows_seq = 1;
    } else if (pull) {
        c->host_probs = dst;
        c->host_probs_used = false;
        c->luma_over_pcie = true;
        rc = run_pass(c, src, g, 0, g.nctu, qp, c->d_out[0]);
        c->luma_over_pcie = false;
        c->host_probs = nullptr;
        direct = rc == 0 && c->host_probs_used;
    } else if (banded && (rc = ensure_side_streams(c)) != 0) {
        return rc;
    } else if (banded) {
        // (the round's first form, kept for ETHCNN_PULL=0 A/B runs)  One big picture (3840x2160: 8.3 MB = 151 us of PCIe against ~100 us of kernels, serial until round 4): the picture is
        // cut on its gate sub-batch boundaries (1024 CTUs in raster order: video_to_cu_depth.py:61-73, so gate scope is intact) and
        // the rows the next piece needs travel on the copy stream while the previous piece computes; each piece is one
        // single-launch pass.  Same passes as a small workspace would plan: results are bit-identical.
        int rows_done = 0, k = 0;
        for (int ctu0 = 0; ctu0 < g.nctu && rc == 0; ctu0 += kSubBatch) {
            const int n = std::min(kSubBatch, g.nctu - ctu0);
            const int row_end = std::min(h, ((ctu0 + n - 1) / g.cw + 1) * kCtu);
            hipEvent_t ready = nullptr;
            if (row_end > rows_done) {
                if (stage_rows) {  // pageable / pitched caller memory: this piece's rows into the pinned staging buffer first -- while
                                   // the previous piece's DMA and kernels run
                    for (int y = rows_done; y < row_end; ++y) std::memcpy(c->h_in[0] + (size_t)y * w, luma + (size_t)y * pitch, (size_t)w);
                    src = c->h_in[0];
                }
                HIPCHK(c, hipMemcpyAsync(c->d_in[0] + (size_t)rows_done * w, src + (size_t)rows_done * w, (size_t)(row_end - rows_done) * w,
                                         hipMemcpyHostToDevice, c->copy_in));
                ready = c->e_band[k++ % 4];
                HIPCHK(c, hipEventRecord(ready, c->copy_in));
                rows_done = row_end;
            }
            rc = run_pass(c, c->d_in[0], g, ctu0, n, qp, c->d_out[0] + (size_t)ctu0 * kNOut, ready);
        }
    } else {
        HIPCHK(c, hipMemcpyAsync(c->d_in[0], src, in_bytes, hipMemcpyHostToDevice, c->stream));
        for (const Pass& p : plan_passes(g.nctu, nframes, c->max_ctus)) {
            rc = run_pass(c, c->d_in[0], g, p.ctu0, p.n, qp, c->d_out[0] + (size_t)p.ctu0 * kNOut);
            if (rc) break;
        }
    }
    if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
    // (letting the single-launch pass write the probabilities straight into page-locked host memory and report through the
    // completion word was measured 3 us SLOWER than this copy + hipStreamSynchronize: 96 heads blocks storing 4-byte words
    // over PCIe; profiles/r03_completion_word.txt)
    if (direct) {
        HIPCHK(c, stream_sync(c));
    } else {
        c->done_armed = 0;
        HIPCHK(c, hipMemcpyAsync(dst, c->d_out[0], out_bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (streamed_seq && __atomic_load_n(c->h_done + 1, __ATOMIC_ACQUIRE) == streamed_seq) {
        // The pass was queued before the staging copy and its tile blocks gave up waiting for rows (this thread was stopped for more
        // than ~1 s between the launch and the copy: SIGSTOP / ptrace, a VM pause, a swap storm on the pageable source) -- they then
        // computed on whatever the staging buffer held.  ethcnn_predict_luma_end / ethcnn_ldp_step_end report this as an error because
        // the CALLER owns the fill there; here the fill is ours and the staging buffer is complete by now: run the pass again on it,
        // not streamed (a silently wrong ETHCNN_OK otherwise).
        c->host_probs = dst;
        c->host_probs_used = false;
        c->luma_over_pcie = true;
        rc = run_pass(c, c->h_in[0], g, 0, g.nctu, qp, c->d_out[0]);
        c->luma_over_pcie = false;
        c->host_probs = nullptr;
        if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
        if (c->host_probs_used) {
            HIPCHK(c, stream_sync(c));
        } else {
            c->done_armed = 0;
            HIPCHK(c, hipMemcpyAsync(dst, c->d_out[0], out_bytes, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        ++c->stream_stage_reruns;
    }
    if (dst != probs) std::memcpy(probs, dst, out_bytes);
    return ETHCNN_OK;
}

// ---- ONE picture with streamed input (the in-process encoder hook converts HM's 16-bit picture to 8 bits row by row: the
// conversion is as long as the prediction, and the prediction can run under it).  begin: the pass is queued on the page-locked
// buffer and waits for its CTU rows (ethcnn_rows_ready); end: result copy + wait.  The pass is the one ethcnn_predict_luma runs.
static int predict_luma_begin_impl(ethcnn_ctx* c, const uint8_t* luma, int w, int h, int qp, float* probs);
extern "C" int ethcnn_predict_luma_begin(ethcnn_ctx* c, const uint8_t* luma, int w, int h, int qp, float* probs) {
    if (!c || !luma || !probs) return c ? set_err(c, ETHCNN_ERR_ARG, "null pointer") : ETHCNN_ERR_ARG;
    const bool was_open = c->ai.open || c->ldp.open;
    const unsigned seq = c->rows_seq;
    const int rc = predict_luma_begin_impl(c, luma, w, h, qp, probs);
    // a begin that fails consumes the picture's number all the same: rows already reported for it (ethcnn_rows_ready may run ahead of
    // begin) must not count for the next streamed picture.  (Not when the failure is "another streamed call is open": that one's rows.)
    if (rc != ETHCNN_OK && !was_open && c->rows_seq == seq && ++c->rows_seq == 0) c->rows_seq = 1;
    return rc;
}
static int predict_luma_begin_impl(ethcnn_ctx* c, const uint8_t* luma, int w, int h, int qp, float* probs) {
    if (c->ai.open || c->ldp.open) return set_err(c, ETHCNN_ERR_ARG, "ethcnn_predict_luma_begin: a streamed call is still open on this context");
    FrameGeom g;
    int rc = make_geom(c, w, h, w, (ptrdiff_t)w * h, &g);
    if (rc) return rc;
    if (!c->have_weights) return set_err(c, ETHCNN_ERR_NOWEIGHTS, "no weights loaded");
    if (!c->h_rows) return set_err(c, ETHCNN_ERR_DEVICE, "ethcnn_predict_luma_begin: no page-locked memory for the row words");
    if (g.ch > kStreamCtuRows) return set_err(c, ETHCNN_ERR_ARG, "ethcnn_predict_luma_begin: more than %d CTU rows", kStreamCtuRows);
    if (g.nctu >= kPipelineMinCtus || g.nctu > c->max_ctus)
        return set_err(c, ETHCNN_ERR_ARG, "ethcnn_predict_luma_begin: %d CTUs: streamed input is for one picture in one pass (< %d CTUs)", g.nctu, std::min(kPipelineMinCtus, c->max_ctus + 1));
    const size_t plane = (size_t)w * h, out_bytes = (size_t)g.nctu * kNOut * 4;
    if (!in_pinned(c, luma, plane))
        return set_err(c, ETHCNN_ERR_ARG, "ethcnn_predict_luma_begin: the luma buffer must come from ethcnn_host_alloc (the kernels read it in place while it is filled)");
    HIPCHK(c, hipSetDevice(c->device));
    rc = ensure_staging(c, plane, out_bytes, 1);
    // (everything that may wait for the stream happens before kernels are queued that wait for the caller)
    if (rc == 0) rc = ensure_workspace(c, g.nctu, chunks_per_frame(g.nctu));
    if (rc == 0 && c->fc1_plan != 0) rc = ensure_fast_weights(c, c->fc1_plan);  // (first use packs and uploads the 16-bit weight images)
    if (rc) return rc;
    c->host_probs = in_pinned(c, probs, out_bytes) ? probs : c->h_out[0];
    c->host_probs_used = false;
    c->luma_over_pcie = true;
    c->tile_wait_rows = c->h_rows;
    rc = run_pass(c, luma, g, 0, g.nctu, qp, c->d_out[0]);
    c->luma_over_pcie = false;
    c->tile_wait_rows = nullptr;
    c->host_probs = nullptr;
    c->ai.direct = rc == 0 && c->host_probs_used;
    if (rc) {  // release whatever is already queued (the result is discarded) and let the stream drain
        for (int cy = 0; cy < g.ch; ++cy) __atomic_store_n(c->h_rows + cy, c->rows_seq, __ATOMIC_RELEASE);
        (void)hipStreamSynchronize(c->stream);
        if (++c->rows_seq == 0) c->rows_seq = 1;
        return rc;
    }
    c->ai.open = true;
    c->ai.probs = probs;
    c->ai.out_bytes = out_bytes;
    return ETHCNN_OK;
}

extern "C" int ethcnn_predict_luma_end(ethcnn_ctx* c) {
    if (!c) return ETHCNN_ERR_ARG;
    if (!c->ai.open) return set_err(c, ETHCNN_ERR_ARG, "ethcnn_predict_luma_end: no picture has been begun");
    c->ai.open = false;
    const unsigned seq = c->rows_seq;
    if (++c->rows_seq == 0) c->rows_seq = 1;  // the next streamed picture's number is fixed from here on
    float* dst = in_pinned(c, c->ai.probs, c->ai.out_bytes) ? c->ai.probs : c->h_out[0];
    if (c->ai.direct) {  // (the launch's last block has written dst itself)
        HIPCHK(c, stream_sync(c));
    } else {
        c->done_armed = 0;
        HIPCHK(c, hipMemcpyAsync(dst, c->d_out[0], c->ai.out_bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (__atomic_load_n(c->h_done + 1, __ATOMIC_ACQUIRE) == seq)
        return set_err(c, ETHCNN_ERR_ROWS_TIMEOUT, "ethcnn_predict_luma_end: the kernels waited 1 s for luma rows that were never reported (ethcnn_rows_ready)");
    if (dst != c->ai.probs) std::memcpy(c->ai.probs, dst, c->ai.out_bytes);
    return ETHCNN_OK;
}

extern "C" int ethcnn_predict_luma(ethcnn_ctx* c, const uint8_t* luma, int w, int h, ptrdiff_t pitch, ptrdiff_t fstride,
                                   int nframes, int qp, float* probs) {
    if (!c || !luma || !probs || nframes < 0) return c ? set_err(c, ETHCNN_ERR_ARG, "null pointer / negative frame count") : ETHCNN_ERR_ARG;
    if (pitch < w) return set_err(c, ETHCNN_ERR_ARG, "pitch %td < width %d", pitch, w);
    const int nctu = ((w + 63) / 64) * ((h + 63) / 64);
    if (w > 0 && h > 0 && nframes > 0 && (long)nframes * nctu < kPipelineMinCtus)  // a picture, not a sequence (strictly below: a pass of
        // exactly kPipelineMinCtus CTUs runs its tile stage on the side stream, which the latency path's H2D copy is not ordered with)
        return predict_luma_latency(c, luma, w, h, pitch, fstride, nframes, qp, probs);
    auto fill = [&](uint8_t* dst, int f0, int nf) -> int {
        return parallel_bands(c, nf, w, h, [&](int f, int r0, int rows) -> int {
            const uint8_t* src = luma + (size_t)(f0 + f) * fstride + (size_t)r0 * pitch;
            uint8_t* d = dst + (size_t)f * w * h + (size_t)r0 * w;
            if (pitch == w) nt_copy(d, src, (size_t)w * rows);
            else for (int y = 0; y < rows; ++y) nt_copy(d + (size_t)y * w, src + (size_t)y * pitch, (size_t)w);
            return 0;
        });
    };
    auto drain = [&](const float* src, int f0, int nf) -> int {
        std::memcpy(probs + (size_t)f0 * nctu * kNOut, src, (size_t)nf * nctu * kNOut * 4);
        return 0;
    };
    return host_pipeline(c, w, h, nframes, qp, fill, drain);
}

// video_to_cu_depth.py:120-145 minus argv/model selection (those live in the launcher).
// shard == false: frames [0, all) -> out_path via temp file + rename.
// shard == true : frames [f0, f1) pwritten at f0 * nctu * 84 into the EXISTING, pre-sized
//                 out_path (one worker per GPU, disjoint ranges, no collective).
// mode 0: the whole file -> out_path (temp + rename); 1: shard, frames [f0, f1) pwritten at their place into an existing, pre-sized
// out_path; 2: range, frames [f0, f1) -> an out_path of their own (temp + rename) = get_prob(n_frames_start, n_frames_end)
static int yuv_frames(ethcnn_ctx* c, const char* yuv, int w, int h, int qp, const char* out_path, int mode,
                      int64_t f0, int64_t f1, int64_t* nframes_out) {
    const bool shard = (mode == 1);
    if (w <= 0 || h <= 0) return set_err(c, ETHCNN_ERR_ARG, "bad frame size %dx%d", w, h);
    struct stat st;
    if (stat(yuv, &st) != 0) return set_err(c, ETHCNN_ERR_IO, "cannot stat %s: %s", yuv, std::strerror(errno));
    const int64_t frame_bytes = (int64_t)w * h * 3 / 2;  // :136  width * height * 3 // 2
    if (frame_bytes == 0 || st.st_size % frame_bytes != 0)  // :137 assert(file_bytes % frame_bytes == 0)
        return set_err(c, ETHCNN_ERR_FORMAT, "%s: size %lld is not a multiple of the %dx%d 4:2:0 frame size %lld", yuv,
                       (long long)st.st_size, w, h, (long long)frame_bytes);
    const int64_t total = st.st_size / frame_bytes;
    if (nframes_out) *nframes_out = total;
    if (mode == 0) { f0 = 0; f1 = total; }
    if (f0 < 0 || f1 < f0 || f1 > total) return set_err(c, ETHCNN_ERR_ARG, "frame range [%lld,%lld) outside 0..%lld", (long long)f0, (long long)f1, (long long)total);
    const int nctu = ((w + 63) / 64) * ((h + 63) / 64);
    FILE* fin = std::fopen(yuv, "rb");
    if (!fin) return set_err(c, ETHCNN_ERR_IO, "cannot open %s: %s", yuv, std::strerror(errno));
    const std::string tmp = std::string(out_path) + ".tmp." + std::to_string((long)getpid());
    FILE* fout = shard ? std::fopen(out_path, "r+b") : std::fopen(tmp.c_str(), "wb");
    if (!fout) {
        std::fclose(fin);
        return set_err(c, ETHCNN_ERR_IO, "cannot open %s for writing: %s", shard ? out_path : tmp.c_str(), std::strerror(errno));
    }
    const int fd = fileno(fin), ofd = fileno(fout);
    // pread lands in a cache-resident bounce buffer and goes on to the pinned staging memory with non-temporal stores
    // (nt_copy): pread straight into the staging buffer writes its lines through the cache (read-for-ownership + write
    // back) beside the DMA engine.  ETHCNN_FILE_IO=direct keeps the single-copy form.  (A read-only mapping of the file
    // + nt_copy, one copy and no syscalls, was measured at HALF the rate: page faults.)
    static const bool bounce = [] { const char* e = dev_env("ETHCNN_FILE_IO"); return !(e && std::strcmp(e, "direct") == 0); }();
    constexpr size_t kBounce = 128u << 10;
    auto fill = [&](uint8_t* dst, int g0, int nf) -> int {
        // luma only; chroma (w*h/2 bytes per frame) is never read (:47-48)
        const int rc = parallel_bands(c, nf, w, h, [&](int f, int r0, int rows) -> int {
            size_t got = 0;
            const size_t want = (size_t)w * rows;
            const off_t off = (off_t)(f0 + g0 + f) * frame_bytes + (off_t)r0 * w;
            uint8_t* d = dst + (size_t)f * w * h + (size_t)r0 * w;
            if (bounce) {
                alignas(64) static thread_local uint8_t tmp[kBounce];
                while (got < want) {
                    const ssize_t r = pread(fd, tmp, std::min(kBounce, want - got), off + (off_t)got);
                    if (r <= 0) return ETHCNN_ERR_IO;
                    nt_copy(d + got, tmp, (size_t)r);
                    got += (size_t)r;
                }
                return 0;
            }
            while (got < want) {
                const ssize_t r = pread(fd, d + got, want - got, off + (off_t)got);
                if (r <= 0) return ETHCNN_ERR_IO;
                got += (size_t)r;
            }
            return 0;
        });
        return rc ? set_err(c, rc, "short read in %s (frames %lld..%lld)", yuv, (long long)(f0 + g0), (long long)(f0 + g0 + nf - 1)) : 0;
    };
    auto drain = [&](const float* src, int g0, int nf) -> int {
        const size_t bytes = (size_t)nf * nctu * kNOut * 4;
        const off_t off = (off_t)((shard ? f0 : 0) + g0) * nctu * kNOut * 4;
        size_t done = 0;
        while (done < bytes) {
            const ssize_t r = pwrite(ofd, (const char*)src + done, bytes - done, off + (off_t)done);
            if (r <= 0) return set_err(c, ETHCNN_ERR_IO, "write to %s failed: %s", out_path, std::strerror(errno));
            done += (size_t)r;
        }
        return 0;
    };
    int rc = host_pipeline(c, w, h, (int)(f1 - f0), qp, fill, drain);
    std::fclose(fin);
    if (std::fclose(fout) != 0 && rc == 0) rc = set_err(c, ETHCNN_ERR_IO, "close of output failed");
    if (!shard) {
        if (rc == 0 && std::rename(tmp.c_str(), out_path) != 0)
            rc = set_err(c, ETHCNN_ERR_IO, "rename %s -> %s failed: %s", tmp.c_str(), out_path, std::strerror(errno));
        if (rc != 0) std::remove(tmp.c_str());
    }
    return rc;
}

extern "C" int ethcnn_predict_yuv_file(ethcnn_ctx* c, const char* yuv, int w, int h, int qp, const char* out_path,
                                       int64_t* nframes_out) {
    if (!c || !yuv || !out_path) return c ? set_err(c, ETHCNN_ERR_ARG, "null path") : ETHCNN_ERR_ARG;
    return yuv_frames(c, yuv, w, h, qp, out_path, 0, 0, 0, nframes_out);
}

// ---- the whole file over several GPUs from ONE process: a worker THREAD per listed device ("process-per-GPU or
// thread-per-GPU"; 8e: contiguous frame ranges, no collective).  The reference's caller blocks in system() (TAppEncCfg.cpp:2317-2321):
// what the encoder sees is the wall time of the command, and N Python interpreters that each create a context, parse and crc a
// checkpoint and pin a staging ring to do ~18 ms of GPU work (C4 / 8) cost more than they share out.  Here the calling context is
// worker 0; a peer context per further device is created once (cached with the context, destroyed with it) and takes a COPY of the
// caller's weights, thresholds and plan -- no second checkpoint parse.  Each worker preads its frames and pwrites them at
// frame_begin * nctu * 84 into a pre-sized temp file; one rename at the end.  Byte-identical to ethcnn_predict_yuv_file.
// worker k of n gets frames [floor(k F / n), floor((k + 1) F / n)): the split of sharding.frame_range (the process-per-GPU form), in one place
extern "C" int ethcnn_shard_range(int64_t nframes, int workers, int k, int64_t* frame_begin, int64_t* frame_end) {
    if (nframes < 0 || workers <= 0 || k < 0 || k >= workers || !frame_begin || !frame_end) return ETHCNN_ERR_ARG;
    *frame_begin = nframes * k / workers;
    *frame_end = nframes * (k + 1) / workers;
    return ETHCNN_OK;
}

static int sync_peer(const ethcnn_ctx* c, ethcnn_ctx* p) {  // (on the peer's worker thread: errors stay in p->err)
    if (p->weights_from != c || p->weights_gen != c->weights_gen) {
        const int rc = ethcnn_load_blob(p, c->blob.data(), c->blob.size());
        if (rc) return rc;
        p->weights_from = c;
        p->weights_gen = c->weights_gen;
    }
    p->thr1 = c->thr1;
    p->thr2 = c->thr2;
    p->fc1_plan = c->fc1_plan;
    p->max_ctus = c->max_ctus;
    for (int pl : {2, 3}) {  // the accuracy guard is a function of the weights: decided once, by the caller's context
        p->guard_state[pl] = c->guard_state[pl] == 3 ? 0 : c->guard_state[pl];
        p->guard_bound[pl] = c->guard_bound[pl];
        p->guard_measured[pl] = c->guard_measured[pl];
        p->guard_info[pl] = c->guard_info[pl];
    }
    return 0;
}
extern "C" int ethcnn_predict_yuv_file_sharded(ethcnn_ctx* c, const int* devices, int ndevices, const char* yuv, int w, int h, int qp,
                                               const char* out_path, int64_t* nframes_out) {
    if (!c || !yuv || !out_path || !devices || ndevices < 1 || ndevices > 64)
        return c ? set_err(c, ETHCNN_ERR_ARG, "ethcnn_predict_yuv_file_sharded: null path / device list, or not 1..64 devices") : ETHCNN_ERR_ARG;
    if (devices[0] != c->device)
        return set_err(c, ETHCNN_ERR_ARG, "ethcnn_predict_yuv_file_sharded: devices[0] = %d, but this context (worker 0) lives on device %d", devices[0], c->device);
    if (!c->have_weights) return set_err(c, ETHCNN_ERR_NOWEIGHTS, "no weights loaded");
    if (w <= 0 || h <= 0) return set_err(c, ETHCNN_ERR_ARG, "bad frame size %dx%d", w, h);
    struct stat st;
    if (stat(yuv, &st) != 0) return set_err(c, ETHCNN_ERR_IO, "cannot stat %s: %s", yuv, std::strerror(errno));
    const int64_t frame_bytes = (int64_t)w * h * 3 / 2;
    if (frame_bytes == 0 || st.st_size % frame_bytes != 0)
        return set_err(c, ETHCNN_ERR_FORMAT, "%s: size %lld is not a multiple of the %dx%d 4:2:0 frame size %lld", yuv, (long long)st.st_size, w, h,
                       (long long)frame_bytes);
    const int64_t total = st.st_size / frame_bytes;
    if (nframes_out) *nframes_out = total;
    if (ndevices == 1 || total <= 1) return yuv_frames(c, yuv, w, h, qp, out_path, 0, 0, 0, nullptr);
    if (c->fc1_plan) {  // (guard + weight images once, here; the peers inherit the verdict)
        HIPCHK(c, hipSetDevice(c->device));
        const int rc = ensure_fast_weights(c, c->fc1_plan);
        if (rc) return rc;
    }
    // workers: the caller's context + one cached peer per further list entry (a device may be listed more than once).  A peer is created
    // (first call) and brought up to date BY ITS OWN WORKER THREAD, side by side with the others: a context costs ~60 ms of runtime calls,
    // eight of them one after the other cost more than the whole job (profiles/r06_cold_start_first_form.txt)
    const int nw = (int)std::min<int64_t>(ndevices, total);
    if (c->shard_workers != nw) {  // the host budget is divided by the worker count: pools of another division are rebuilt
        delete c->pool;
        c->pool = nullptr;
        for (ethcnn_ctx* p : c->peers)
            if (p) { delete p->pool; p->pool = nullptr; p->shard_workers = nw; }
        c->shard_workers = nw;
    }
    if ((int)c->peers.size() < nw - 1) c->peers.resize(nw - 1, nullptr);
    std::vector<std::string> werr(nw);
    auto prepare_peer = [&](int k) -> int {  // runs on worker k's thread; touches only c->peers[k - 1] and reads the caller's context
        ethcnn_ctx*& p = c->peers[k - 1];
        if (p && p->device != devices[k]) { ethcnn_destroy(p); p = nullptr; }  // another device list than last time
        if (!p) {
            ethcnn_options o{};
            o.device = devices[k];
            o.max_ctus_per_pass = c->max_ctus;
            const int rc = ethcnn_create(&p, &o);
            if (rc) { werr[k] = ethcnn_last_error(nullptr); p = nullptr; return rc; }
            p->shard_workers = nw;
        }
        const int rc = sync_peer(c, p);
        if (rc) werr[k] = p->err;
        return rc;
    };
    const int nctu = ((w + 63) / 64) * ((h + 63) / 64);
    const std::string tmp = std::string(out_path) + ".tmp." + std::to_string((long)getpid());
    {   // pre-size the temp file: every worker pwrites its own range
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || ftruncate(fileno(f), (off_t)(total * nctu * kNOut * 4)) != 0) {
            if (f) std::fclose(f);
            std::remove(tmp.c_str());
            return set_err(c, ETHCNN_ERR_IO, "cannot create %s: %s", tmp.c_str(), std::strerror(errno));
        }
        std::fclose(f);
    }
    std::vector<int> rcs(nw, 0);
    std::vector<std::thread> th;
    auto range = [&](int k, int64_t* f0, int64_t* f1) { (void)ethcnn_shard_range(total, nw, k, f0, f1); };
    auto work = [&](int k) {
        int64_t f0, f1;
        range(k, &f0, &f1);
        rcs[k] = prepare_peer(k);
        if (rcs[k] == 0) {
            rcs[k] = yuv_frames(c->peers[k - 1], yuv, w, h, qp, tmp.c_str(), 1, f0, f1, nullptr);
            if (rcs[k]) werr[k] = c->peers[k - 1]->err;
        }
    };
    std::vector<int> inline_k;  // workers whose thread could not be started (no exception may cross the C ABI): their share runs here, afterwards
    for (int k = 1; k < nw; ++k) {
        try {
            th.emplace_back(work, k);
        } catch (...) {
            inline_k.push_back(k);
        }
    }
    {
        int64_t f0, f1;
        range(0, &f0, &f1);
        rcs[0] = yuv_frames(c, yuv, w, h, qp, tmp.c_str(), 1, f0, f1, nullptr);
    }
    for (int k : inline_k) work(k);
    for (auto& t : th) t.join();
    for (int k = 0; k < nw; ++k)
        if (rcs[k]) {
            std::remove(tmp.c_str());
            if (k > 0) return set_err(c, rcs[k], "worker %d (device %d): %s", k, devices[k], werr[k].c_str());
            return rcs[0];
        }
    if (std::rename(tmp.c_str(), out_path) != 0) {
        const int e = errno;
        std::remove(tmp.c_str());
        return set_err(c, ETHCNN_ERR_IO, "rename %s -> %s failed: %s", tmp.c_str(), out_path, std::strerror(e));
    }
    return ETHCNN_OK;
}

extern "C" int ethcnn_predict_yuv_range(ethcnn_ctx* c, const char* yuv, int w, int h, int qp, const char* out_path,
                                        int64_t frame_begin, int64_t frame_end) {
    if (!c || !yuv || !out_path) return c ? set_err(c, ETHCNN_ERR_ARG, "null path") : ETHCNN_ERR_ARG;
    return yuv_frames(c, yuv, w, h, qp, out_path, 2, frame_begin, frame_end, nullptr);
}

extern "C" int ethcnn_predict_yuv_shard(ethcnn_ctx* c, const char* yuv, int w, int h, int qp, const char* out_path,
                                        int64_t frame_begin, int64_t frame_end) {
    if (!c || !yuv || !out_path) return c ? set_err(c, ETHCNN_ERR_ARG, "null path") : ETHCNN_ERR_ARG;
    return yuv_frames(c, yuv, w, h, qp, out_path, 1, frame_begin, frame_end, nullptr);
}

// [p, p + bytes) inside a buffer from ethcnn_host_alloc: page-locked and mapped at the same address on the device (unified
// addressing), so kernels can read / write it in place -- one PCIe crossing, no staging copy, no copy launch
bool in_pinned(const ethcnn_ctx* c, const void* p, size_t bytes) {
    for (const auto& r : c->pinned)
        if ((const char*)p >= r.first && (const char*)p + bytes <= r.first + r.second) return true;
    return false;
}
