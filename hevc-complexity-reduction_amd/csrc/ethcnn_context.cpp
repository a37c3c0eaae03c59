// ethcnn_context.cpp -- lifecycle of the context, workspace, thresholds, profiling, execution-plan switches, device plumbing (include/ethcnn.h)
#include "ethcnn_ctx.h"

static thread_local std::string g_create_err;

int set_err(ethcnn_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}

extern "C" const char* ethcnn_version(void) { return "ethcnn-mi355x 0.1 (gfx950)"; }

extern "C" const char* ethcnn_last_error(const ethcnn_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

// ------------------------------------------------------------------ workspace -------
int ensure_side_streams(ethcnn_ctx* c) {
    hipStream_t* ss[] = {&c->copy_in, &c->copy_out, &c->s_tile};
    for (hipStream_t* s : ss)
        if (!*s) HIPCHK(c, hipStreamCreateWithFlags(s, hipStreamNonBlocking));
    return 0;
}

static void free_workspace(ethcnn_ctx* c) {
    Workspace& w = c->ws;
    void* ptrs[] = {w.featb, w.xs, w.xm, w.xl, w.feat, w.h1, w.h2, w.logits, w.raw, w.flags, c->xs1, c->xm1, c->xl1, c->h1_1, c->flags1};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    w = Workspace();
    c->xs1 = c->xm1 = c->xl1 = nullptr;
    c->h1_1 = nullptr;
    c->flags1 = nullptr;
}

// the buffer set of pass parity p
Workspace ws_view(const ethcnn_ctx* c, int p) {
    Workspace v = c->ws;
    if (p) {
        v.xs = c->xs1;
        v.xm = c->xm1;
        v.xl = c->xl1;
        v.h1 = c->h1_1;
        v.flags = c->flags1;
    }
    return v;
}

// `chunks`: gate sub-batches of the pass (sync_words: their predicates)
int ensure_workspace(ethcnn_ctx* c, int n, int chunks) {
    Workspace& w = c->ws;
    const int cap = (n + 15) / 16 * 16;
    if (cap > w.cap) {
        free_workspace(c);  // hipFree synchronises: nothing in flight still uses the old buffers
        HIPCHK(c, hipMalloc((void**)&w.xs, (size_t)cap * 4096));
        HIPCHK(c, hipMalloc((void**)&w.xm, (size_t)cap * 2048));
        HIPCHK(c, hipMalloc((void**)&w.xl, (size_t)cap * 512));
        HIPCHK(c, hipMalloc((void**)&w.feat, (size_t)cap * kNFeat * 4));
        HIPCHK(c, hipMalloc((void**)&w.h1, (size_t)cap * kNVec * 4));
        HIPCHK(c, hipMalloc((void**)&w.h2, (size_t)cap * kNFc2 * 4));
        HIPCHK(c, hipMalloc((void**)&w.logits, (size_t)cap * kNOut * 4));
        HIPCHK(c, hipMalloc((void**)&w.raw, (size_t)cap * kNOut * 4));
        HIPCHK(c, hipMalloc((void**)&c->xs1, (size_t)cap * 4096));
        HIPCHK(c, hipMalloc((void**)&c->xm1, (size_t)cap * 2048));
        HIPCHK(c, hipMalloc((void**)&c->xl1, (size_t)cap * 512));
        HIPCHK(c, hipMalloc((void**)&c->h1_1, (size_t)cap * kNVec * 4));
        w.cap = cap;
    }
    if (c->fc1_plan != 0 && !w.featb)  // plans 2 / 3: the features as fp16 x 2 pieces, 10,752 B per CTU (ethcnn_spec.h)
        HIPCHK(c, hipMalloc((void**)&w.featb, (size_t)((w.cap + 31) / 32) * kFastPairBytes));
    const int words = sync_words(chunks);
    if (words > w.flags_cap) {
        if (w.flags) (void)hipFree(w.flags);  // hipFree synchronises the device: no pass in flight still uses them
        if (c->flags1) (void)hipFree(c->flags1);
        w.flags = c->flags1 = nullptr;
        w.flags_cap = 0;
        HIPCHK(c, hipMalloc((void**)&w.flags, (size_t)words * sizeof(int)));
        HIPCHK(c, hipMalloc((void**)&c->flags1, (size_t)words * sizeof(int)));
        w.flags_cap = words;
    }
    return 0;
}

// ------------------------------------------------------------------ lifecycle -------
extern "C" int ethcnn_create(ethcnn_ctx** out, const ethcnn_options* opt) {
    if (!out) return set_err(nullptr, ETHCNN_ERR_ARG, "ethcnn_create: out is NULL");
    *out = nullptr;
    const auto t_enter = std::chrono::steady_clock::now();
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);  // (the first HIP call of a process: loads and initialises the runtime)
    const auto t_runtime = std::chrono::steady_clock::now();
    if (e != hipSuccess || ndev <= 0)
        return set_err(nullptr, ETHCNN_ERR_DEVICE, "no HIP device available (%s): libethcnn has no CPU fallback",
                       e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    const int dev = opt ? opt->device : 0;
    if (dev < 0 || dev >= ndev) return set_err(nullptr, ETHCNN_ERR_ARG, "device %d out of range (0..%d)", dev, ndev - 1);
    // (where the rest of the call goes: printed under ETHCNN_TIMING=1, scripts/cold_start.py)
    std::vector<std::pair<const char*, std::chrono::steady_clock::time_point>> marks;
    auto mark = [&](const char* what) { marks.emplace_back(what, std::chrono::steady_clock::now()); };
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess)
        return set_err(nullptr, ETHCNN_ERR_DEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(nullptr, ETHCNN_ERR_DEVICE, "device %d is %s; libethcnn is built for gfx950 only", dev, prop.gcnArchName);
    ethcnn_ctx* c = new (std::nothrow) ethcnn_ctx();
    if (!c) return set_err(nullptr, ETHCNN_ERR_NOMEM, "out of memory");
    c->device = dev;
    std::snprintf(c->devname, sizeof c->devname, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    // the workspace is a whole number of 1024-CTU sub-batches, at most what the kernels' 32-bit offsets cover (ethcnn_spec.h)
    if (opt && opt->max_ctus_per_pass > 0)
        c->max_ctus = std::min(kMaxCtusPerPass, std::max(1024, (int)(((long long)opt->max_ctus_per_pass + 1023) / 1024 * 1024)));
    if (opt && opt->host_threads > 0) c->host_threads_opt = opt->host_threads;
    mark("device properties");
    // the main stream only: a HIP stream costs 10-15 ms to create (profiles/r06_cold_start.txt: "4 streams 68 ms" of a 260 ms command whose GPU
    // work is 41 us), and one picture -- the reference's own C1 run, the encoder hook, every LDP frame -- never leaves the main stream.
    // The H2D / D2H / CTU-load streams are created by the first call that pipelines (ensure_side_streams)
    if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        ethcnn_destroy(c);
        return set_err(nullptr, ETHCNN_ERR_DEVICE, "cannot create a HIP stream on device %d", dev);
    }
    mark("device init + main stream");
    {
        hipEvent_t* evs[] = {&c->e_tile[0], &c->e_tile[1], &c->e_trunk[0], &c->e_trunk[1], &c->e_main, &c->e_fc1[0], &c->e_fc1[1],
                             &c->e_band[0], &c->e_band[1], &c->e_band[2], &c->e_band[3]};
        for (hipEvent_t* e : evs)
            if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) {
                ethcnn_destroy(c);
                return set_err(nullptr, ETHCNN_ERR_DEVICE, "cannot create HIP events on device %d", dev);
            }
    }
    if (const char* e = dev_env("ETHCNN_OVERLAP")) c->overlap = std::atoi(e) != 0;  // development knob (A/B runs)
    if (const char* e = dev_env("ETHCNN_SMALL")) c->small_launch = std::atoi(e) != 0;  // development knob (A/B runs)
    if (const char* e = std::getenv("ETHCNN_FC1_PLAN")) {  // user-facing: start contexts in plan 2 / 3
        const int pl = std::atoi(e);
        c->fc1_plan = (pl == 2 || pl == 3) ? pl : 0;
    }
    if (const char* e = dev_env("ETHCNN_DONE_WORD")) c->done_sync = std::atoi(e) != 0;  // development knob (A/B runs)
    if (const char* e = dev_env("ETHCNN_PULL")) c->pull = std::atoi(e) != 0;            // development knob (A/B runs)
    if (const char* e = dev_env("ETHCNN_TILE_AFTER_FC1")) c->tile_after_fc1 = std::atoi(e) != 0;  // development knob (A/B runs)
    if (const char* e = dev_env("ETHCNN_LSTM_ONE_LAUNCH")) c->lstm_one_launch = std::atoi(e) != 0;  // development knob (A/B runs)
    mark("11 events");
    if (hipHostMalloc((void**)&c->h_done, 64, hipHostMallocDefault) != hipSuccess) {
        ethcnn_destroy(c);
        return set_err(nullptr, ETHCNN_ERR_DEVICE, "cannot allocate the completion word on device %d", dev);
    }
    *c->h_done = 0;
    c->h_done[1] = 0;
    if (hipHostMalloc((void**)&c->h_rows, kStreamCtuRows * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) c->h_rows = nullptr;  // (streamed
    // steps then report ETHCNN_ERR_DEVICE; everything else works)
    if (c->h_rows) std::memset(c->h_rows, 0, kStreamCtuRows * sizeof(unsigned));
    c->tile_blocks = c->cus = prop.multiProcessorCount;
    mark("page-locked words");
    {   // the GPU's NUMA node -> its CPU list (/sys/devices/system/node/nodeN/cpulist: "64-127,192-255"); ETHCNN_NUMA_BIND=0 opts out
        int node = -1;
        const char* off = std::getenv("ETHCNN_NUMA_BIND");
        const bool asked = !(off && std::atoi(off) == 0);
        hipError_t qe = hipErrorNotSupported;
        if (asked) {  // the PCI device's own sysfs entry first (works inside containers that see one GPU of eight) ...
            char pci[96];
            std::snprintf(pci, sizeof pci, "/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node", prop.pciDomainID, prop.pciBusID, prop.pciDeviceID);
            if (FILE* f = std::fopen(pci, "r")) {
                if (std::fscanf(f, "%d", &node) == 1 && node >= 0) qe = hipSuccess;
                std::fclose(f);
            }
            if (qe != hipSuccess) qe = hipDeviceGetAttribute(&node, hipDeviceAttributeHostNumaId, dev);  // ... then the runtime's view
        }
        (void)hipGetLastError();  // an optional query: its failure must not stay behind as the thread's sticky last error
        if (qe == hipSuccess && node >= 0) {
            char path[96], buf[1024];
            std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
            if (FILE* f = std::fopen(path, "r")) {
                if (std::fgets(buf, sizeof buf, f)) {
                    CPU_ZERO(&c->numa.set);
                    int n = 0;
                    for (char* q = buf; *q && *q != '\n';) {
                        char* end;
                        const long a = std::strtol(q, &end, 10);
                        long b = a;
                        if (end == q) break;
                        if (*end == '-') { q = end + 1; b = std::strtol(q, &end, 10); }
                        for (long k = a; k <= b && k < CPU_SETSIZE; ++k) { CPU_SET((int)k, &c->numa.set); ++n; }
                        q = (*end == ',') ? end + 1 : end;
                    }
                    cpu_set_t own;  // what this process may use: never pin outside it
                    if (n > 0 && sched_getaffinity(0, sizeof own, &own) == 0) {
                        n = 0;
                        for (int k = 0; k < CPU_SETSIZE; ++k) {
                            if (CPU_ISSET(k, &c->numa.set) && !CPU_ISSET(k, &own)) CPU_CLR(k, &c->numa.set);
                            if (CPU_ISSET(k, &c->numa.set)) ++n;
                        }
                    }
                    c->numa.valid = n > 0;
                    if (c->numa.valid) {
                        const size_t L = std::strlen(c->devname);
                        std::snprintf(c->devname + L, sizeof c->devname - L, ", host NUMA node %d", node);
                    }
                }
                std::fclose(f);
            }
        }
    }
    if (const char* e = dev_env("ETHCNN_TILE_BLOCKS")) c->tile_blocks = std::max(1, std::atoi(e));
    mark("NUMA lookup");
    if (const char* tm = std::getenv("ETHCNN_TIMING"))
        if (std::atoi(tm) != 0) {
            std::string line = "ethcnn_create timing (ms): first HIP call (runtime init) " + std::to_string(std::chrono::duration<double, std::milli>(t_runtime - t_enter).count());
            auto prev = t_runtime;
            for (const auto& m : marks) {
                char buf[96];
                std::snprintf(buf, sizeof buf, " | %s %.2f", m.first, std::chrono::duration<double, std::milli>(m.second - prev).count());
                line += buf;
                prev = m.second;
            }
            std::fprintf(stderr, "%s\n", line.c_str());
        }
    c->startup_ms[0] = std::chrono::duration<double, std::milli>(t_runtime - t_enter).count();
    c->startup_ms[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
    *out = c;
    return ETHCNN_OK;
}

extern "C" void ethcnn_destroy(ethcnn_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if ((c->ldp.open || c->ai.open) && c->h_rows)  // destroyed between a streamed begin and its end: release the kernels that wait for
        for (int cy = 0; cy < kStreamCtuRows; ++cy) __atomic_store_n(c->h_rows + cy, c->rows_seq, __ATOMIC_RELEASE);  // rows (else: 1 s each)
    (void)hipDeviceSynchronize();  // on THIS context's device: a launch still in flight may store the completion word
    if (c->h_done) { (void)hipHostFree(c->h_done); c->h_done = nullptr; }
    if (c->h_rows) { (void)hipHostFree(c->h_rows); c->h_rows = nullptr; }
    for (auto& p : c->pending) { c->ev_pool.push_back(p.a); c->ev_pool.push_back(p.b); }
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    free_workspace(c);
    free_staging(c);
    {   // workers of ethcnn_predict_yuv_file_sharded: torn down side by side (a context's teardown is ~15-30 ms of runtime calls)
        std::vector<std::thread> th;
        for (ethcnn_ctx* p : c->peers)
            if (p) th.emplace_back([p] { ethcnn_destroy(p); });
        for (auto& t : th) t.join();
        c->peers.clear();
    }
    (void)hipSetDevice(c->device);
    for (const auto& r : c->pinned) (void)hipHostFree(const_cast<char*>(r.first));  // ethcnn_host_alloc buffers die with the context
    delete c->pool;
    if (c->dw_arena) (void)hipFree(c->dw_arena);
    if (c->dw_fast) (void)hipFree(c->dw_fast);
    if (c->dw_trunk16) (void)hipFree(c->dw_trunk16);
    if (c->dw_heads16) (void)hipFree(c->dw_heads16);
    {
        void* lp[] = {c->d_lstm, c->d_vec, c->d_state[0], c->d_state[1], c->d_lprobs, c->d_lgate, c->d_ssync};
        for (void* p : lp)
            if (p) (void)hipFree(p);
    }
    {
        hipEvent_t evs[] = {c->e_tile[0], c->e_tile[1], c->e_trunk[0], c->e_trunk[1], c->e_main, c->e_fc1[0], c->e_fc1[1], c->e_band[0], c->e_band[1], c->e_band[2], c->e_band[3]};
        for (hipEvent_t e : evs)
            if (e) (void)hipEventDestroy(e);
    }
    hipStream_t streams[] = {c->stream, c->copy_in, c->copy_out, c->s_tile};
    for (hipStream_t st : streams)
        if (st) (void)hipStreamDestroy(st);
    delete c;
}

// where ethcnn_create's time went (cold start of the drop-in command: VERDICT r05 weak 8)
extern "C" int ethcnn_get_startup_times(const ethcnn_ctx* c, double* runtime_init_ms, double* create_ms) {
    if (!c) return ETHCNN_ERR_ARG;
    if (runtime_init_ms) *runtime_init_ms = c->startup_ms[0];
    if (create_ms) *create_ms = c->startup_ms[1];
    return ETHCNN_OK;
}

extern "C" int ethcnn_device_name(const ethcnn_ctx* c, char* out, size_t cap) {
    if (!c || !out || cap == 0) return ETHCNN_ERR_ARG;
    std::snprintf(out, cap, "%s", c->devname);
    return ETHCNN_OK;
}

// ----------------------------------------------------------------- thresholds -------
extern "C" int ethcnn_set_thresholds(ethcnn_ctx* c, float t1, float t2) {
    if (!c) return ETHCNN_ERR_ARG;
    c->thr1 = t1;
    c->thr2 = t2;
    return ETHCNN_OK;
}
extern "C" int ethcnn_get_thresholds(const ethcnn_ctx* c, float* t1, float* t2) {
    if (!c || !t1 || !t2) return ETHCNN_ERR_ARG;
    *t1 = c->thr1;
    *t2 = c->thr2;
    return ETHCNN_OK;
}
extern "C" int ethcnn_load_thresholds(ethcnn_ctx* c, const char* path) {
    if (!c || !path) return ETHCNN_ERR_ARG;
    char err[400];
    float a, b;
    const int rc = parse_thr_info(path, &a, &b, err, sizeof err);
    if (rc) return set_err(c, rc, "%s", err);
    c->thr1 = a;
    c->thr2 = b;
    return ETHCNN_OK;
}

// ------------------------------------------------------------------ profiling -------
hipEvent_t get_event(ethcnn_ctx* c) {
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) e = nullptr;
    return e;
}
static void drain_events(ethcnn_ctx* c) {
    for (auto& p : c->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            c->times.ms[p.stage] += ms;
        } else {  // keep ms / timed / timed_ctus consistent: the failed pair leaves the sample
            c->times.timing_errors++;
            c->times.timed[p.stage]--;
            c->times.timed_ctus[p.stage] -= p.ctus;
        }
        c->ev_pool.push_back(p.a);
        c->ev_pool.push_back(p.b);
    }
    c->pending.clear();
}
extern "C" int ethcnn_set_profiling(ethcnn_ctx* c, int on) {
    if (!c) return ETHCNN_ERR_ARG;
    drain_events(c);
    c->profiling = on < 0 ? 0 : (on > 2 ? 2 : on);
    c->fc1_sample = 0;  // the first pass after this call is a timed one
    return ETHCNN_OK;
}
extern "C" int ethcnn_set_pass_pipeline(ethcnn_ctx* c, int on) {
    if (!c) return ETHCNN_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());  // nothing of the old mode is in flight when the mode changes
    c->overlap = on ? 1 : 0;
    return ETHCNN_OK;
}

extern "C" int ethcnn_set_fc1_plan(ethcnn_ctx* c, int plan) {
    if (!c) return ETHCNN_ERR_ARG;
    if (plan != 0 && plan != 2 && plan != 3)  // (1 was round 4's bf16 x 3 form of FC1: removed, dominated by plan 2 in every metric)
        return set_err(c, ETHCNN_ERR_ARG, "plan must be 0 (exact fp32, default), 2 (FC1 as fp16 x 2) or 3 (FC1, trunk and heads as fp16 x 2), got %d", plan);
    c->fc1_plan = plan;  // takes effect with the next pass enqueued
    return ETHCNN_OK;
}
extern "C" int ethcnn_get_fc1_plan(const ethcnn_ctx* c) { return c ? c->fc1_plan : ETHCNN_ERR_ARG; }

extern "C" int ethcnn_set_small_pass_launch(ethcnn_ctx* c, int on) {
    if (!c) return ETHCNN_ERR_ARG;
    c->small_launch = on ? 1 : 0;  // takes effect with the next pass enqueued; results do not depend on it
    return ETHCNN_OK;
}

extern "C" int ethcnn_get_stage_times(ethcnn_ctx* c, ethcnn_stage_times* out) {
    if (!c || !out) return ETHCNN_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    drain_events(c);
    *out = c->times;
    return ETHCNN_OK;
}
extern "C" int ethcnn_reset_stage_times(ethcnn_ctx* c) {
    if (!c) return ETHCNN_ERR_ARG;
    drain_events(c);
    c->times = ethcnn_stage_times{};
    return ETHCNN_OK;
}

// ------------------------------------------------------------ device plumbing -------
extern "C" int ethcnn_device_alloc(ethcnn_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return ETHCNN_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(out, bytes ? bytes : 1));
    return ETHCNN_OK;
}
extern "C" int ethcnn_device_free(ethcnn_ctx* c, void* p) {
    if (!c) return ETHCNN_ERR_ARG;
    if (p) HIPCHK(c, hipFree(p));
    return ETHCNN_OK;
}
extern "C" int ethcnn_host_alloc(ethcnn_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return ETHCNN_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    AffinityScope on_gpu_node(c->numa);
    HIPCHK(c, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    c->pinned.emplace_back((const char*)*out, bytes ? bytes : 1);
    return ETHCNN_OK;
}
extern "C" int ethcnn_host_free(ethcnn_ctx* c, void* p) {
    if (!c) return ETHCNN_ERR_ARG;
    if (p) {
        HIPCHK(c, hipDeviceSynchronize());  // a kernel may still be reading / writing it directly
        for (size_t i = 0; i < c->pinned.size(); ++i)
            if (c->pinned[i].first == (const char*)p) { c->pinned.erase(c->pinned.begin() + (long)i); break; }
        HIPCHK(c, hipHostFree(p));
    }
    return ETHCNN_OK;
}
extern "C" int ethcnn_memcpy_h2d(ethcnn_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || !dst || !src) return ETHCNN_ERR_ARG;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ETHCNN_OK;
}
extern "C" int ethcnn_memcpy_d2h(ethcnn_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || !dst || !src) return ETHCNN_ERR_ARG;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ETHCNN_OK;
}
extern "C" int ethcnn_synchronize(ethcnn_ctx* c) {
    if (!c) return ETHCNN_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, stream_sync(c));
    return ETHCNN_OK;
}

// What this GPU sustains in exact-fp32 MFMAs with nothing else issued, over about `seconds` of pure matrix work (three waves per
// SIMD, four independent accumulators each).  A calibration for reading roofline fractions: the data-sheet peak is 157.3.
extern "C" int ethcnn_measure_mfma_rate(ethcnn_ctx* c, double seconds, double* tflops) {
    if (!c || !tflops || !(seconds > 0.0) || seconds > 5.0) return c ? set_err(c, ETHCNN_ERR_ARG, "seconds must be in (0, 5]") : ETHCNN_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    c->done_armed = 0;
    float* sink = nullptr;
    HIPCHK(c, hipMalloc((void**)&sink, 4));
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipFree(sink); return set_err(c, ETHCNN_ERR_DEVICE, "cannot create HIP events"); }
    const int blocks = c->cus * 3;   // three 4-wave blocks per CU = three waves per SIMD
    auto run = [&](int iters, float* ms) -> hipError_t {
        hipError_t e = hipEventRecord(e0, c->stream);
        launch_mfma_rate(blocks, iters, sink, c->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(ms, e0, e1);
        return e;
    };
    float ms = 0.0f;
    hipError_t e = run(2000, &ms);                      // ~2 ms: sizes the real run (and ramps the clock)
    int iters = 2000;
    if (e == hipSuccess && ms > 0.0f) iters = (int)std::min(2.0e8, std::max(2000.0, 2000.0 * seconds * 1e3 / ms));
    if (e == hipSuccess) e = run(iters, &ms);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (e != hipSuccess) return set_err(c, ETHCNN_ERR_DEVICE, "MFMA rate measurement failed: %s", hipGetErrorString(e));
    *tflops = (double)blocks * 4.0 * (double)iters * 32.0 * 2048.0 / ((double)ms * 1e-3) * 1e-12;
    return ETHCNN_OK;
}
