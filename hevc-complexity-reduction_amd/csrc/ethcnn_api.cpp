// ethcnn_api.cpp -- the C ABI of libethcnn.so (include/ethcnn.h): context, weights,
// the per-pass kernel pipeline, host<->device staging and the YUV-file driver.
//
// Mirrors /root/reference/HM-16.5_Test_AI/bin/video_to_cu_depth.py (driver) around
// net_CNN.py (network).  There is no CPU compute path in this library.
#include <hip/hip_runtime.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "ethcnn_kernels.h"
#include "ethcnn_spec.h"

using namespace ethcnn;

// Persistent host worker pool for the staging fill (file pread / memcpy into pinned memory):
// one memcpy stream moves ~10 GB/s while PCIe Gen5 x16 takes ~50, and a fill lasts well under a
// millisecond, so threads are created once per context, not once per group.
// CPUs of the NUMA node the GPU hangs off.  On a two-socket host the staging path runs at 12 M CTU/s when the pinned
// buffers and the fill threads live on that node and at 8 M when they live on the other one (profiles/r02_host_copy.txt):
// the DMA engine then pulls every byte across the socket interconnect.
struct NumaCpus {
    bool valid = false;
    cpu_set_t set;  // the node's CPUs INTERSECTED with the affinity mask the process was started with (taskset, an
                    // orchestrator's CPU set): threads are never moved onto CPUs the user excluded; empty -> no pinning
};
// runs the enclosed allocations / thread start-ups on the GPU's node, then puts the caller's affinity back
class AffinityScope {
public:
    explicit AffinityScope(const NumaCpus& n) {
        if (n.valid && sched_getaffinity(0, sizeof saved_, &saved_) == 0 && sched_setaffinity(0, sizeof n.set, &n.set) == 0) on_ = true;
    }
    ~AffinityScope() {
        if (on_) (void)sched_setaffinity(0, sizeof saved_, &saved_);
    }
private:
    cpu_set_t saved_;
    bool on_ = false;
};

class HostPool {
public:
    HostPool(int nthreads, const NumaCpus& numa) {
        for (int t = 1; t < nthreads; ++t)
            workers_.emplace_back([this, numa] {
                if (numa.valid) (void)sched_setaffinity(0, sizeof numa.set, &numa.set);  // the worker, not the caller
                loop();
            });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    int size() const { return (int)workers_.size() + 1; }
    // fn(u) for u in [0, n), the caller takes part; first non-zero return wins
    int run(int n, const std::function<int(int)>& fn) {
        if (n <= 0) return 0;
        if (workers_.empty() || n == 1) {
            for (int u = 0; u < n; ++u)
                if (int r = fn(u)) return r;
            return 0;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            n_ = n;
            next_.store(0);
            rc_.store(0);
            pending_ = (int)workers_.size();
            ++gen_;
        }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
        return rc_.load();
    }

private:
    void drain() {
        for (int u = next_.fetch_add(1); u < n_ && rc_.load() == 0; u = next_.fetch_add(1))
            if (int r = (*fn_)(u)) rc_.store(r);
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            drain();
            std::lock_guard<std::mutex> lk(m_);
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<int(int)>* fn_ = nullptr;
    std::atomic<int> next_{0}, rc_{0};
    int n_ = 0, pending_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false;
};

// A 2-deep ring made the H2D of group i and the kernels of group i one serial stage of the pipeline (the host could not
// start filling group i+1 before group i-1 had completely finished): 32-35 GB/s of luma where the DMA engine alone does
// 57 (scripts/ubench/host_copy.cpp, profiles/r02_host_copy.txt).  Four buffers let fill, H2D and kernels of three
// different groups run at the same time.
constexpr int kStageBufs = 4;
constexpr int kPipelineMinCtus = 8192;  // passes below this run all their stages on the main stream (run_pass)

struct ethcnn_ctx {
    int device = 0;
    hipStream_t stream = nullptr;   // compute
    hipStream_t copy_in = nullptr;  // H2D of the next pass
    hipStream_t copy_out = nullptr; // D2H of the previous pass
    char devname[128] = {0};
    std::string err;

    bool have_weights = false;
    std::vector<float> blob;
    DeviceWeights dw;
    float* dw_arena = nullptr;

    float thr1 = 0.5f, thr2 = 0.5f;  // shipped Thr_info.txt: 0.5 x 6

    // LDP (ETH-LSTM one step): the LSTM checkpoint payload as stored, and per-frame buffers
    bool have_lstm = false;
    std::vector<float> lstm_blob;
    float* d_lstm = nullptr;
    float* d_vec = nullptr;       // [lstm_cap][448]
    float* d_state[2] = {nullptr, nullptr};  // [lstm_cap][2][448] in / out
    float* d_lprobs = nullptr;    // [lstm_cap][21]
    int lstm_cap = 0;
    int state_cur = -1;           // d_state[state_cur] = (c, h) left by the last ethcnn_ldp_step; -1 = none
    int state_nctu = 0;

    Workspace ws;
    // Cross-pass software pipeline ("pass pipeline"): the tile stage of pass i+1 (HBM-bound, no MFMA) runs
    // on a side stream beside the MFMA-bound FC1 of pass i.  What two passes in flight would share is double-buffered by pass
    // parity: the tile outputs (xs/xm/xl), h1 and the gate flags.  feat is produced and consumed on the main stream only.
    // (Heads + gate on a third stream were measured too: never co-resident with anything, profiles/r02_overlap_trace.txt.)
    uint4 *xs1 = nullptr, *xm1 = nullptr, *xl1 = nullptr;
    float* h1_1 = nullptr;
    int* flags1 = nullptr;
    hipStream_t s_tile = nullptr;
    hipEvent_t e_tile[2] = {}, e_trunk[2] = {}, e_main = nullptr;
    hipEvent_t e_fc1[2] = {};     // fast plans: "FC1 of the pass on buffer set p has finished" (the next pass's tile stage starts behind it)
    int tile_after_fc1 = 0;       // fast plans: the CTU-load stage of pass i+1 beside heads + gates of pass i instead of beside its FC1
    hipEvent_t e_band[4] = {};  // one big picture, host -> host: "the rows of piece k are in HBM" (predict_luma_latency)
    int* d_lgate = nullptr;  // LSTM heads launch: gate predicates + ticket tree (lstm_gate_words)
    int lgate_chunks = 0;    // its capacity in ints
    bool lgate_clean = false;
    int lgate_n = -1;        // frame size the area's layout was last zeroed for
    int lstm_epoch = 0;      // claim tag of the one-launch frame kernel (ethcnn_lstm.hip)
    int lstm_one_launch = 1; // cells + heads of an LDP frame as ONE dataflow launch (experiments build only: env ETHCNN_LSTM_ONE_LAUNCH=0: two launches)
    int tile_blocks = 256;   // blocks of the side-stream tile stage: one per CU (ETHCNN_TILE_BLOCKS)
    unsigned pass_idx = 0;   // parity selects the buffer set
    int last_parity = 0;     // of the last pass (debug_fetch reads its h1)
    int small_launch = 1;    // 1 = a small pass (one picture) is ONE launch (ethcnn_small.hip); 0 = tile / trunk / FC1 / heads / gate
                             // launches (ethcnn_set_small_pass_launch, experiments build only: env ETHCNN_SMALL=0)
    bool luma_over_pcie = false;  // set around a call whose luma pointer is page-locked HOST memory used in place (ethcnn_ldp_step, one
                                  // picture through ethcnn_predict_luma): the single-launch pass's direct gather reads every pixel three
                                  // times (S / M / L units) in 8-16 byte pieces -- fine in HBM, slow across PCIe -- so such a call runs the
                                  // PULL form (one coalesced read by the launch's first blocks: All-Intra pictures) or keeps the
                                  // tile-stage launch (the LDP front-end)
    int pull = 1;                 // 1 = single-launch passes over page-locked host luma pull it themselves (env ETHCNN_PULL=0, experiments
                                  // build: DMA into HBM first / tile-stage launch, the round-3 forms)
    int* d_ssync = nullptr;  // its sync area: zero between launches by construction (every word is reset by its last user)
    int ssync_cap = 0;       // in ints
    bool ssync_clean = false;
    int small_epoch = 0;     // claim tag of the last single-launch pass (1 .. 2^30, wraps: the area is re-zeroed then)
    int fc1_plan = 0;        // 0 = exact fp32 FC1 (default; bit-identical to the oracle); 2 / 3 = "fast": FC1 of the multi-launch path on the
                             // 16-bit matrix pipe with split operands, fp16 x 2 (ethcnn_fc1_fast.hip; 3: trunk and heads as well; ethcnn_set_fc1_plan, env ETHCNN_FC1_PLAN)
    uint16_t* dw_fast = nullptr;                // W1 in the form of plan 2 (packed on first use), 4.8 MB
    uint16_t* dw_trunk16 = nullptr;             // plan 3: the trunk's A operands as fp16 x 2 pieces + its per-lane constants (one allocation)
    uint16_t* dw_heads16 = nullptr;             // plan 3: FC2 / FC3 A operands as fp16 x 2 pieces (kHeads16Halves)
    int last_fast = 0;       // the FC1 plan of the last pass (debug_fetch reads the features of plans 1 / 2 from ws.featb)
    int fused = 0;           // 1 = big passes run FC1 + heads + gates as ONE launch (ethcnn_set_fused_launch, experiments build only: env ETHCNN_FUSED=1).
                             // Off by default: measured equal-to-1 % slower than the separate launches
    // completion word (page-locked host memory): the last block of a latency-path launch stores the launch's sequence number
    // there and the host spins on it instead of calling hipStreamSynchronize (~5 us sooner, scripts/ubench/launch_rtt.hip)
    unsigned* h_done = nullptr;  // (word 1: "a tile block of streamed picture <seq> gave up waiting for its rows", ethcnn_tile.hip)
    // streamed input (ethcnn_ldp_step_begin / ethcnn_rows_ready / ethcnn_ldp_step_end): one page-locked word per CTU row, holding the
    // number of the streamed picture whose rows are in the caller's buffer; the number of the NEXT streamed picture is fixed when
    // the previous one ends, so filling threads may report rows before begin has been called
    unsigned* h_rows = nullptr;
    unsigned rows_seq = 1;
    const unsigned* tile_wait_rows = nullptr;  // set around the tile launch of a streamed step
    int stream_stage_reruns = 0;               // passes of predict_luma_latency repeated because their streamed staging copy came > 1 s late
    float* host_probs = nullptr;               // set around a host -> host single-launch pass: page-locked destination its last block copies the
                                               // probabilities to (then no copy launch behind the kernel: the caller waits on the completion word)
    bool host_probs_used = false;              // ... and whether the pass took it (single-launch form, completion word armed)
    struct LdpPending { bool open = false, streamed = false; float* probs = nullptr; float* d_probs = nullptr; size_t pbytes = 0; int out = 0, in = -1, nctu = 0; } ldp;
    struct LumaPending { bool open = false, direct = false; float* probs = nullptr; size_t out_bytes = 0; } ai;  // ethcnn_predict_luma_begin ... _end
    unsigned done_seq = 0;     // last number handed out
    unsigned done_armed = 0;   // != 0: the LAST operation enqueued on the main stream stores this number when all its outputs are final
    int done_sync = 1;         // experiments build only: env ETHCNN_DONE_WORD=0: always hipStreamSynchronize (A/B runs)
    int cus = 0;             // compute units of the device
    int gate_fold = 0;       // 1 = the heads launch applies the gates itself (sub-batch arrival counters); 0 = k5_gate launch behind it.
                             // Off by default: measured equal (single pictures) to 0.4 % slower (C3) than the separate launch (experiments build only: env ETHCNN_GATE_FOLD=1)
    bool main_dirty = false; // main-stream work since e_main was last recorded (single-picture passes, LDP steps): the event is
                             // recorded lazily, by the next PIPELINED pass -- not as a barrier packet behind every small call
    int overlap = 1;         // 1 = pass pipeline on (tile stage on its own stream, beside FC1 of the previous pass);
                             // 0 = every stage on the main stream (ethcnn_set_pass_pipeline, experiments build only: env ETHCNN_OVERLAP=0)
    int max_ctus = kMaxCtusPerPass;
    int host_threads_opt = 0;  // ethcnn_options.host_threads (0 = automatic, see host_pool)
    int last_n = 0;  // CTUs of the last pass (debug_fetch)
    bool debug_capture = false;  // also store FC2 outputs, logits and ungated probabilities (1.7 KB/CTU of writes)

    int profiling = 0;  // 0 off, 1 dominant kernel (FC1) on every 3rd pass, 2 every stage
    unsigned fc1_sample = 0;
    struct Ev { hipEvent_t a, b; int stage; long ctus; };
    std::vector<Ev> pending;
    std::vector<hipEvent_t> ev_pool;
    ethcnn_stage_times times{};

    // staging ring for the host / file entry points: kStageBufs groups in flight (fill | H2D | kernels + D2H | drain)
    uint8_t* h_in[kStageBufs] = {};
    float* h_out[kStageBufs] = {};
    uint8_t* d_in[kStageBufs] = {};
    float* d_out[kStageBufs] = {};
    hipEvent_t ev_in[kStageBufs] = {}, ev_comp[kStageBufs] = {}, ev_out[kStageBufs] = {};  // created with the ring, destroyed with it
    size_t in_cap = 0, out_cap = 0;
    std::vector<std::pair<const char*, size_t>> pinned;  // ethcnn_host_alloc'ed ranges: device-addressable as they are
    HostPool* pool = nullptr;  // created on first use by the host / file entry points
    NumaCpus numa;             // the GPU's host NUMA node (staging buffers + fill threads are placed there)
};

constexpr int kStreamCtuRows = 1024;  // row words of a streamed picture (65,536 luma rows)

static thread_local std::string g_create_err;

static int set_err(ethcnn_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}

#define HIPCHK(c, call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return set_err((c), ETHCNN_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

extern "C" const char* ethcnn_version(void) { return "ethcnn-mi355x 0.1 (gfx950)"; }

extern "C" const char* ethcnn_last_error(const ethcnn_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

// ------------------------------------------------------------------ workspace -------
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
static Workspace ws_view(const ethcnn_ctx* c, int p) {
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

// `chunks`: gate sub-batches of the pass; the sync area also holds the sub-batch arrival counters and the fused launch's tile counters (sync_words)
static int ensure_workspace(ethcnn_ctx* c, int n, int chunks) {
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
    const int words = sync_words(std::max(n, w.cap), chunks);
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
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return set_err(nullptr, ETHCNN_ERR_DEVICE, "no HIP device available (%s): libethcnn has no CPU fallback",
                       e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    const int dev = opt ? opt->device : 0;
    if (dev < 0 || dev >= ndev) return set_err(nullptr, ETHCNN_ERR_ARG, "device %d out of range (0..%d)", dev, ndev - 1);
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
    if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->copy_in, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->copy_out, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->s_tile, hipStreamNonBlocking) != hipSuccess) {
        ethcnn_destroy(c);  // releases whichever streams were created
        return set_err(nullptr, ETHCNN_ERR_DEVICE, "cannot create HIP streams on device %d", dev);
    }
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
    if (const char* e = dev_env("ETHCNN_FUSED")) c->fused = std::atoi(e) != 0;      // development knob (A/B runs)
    if (const char* e = dev_env("ETHCNN_GATE_FOLD")) c->gate_fold = std::atoi(e) != 0;  // development knob (A/B runs)
    if (const char* e = dev_env("ETHCNN_DONE_WORD")) c->done_sync = std::atoi(e) != 0;  // development knob (A/B runs)
    if (const char* e = dev_env("ETHCNN_PULL")) c->pull = std::atoi(e) != 0;            // development knob (A/B runs)
    if (const char* e = dev_env("ETHCNN_TILE_AFTER_FC1")) c->tile_after_fc1 = std::atoi(e) != 0;  // development knob (A/B runs)
    if (const char* e = dev_env("ETHCNN_LSTM_ONE_LAUNCH")) c->lstm_one_launch = std::atoi(e) != 0;  // development knob (A/B runs)
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
    *out = c;
    return ETHCNN_OK;
}

static void free_staging(ethcnn_ctx* c) {
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

extern "C" int ethcnn_device_name(const ethcnn_ctx* c, char* out, size_t cap) {
    if (!c || !out || cap == 0) return ETHCNN_ERR_ARG;
    std::snprintf(out, cap, "%s", c->devname);
    return ETHCNN_OK;
}

// -------------------------------------------------------------------- weights -------
static int upload_weights(ethcnn_ctx* c) {
    HIPCHK(c, hipSetDevice(c->device));
    // one arena: trunk_w | trunk_b | fc1 image (BN 112) | fc1_b | fc2 w,b x3 | fc3 w,b x3 | fc1 image (BN 64)
    // (each 64-float aligned)
    std::vector<size_t> sizes = {(size_t)3 * kTrunkWFrags * 64, (size_t)3 * kTrunkBFrags * 64,
                                 (size_t)kNFeat * kNVec, (size_t)kNVec};
    for (int h = 0; h < 3; ++h) { sizes.push_back((size_t)(kN1[h] + 1) * kN2[h]); sizes.push_back((size_t)kN2[h]); }
    for (int h = 0; h < 3; ++h) { sizes.push_back((size_t)(kN2[h] + 1) * kN3[h]); sizes.push_back((size_t)kN3[h]); }
    sizes.push_back((size_t)kNFeat * kNVec);  // [16] fc1 image BN 64
    sizes.push_back((size_t)kNFeat * kNVec);  // [17] fc1 image BN 32
    sizes.push_back((size_t)kNFeat * kNVec);  // [18] fc1 image BN 16
    sizes.push_back((size_t)kNFeat * kNVec);  // [19] fc1 in MFMA-operand (lane) order
    for (int h = 0; h < 3; ++h) sizes.push_back((size_t)kN1[h] * kN2[h]);  // [20..22] fc2 in MFMA-operand order
    std::vector<size_t> offs;
    size_t total = 0;
    for (size_t s : sizes) { offs.push_back(total); total += (s + 63) / 64 * 64; }
    std::vector<float> host(total, 0.0f);
    const float* blob = c->blob.data();
    pack_trunk_fragments(blob, host.data() + offs[0], host.data() + offs[1]);
    {
        std::vector<float> wcat((size_t)kNFeat * kNVec);
        pack_fc1(blob, wcat.data(), host.data() + offs[3]);
        pack_fc1_image(wcat.data(), 112, 16, host.data() + offs[2]);
        pack_fc1_image(wcat.data(), 64, 32, host.data() + offs[16]);
        pack_fc1_image(wcat.data(), 32, 32, host.data() + offs[17]);
        pack_fc1_image(wcat.data(), 16, 32, host.data() + offs[18]);
        pack_fc1_lane_image(wcat.data(), host.data() + offs[19]);
    }
    for (int h = 0; h < 3; ++h) {
        std::memcpy(host.data() + offs[4 + 2 * h], blob + kOffFc2W[h], sizes[4 + 2 * h] * 4);
        std::memcpy(host.data() + offs[5 + 2 * h], blob + kOffFc2B[h], sizes[5 + 2 * h] * 4);
        std::memcpy(host.data() + offs[10 + 2 * h], blob + kOffFc3W[h], sizes[10 + 2 * h] * 4);
        std::memcpy(host.data() + offs[11 + 2 * h], blob + kOffFc3B[h], sizes[11 + 2 * h] * 4);
        pack_fc2_lane_image(blob + kOffFc2W[h], kN1[h], kN2[h], host.data() + offs[20 + h]);
    }
    if (!c->dw_arena) HIPCHK(c, hipMalloc((void**)&c->dw_arena, total * 4));
    HIPCHK(c, hipDeviceSynchronize());  // no pass in flight (on any of the streams) may still read the old arena
    HIPCHK(c, hipMemcpy(c->dw_arena, host.data(), total * 4, hipMemcpyHostToDevice));
    DeviceWeights& d = c->dw;
    d.trunk_w = c->dw_arena + offs[0];
    d.trunk_b = c->dw_arena + offs[1];
    d.fc1_img112 = c->dw_arena + offs[2];
    d.fc1_img64 = c->dw_arena + offs[16];
    d.fc1_img32 = c->dw_arena + offs[17];
    d.fc1_img16 = c->dw_arena + offs[18];
    d.fc1_lane16 = c->dw_arena + offs[19];
    for (int h = 0; h < 3; ++h) d.fc2_lane[h] = c->dw_arena + offs[20 + h];
    d.fc1_b = c->dw_arena + offs[3];
    for (int h = 0; h < 3; ++h) {
        d.fc2_w[h] = c->dw_arena + offs[4 + 2 * h];
        d.fc2_b[h] = c->dw_arena + offs[5 + 2 * h];
        d.fc3_w[h] = c->dw_arena + offs[10 + 2 * h];
        d.fc3_b[h] = c->dw_arena + offs[11 + 2 * h];
    }
    d.fc1_fast = nullptr;  // the fast plans' images of W1 belong to the previous weights: repacked on the next such pass
    d.trunk16_w = nullptr;
    d.heads16_w = nullptr;
    d.trunk16_c = nullptr;
    c->have_weights = true;
    return ETHCNN_OK;
}

// plan 2: W1 as fp16 x 2 pieces in the MFMA's B-operand order (ethcnn_weights.cpp::pack_fc1_fast_image), once per weight load
static int ensure_fast_weights(ethcnn_ctx* c, int plan) {
    if (plan == 3) {  // plan 3 = plan 2's FC1 + the trunk's convolutions as fp16 x 2 (ethcnn_trunk_fast.hip)
        int rc = ensure_fast_weights(c, 2);
        if (rc || c->dw.trunk16_w) return rc;
        const size_t wbytes = (size_t)3 * kTrunk16Halves * 2, cbytes = (size_t)3 * kTrunk16Consts * 4;
        std::vector<uint16_t> wimg((size_t)3 * kTrunk16Halves);
        std::vector<float> cimg((size_t)3 * kTrunk16Consts);
        pack_trunk_f16(c->blob.data(), c->dw.fast_scale_a, wimg.data(), cimg.data(), &c->dw.trunk16_s);
        if (!c->dw_trunk16) HIPCHK(c, hipMalloc((void**)&c->dw_trunk16, wbytes + cbytes));
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, hipMemcpy(c->dw_trunk16, wimg.data(), wbytes, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(reinterpret_cast<char*>(c->dw_trunk16) + wbytes, cimg.data(), cbytes, hipMemcpyHostToDevice));
        c->dw.trunk16_w = c->dw_trunk16;
        c->dw.trunk16_c = reinterpret_cast<float*>(reinterpret_cast<char*>(c->dw_trunk16) + wbytes);
        // ... and the heads' FC2 / FC3 (ethcnn_heads_fast.hip): scales from guaranteed bounds; degenerate weights (a zero / non-finite
        // bound) keep the exact heads
        std::vector<uint16_t> himg((size_t)kHeads16Halves);
        if (pack_heads_f16(c->blob.data(), fast_feature_bound(c->blob.data()), himg.data(), &c->dw.heads16_s)) {
            if (!c->dw_heads16) HIPCHK(c, hipMalloc((void**)&c->dw_heads16, (size_t)kHeads16Halves * 2));
            HIPCHK(c, hipMemcpy(c->dw_heads16, himg.data(), (size_t)kHeads16Halves * 2, hipMemcpyHostToDevice));
            c->dw.heads16_w = c->dw_heads16;
        }
        return 0;
    }
    if (c->dw.fc1_fast) return 0;
    const size_t n16 = (size_t)kNFeat * kNVec * fast_pieces(plan);
    std::vector<float> wcat((size_t)kNFeat * kNVec), b1(kNVec);
    pack_fc1(c->blob.data(), wcat.data(), b1.data());
    {   // the feature order of the plans is a table of the trunk's register order: it must be a permutation of 0 .. 2687
        std::vector<char> seen(kNFeat, 0);
        for (int ch = 0; ch < kFastChunks; ++ch)
            for (int s8 = 0; s8 < 16; ++s8) {
                const int k = fast_feature_k(ch, s8 >> 3, s8 & 7);
                if (k < 0 || k >= kNFeat || seen[k]) return set_err(c, ETHCNN_ERR_ARG, "internal: fast FC1 feature order is not a permutation (chunk %d)", ch);
                seen[k] = 1;
            }
    }
    float sw = 1.0f;
    {
        // powers of two that put the largest possible |feature| and the largest |weight| at <= 2^14 (fp16 overflows at 65504): the
        // feature bound is a guarantee derived from the conv weights (|input| <= 1), not an observation
        float wmax = 0.0f;
        for (float v : wcat) wmax = std::max(wmax, std::fabs(v));
        const float fmax = fast_feature_bound(c->blob.data());
        if (!(wmax > 0.0f) || !(fmax > 0.0f) || !std::isfinite(wmax) || !std::isfinite(fmax))
            return set_err(c, ETHCNN_ERR_ARG, "FC1 plan 2 needs finite, non-zero weights (max |W1| %g, feature bound %g)", (double)wmax, (double)fmax);
        sw = std::exp2f(14.0f - std::ceil(std::log2(wmax)));
        c->dw.fast_scale_w = sw;
        c->dw.fast_scale_a = std::exp2f(14.0f - std::ceil(std::log2(fmax)));
    }
    std::vector<uint16_t> img(n16);
    pack_fc1_fast_image(wcat.data(), plan, sw, img.data());
    if (!c->dw_fast) HIPCHK(c, hipMalloc((void**)&c->dw_fast, n16 * 2));
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(c->dw_fast, img.data(), n16 * 2, hipMemcpyHostToDevice));
    c->dw.fc1_fast = c->dw_fast;
    return 0;
}

extern "C" int ethcnn_load_blob(ethcnn_ctx* c, const float* blob, size_t nfloats) {
    if (!c || !blob) return ETHCNN_ERR_ARG;
    if (nfloats != kBlobFloats) return set_err(c, ETHCNN_ERR_ARG, "blob must hold %zu floats, got %zu", kBlobFloats, nfloats);
    c->blob.assign(blob, blob + nfloats);
    return upload_weights(c);
}

extern "C" int ethcnn_load_synthetic(ethcnn_ctx* c, uint64_t seed, double head_gain) {
    if (!c) return ETHCNN_ERR_ARG;
    c->blob.resize(kBlobFloats);
    synth_blob(seed, head_gain, c->blob.data());
    return upload_weights(c);
}

extern "C" int ethcnn_load_checkpoint(ethcnn_ctx* c, const char* prefix) {
    if (!c || !prefix) return ETHCNN_ERR_ARG;
    std::vector<float> blob(kBlobFloats);
    char err[400];
    const int rc = ckpt_load_blob(prefix, blob.data(), err, sizeof err);
    if (rc) return set_err(c, rc, "%s", err);
    c->blob.swap(blob);
    return upload_weights(c);
}

extern "C" int ethcnn_get_blob(const ethcnn_ctx* c, float* out, size_t nfloats) {
    if (!c || !out || nfloats != kBlobFloats || !c->have_weights) return ETHCNN_ERR_ARG;
    std::memcpy(out, c->blob.data(), nfloats * 4);
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
static hipEvent_t get_event(ethcnn_ctx* c) {
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) e = nullptr;
    return e;
}
struct StageTimer {
    ethcnn_ctx* c;
    int stage;
    hipEvent_t a = nullptr, b = nullptr;
    bool on;
    long ctus;
    hipStream_t stream;  // the stream the stage is launched on (HIP events see only their own stream)
    StageTimer(ethcnn_ctx* c_, int st, long n = 0, hipStream_t s = nullptr) : c(c_), stage(st), ctus(n), stream(s ? s : c_->stream) {
        on = c->profiling >= 2 || (c->profiling == 1 && st == ETHCNN_STAGE_FC1 && (c->fc1_sample++ % 3) == 0);
        if (on) {
            a = get_event(c);
            b = get_event(c);
            if (!a || !b || hipEventRecord(a, stream) != hipSuccess) fail();
        }
    }
    // a timing failure never fails the pass; it is counted (ethcnn_stage_times.timing_errors) so a
    // reader of the stage times knows the sample is incomplete
    void fail() {
        on = false;
        c->times.timing_errors++;
        if (a) c->ev_pool.push_back(a);
        if (b) c->ev_pool.push_back(b);
        a = b = nullptr;
    }
    ~StageTimer() {
        if (on && hipEventRecord(b, stream) != hipSuccess) fail();
        if (on) {
            c->pending.push_back({a, b, stage, ctus});
            c->times.timed[stage]++;
            c->times.timed_ctus[stage] += ctus;
        }
        c->times.launches[stage]++;
    }
};
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

extern "C" int ethcnn_set_fused_launch(ethcnn_ctx* c, int on) {
    if (!c) return ETHCNN_ERR_ARG;
    c->fused = (on == 1);       // takes effect with the next pass enqueued; results do not depend on it
    c->gate_fold = (on == 2);
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

// ------------------------------------------------------------------- pipeline -------
static int make_geom(ethcnn_ctx* c, int w, int h, ptrdiff_t pitch, ptrdiff_t fstride, FrameGeom* g) {
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
static unsigned done_arm(ethcnn_ctx* c) {
    if (!c->done_sync || !c->h_done) return 0;
    if (++c->done_seq == 0) ++c->done_seq;
    return c->done_seq;
}
// wait for everything enqueued on the main stream: through the completion word when the last enqueued launch carries one
// (bounded: a launch that never reports -- a device fault -- falls through to hipStreamSynchronize, which returns the error)
static hipError_t stream_sync(ethcnn_ctx* c) {
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

static int serial_end(ethcnn_ctx* c) {
    c->main_dirty = true;  // the next pipelined tile stage records e_main behind this work and waits for it
    return 0;
}

// the single-launch form of a small pass (ethcnn_small.hip); fc1_out: ws.h1 (All-Intra) or the caller's vectors (resi).
// Asynchronous on the main stream.
static int run_small_pass(ethcnn_ctx* c, const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, bool resi, const Workspace& w,
                          float* fc1_out, float qn, float* d_probs, int nchunks, bool pull = false, const unsigned* wait_rows = nullptr) {
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
static int run_pass(ethcnn_ctx* c, const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, int qp,
                    float* d_probs_pass, hipEvent_t input_ready = nullptr) {
    const int cpf = chunks_per_frame(g.nctu);
    const long nchunks = (ctu0 + n - 1) / g.nctu * cpf + ((ctu0 + n - 1) % g.nctu) / kSubBatch + 1 -
                         (ctu0 / g.nctu * cpf + (ctu0 % g.nctu) / kSubBatch);
    c->done_armed = 0;
    int rc = ensure_workspace(c, n, (int)nchunks);
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
    // the tile stage also zeroes the pass's sync area (gate predicates, sub-batch arrival counters, tile completion counters of the fused launch)
    if (fold3) {
        // (no tile launch; the folded trunk below clears the sync area itself)
    } else if (fold) {  // no tile launch: only the pass's sync area is cleared (what the tile stage does on the way)
        HIPCHK(c, hipMemsetAsync(w.flags, 0, (size_t)sync_words(n, (int)nchunks) * sizeof(int), s_tile));
    } else {
        StageTimer t(c, ETHCNN_STAGE_TILE, n, s_tile);
        launch_tile(d_luma, g, ctu0, n, w, sync_words(n, (int)nchunks), s_tile, side_tile ? c->tile_blocks : 0,
                    side_tile ? nullptr : c->tile_wait_rows, c->rows_seq, c->h_done + 1);  // (streamed input: a single main-stream pass)
    }
    LAUNCH_OK("tile");
    if (side_tile) {
        HIPCHK(c, hipEventRecord(c->e_tile[p], s_tile));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->e_tile[p], 0));
    }
    const int fast = c->fc1_plan;  // FC1 plans 1 / 2: trunk -> 16-bit feature pieces -> FC1 on the 16-bit matrix pipe
    if (fast && (rc = ensure_fast_weights(c, fast)) != 0) return rc;
    { StageTimer t(c, ETHCNN_STAGE_TRUNK);
      if (fold) launch_trunk_direct(d_luma, g, ctu0, w, c->dw, n, c->stream);
      else if (fold3 && fold3_knob == 1) {
          launch_trunk_f16_fold(d_luma, g, ctu0, n, w, c->dw, sync_words(n, (int)nchunks), c->stream);
          launch_trunk_f16(w, c->dw, n, c->stream, /*ml_only=*/true);
      } else if (fold3) {
          static const int bpc = [] { const char* e = dev_env("ETHCNN_PLAN3_FOLD_BLOCKS"); return e ? std::atoi(e) : 2; }();
          launch_trunk_f16_foldall(d_luma, g, ctu0, n, w, c->dw, sync_words(n, (int)nchunks), c->stream, bpc);
      } else if (fast == 3) launch_trunk_f16(w, c->dw, n, c->stream);
      else launch_trunk(w, c->dw, n, false, c->stream, fast); }
    LAUNCH_OK("trunk");
    if (side_tile) HIPCHK(c, hipEventRecord(c->e_trunk[p], c->stream));
    Workspace wv = w;
    if (!c->debug_capture) wv.h2 = wv.logits = wv.raw = nullptr;
    if (fast) {
        { StageTimer t(c, ETHCNN_STAGE_FC1, n); launch_fc1_fast(w, c->dw, n, w.h1, fast == 3 ? 2 : fast, c->stream, c->cus); }
        LAUNCH_OK("FC1 (16-bit pipe)");
        if (side_tile && c->tile_after_fc1) HIPCHK(c, hipEventRecord(c->e_fc1[p], c->stream));
        // plan 3: the heads on the 16-bit pipe as well (experiments build: ETHCNN_PLAN3_HEADS=0 keeps the exact heads for the A/B)
        static const bool heads16_knob = [] { const char* e = dev_env("ETHCNN_PLAN3_HEADS"); return !e || std::atoi(e) != 0; }();
        { StageTimer t(c, ETHCNN_STAGE_HEADS);
          if (fast == 3 && heads16_knob && c->dw.heads16_w)
              launch_heads_f16(wv, c->dw, n, qn, g.nctu, ctu0, c->thr1, c->thr2, d_probs_pass, c->stream, c->gate_fold ? (int)nchunks : 0);
          else launch_heads(wv, c->dw, n, qn, g.nctu, ctu0, c->thr1, c->thr2, d_probs_pass, c->stream, c->gate_fold ? (int)nchunks : 0); }
        if (!c->gate_fold) { StageTimer t(c, ETHCNN_STAGE_GATE); launch_gate(w, n, g.nctu, ctu0, c->thr2, d_probs_pass, c->stream); }
        LAUNCH_OK("heads / gate");
    } else if (c->fused && fc1_heads_fusable(n)) {
        // big pass: FC1, the heads and the gates are ONE launch (ethcnn_fused.hip); its time is booked under the FC1 stage
        { StageTimer t(c, ETHCNN_STAGE_FC1, n); launch_fc1_heads(wv, c->dw, n, qn, g.nctu, ctu0, c->thr1, c->thr2, d_probs_pass, (int)nchunks, c->stream); }
        LAUNCH_OK("fused FC1 + heads + gate");
    } else {
        { StageTimer t(c, ETHCNN_STAGE_FC1, n); launch_fc1(w, c->dw, n, w.h1, c->stream); }
        LAUNCH_OK("FC1");
        { StageTimer t(c, ETHCNN_STAGE_HEADS); launch_heads(wv, c->dw, n, qn, g.nctu, ctu0, c->thr1, c->thr2, d_probs_pass, c->stream, c->gate_fold ? (int)nchunks : 0); }
        if (!c->gate_fold) { StageTimer t(c, ETHCNN_STAGE_GATE); launch_gate(w, n, g.nctu, ctu0, c->thr2, d_probs_pass, c->stream); }
        LAUNCH_OK("heads / gate");
    }
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
struct Pass { long ctu0; int n; };
static std::vector<Pass> plan_passes(int nctu, int nframes, int max_ctus) {
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

// `nbufs` of the ring are needed by the caller (the single-frame LDP / resi entry points use one)
static int ensure_staging(ethcnn_ctx* c, size_t in_bytes, size_t out_bytes, int nbufs = 1) {
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
        int nt = ethcnn_host_thread_budget(local_workers(), 0);
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
    (void)hipStreamSynchronize(c->s_tile);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->copy_out);
    return result;
}

// [p, p + bytes) inside a buffer from ethcnn_host_alloc: page-locked, DMA-able (and device-addressable) as it is
static bool in_pinned(const ethcnn_ctx* c, const void* p, size_t bytes);

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
        return set_err(c, ETHCNN_ERR_DEVICE, "ethcnn_predict_luma_end: the kernels waited 1 s for luma rows that were never reported (ethcnn_rows_ready)");
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

// ------------------------------------------------ config #5: ETH-LSTM one step -------
static int upload_lstm(ethcnn_ctx* c) {
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_lstm) HIPCHK(c, hipMalloc((void**)&c->d_lstm, (kLstmBlobFloats + kLstmPackFloats) * sizeof(float)));
    std::vector<float> pack(kLstmPackFloats);  // the LSTMCell kernels in the cell kernel's load order, behind the blob
    pack_lstm_kernels(c->lstm_blob.data(), pack.data());
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpyAsync(c->d_lstm, c->lstm_blob.data(), kLstmBlobFloats * sizeof(float), hipMemcpyHostToDevice,
                             c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_lstm + kLstmBlobFloats, pack.data(), kLstmPackFloats * sizeof(float), hipMemcpyHostToDevice,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_lstm = true;
    return ETHCNN_OK;
}

extern "C" int ethcnn_load_lstm_blob(ethcnn_ctx* c, const float* blob, size_t nfloats) {
    if (!c || !blob) return ETHCNN_ERR_ARG;
    if (nfloats != kLstmBlobFloats)
        return set_err(c, ETHCNN_ERR_ARG, "LSTM blob must hold %zu floats, got %zu", kLstmBlobFloats, nfloats);
    c->lstm_blob.assign(blob, blob + nfloats);
    return upload_lstm(c);
}

extern "C" int ethcnn_load_lstm_synthetic(ethcnn_ctx* c, uint64_t seed, double head_gain) {
    if (!c) return ETHCNN_ERR_ARG;
    c->lstm_blob.resize(kLstmBlobFloats);
    synth_lstm_blob(seed, head_gain, c->lstm_blob.data());
    return upload_lstm(c);
}

extern "C" int ethcnn_load_lstm_checkpoint(ethcnn_ctx* c, const char* prefix) {
    if (!c || !prefix) return ETHCNN_ERR_ARG;
    std::vector<float> blob(kLstmBlobFloats);
    char err[400];
    const int rc = ckpt_load_table(prefix, kLstmTensors, kNumLstmTensors, blob.data(), err, sizeof err);
    if (rc) return set_err(c, rc, "%s", err);
    c->lstm_blob.swap(blob);
    return upload_lstm(c);
}

extern "C" int ethcnn_get_lstm_blob(const ethcnn_ctx* c, float* out, size_t nfloats) {
    if (!c || !out || nfloats != kLstmBlobFloats || !c->have_lstm) return ETHCNN_ERR_ARG;
    std::memcpy(out, c->lstm_blob.data(), nfloats * 4);
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

// [p, p + bytes) inside a buffer from ethcnn_host_alloc: page-locked and mapped at the same address on the device (unified
// addressing), so kernels can read / write it in place -- one PCIe crossing, no staging copy, no copy launch
static bool in_pinned(const ethcnn_ctx* c, const void* p, size_t bytes) {
    for (const auto& r : c->pinned)
        if ((const char*)p >= r.first && (const char*)p + bytes <= r.first + r.second) return true;
    return false;
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
        // computed on rows that never arrived: the output state is garbage, the INPUT state (the other buffer) is untouched and stays the
        // resident one, so the caller may run the frame again (ethcnn_ldp_step on the by now complete buffer) with the same arguments
        c->state_cur = c->ldp.in;
        return set_err(c, ETHCNN_ERR_DEVICE, "ethcnn_ldp_step_end: the kernels waited 1 s for luma rows that were never reported (ethcnn_rows_ready)");
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

extern "C" int ethcnn_set_debug_capture(ethcnn_ctx* c, int on) {
    if (!c) return ETHCNN_ERR_ARG;
    c->debug_capture = (on != 0);
    return ETHCNN_OK;
}

extern "C" int ethcnn_debug_fetch(ethcnn_ctx* c, int which, float* out, size_t nfloats) {
    if (!c || !out) return ETHCNN_ERR_ARG;
    if (which >= ETHCNN_DBG_FC2 && which <= ETHCNN_DBG_RAW_PROBS && !c->debug_capture)
        return set_err(c, ETHCNN_ERR_ARG, "debug_fetch(%d): call ethcnn_set_debug_capture(ctx, 1) before the pass", which);
    const float* src = nullptr;
    size_t per = 0;
    switch (which) {
        case ETHCNN_DBG_FEATURES: src = c->ws.feat; per = kNFeat; break;
        case ETHCNN_DBG_FC1: src = ws_view(c, c->last_parity).h1; per = kNVec; break;
        case ETHCNN_DBG_FC2: src = c->ws.h2; per = kNFc2; break;
        case ETHCNN_DBG_LOGITS: src = c->ws.logits; per = kNOut; break;
        case ETHCNN_DBG_RAW_PROBS: src = c->ws.raw; per = kNOut; break;
        default: return set_err(c, ETHCNN_ERR_ARG, "unknown debug tensor %d", which);
    }
    if (!src || nfloats > (size_t)c->last_n * per) return set_err(c, ETHCNN_ERR_ARG, "debug_fetch: last pass had %d CTUs", c->last_n);
    if (which != ETHCNN_DBG_FEATURES) return ethcnn_memcpy_d2h(c, out, src, nfloats * 4);
    if (c->last_fast) {
        // plans 2 / 3: the trunk left every feature as two fp16 pieces: add them back, (h0 + h1) / scale (plan 2: equal to the feature to
        // 2^-24 relative)
        const int plan = c->last_fast == 3 ? 2 : c->last_fast, np = fast_pieces(plan);  // (plan 3 writes plan 2's form)
        const size_t n = (nfloats + kNFeat - 1) / kNFeat, pairs = (n + 31) / 32;
        std::vector<uint16_t> rawb(pairs * (size_t)(fast_pair_bytes(plan) / 2));
        int rc = ethcnn_memcpy_d2h(c, rawb.data(), c->ws.featb, rawb.size() * 2);
        if (rc) return rc;
        auto hf = [](uint16_t h) { return f16_f32(h); };
        const float inv = 1.0f / c->dw.fast_scale_a;
        for (size_t row = 0; row * kNFeat < nfloats; ++row)
            for (int ch = 0; ch < kFastChunks; ++ch)
                for (int kh = 0; kh < 2; ++kh)
                    for (int idx = 0; idx < 8; ++idx) {
                        const size_t o = row * kNFeat + (size_t)fast_feature_k(ch, kh, idx);
                        if (o >= nfloats) continue;
                        const uint16_t* rec = rawb.data() + ((row / 32) * kFastChunks + ch) * np * 512 + (kh * 32 + row % 32) * 8 + idx;
                        out[o] = (hf(rec[0]) + hf(rec[512])) * inv;
                    }
        return ETHCNN_OK;
    }
    // features live as [group of 16 CTUs][k/4][16][4] (ethcnn_dense.hip); hand back [n][2688]
    const size_t n = (nfloats + kNFeat - 1) / kNFeat, groups = (n + 15) / 16;
    std::vector<float> rawf(groups * 16 * kNFeat);
    int rc = ethcnn_memcpy_d2h(c, rawf.data(), src, rawf.size() * 4);
    if (rc) return rc;
    for (size_t i = 0; i < nfloats; ++i) {
        const size_t row = i / kNFeat, k = i % kNFeat;
        out[i] = rawf[((row / 16) * (kNFeat / 4) + k / 4) * 64 + (row % 16) * 4 + (k % 4)];
    }
    return ETHCNN_OK;
}
