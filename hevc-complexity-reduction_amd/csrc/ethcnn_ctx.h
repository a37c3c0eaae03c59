// ethcnn_ctx.h -- internal: the context behind the C ABI (include/ethcnn.h) and the helpers its five translation units share.
//   ethcnn_context.cpp  lifecycle, workspace, thresholds, profiling, execution-plan switches, device plumbing
//   ethcnn_model.cpp    weights: upload, the 16-bit plans' images + their accuracy guard, LSTM bundle, introspection
//   ethcnn_pass.cpp     one pass over CTUs in HBM (the kernel pipeline), the device entry point
//   ethcnn_host.cpp     host / file entry points: staging ring, worker pool, latency path, streamed pictures, sharded file driver
//   ethcnn_ldp.cpp      config #5: resi vectors, ETH-LSTM step, the per-frame LDP calls
// Mirrors /root/reference/HM-16.5_Test_AI/bin/video_to_cu_depth.py (driver) around net_CNN.py (network).  There is no CPU
// compute path in this library.
#pragma once
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
    double startup_ms[2] = {0, 0};  // ethcnn_create: its first HIP call (runtime initialisation) / the whole call

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
    // load-time accuracy guard of the 16-bit plans (ethcnn_model.cpp::check_fast_plan), per plan, reset by every weight load
    int guard_state[4] = {0, 0, 0, 0};          // 0 not evaluated, 1 accepted, 2 refused, 3 calibrating
    double guard_bound[4] = {0, 0, 0, 0};       // a-priori worst case of a probability's error (ethcnn_spec.h::fast_plan_floor_bound)
    double guard_measured[4] = {-1, -1, -1, -1}; // max |dp| vs the exact plan on the calibration picture; < 0: not measured (the bound sufficed)
    FastGuard guard_info[4] = {};
    int last_fast = 0;       // the FC1 plan of the last pass (debug_fetch reads the features of plans 1 / 2 from ws.featb)
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
    struct LdpPending { bool open = false, streamed = false; float* probs = nullptr; float* d_probs = nullptr; size_t pbytes = 0; int out = 0, in = -1, nctu = 0;
                        bool in_from_host = false; int prev_cur = -1, prev_nctu = 0; } ldp;  // (prev_*: the state resident before the step, for the give-up path)
    struct LumaPending { bool open = false, direct = false; float* probs = nullptr; size_t out_bytes = 0; } ai;  // ethcnn_predict_luma_begin ... _end
    unsigned done_seq = 0;     // last number handed out
    unsigned done_armed = 0;   // != 0: the LAST operation enqueued on the main stream stores this number when all its outputs are final
    int done_sync = 1;         // experiments build only: env ETHCNN_DONE_WORD=0: always hipStreamSynchronize (A/B runs)
    int cus = 0;             // compute units of the device
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
    // ethcnn_predict_yuv_file_sharded: the contexts of workers 1.., created on first use, destroyed with this one; a peer remembers
    // whose weights (and which load of them) it holds
    std::vector<ethcnn_ctx*> peers;
    int shard_workers = 1;             // workers of this process the host-thread budget is divided by
    const ethcnn_ctx* weights_from = nullptr;
    unsigned weights_gen = 0;          // counts weight loads of this context
    HostPool* pool = nullptr;  // created on first use by the host / file entry points
    NumaCpus numa;             // the GPU's host NUMA node (staging buffers + fill threads are placed there)
};

constexpr int kStreamCtuRows = 1024;  // row words of a streamed picture (65,536 luma rows)

// ---- shared helpers (definitions: see the file named)
int set_err(ethcnn_ctx* c, int code, const char* fmt, ...);  // ethcnn_context.cpp

#define HIPCHK(c, call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return set_err((c), ETHCNN_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// ethcnn_context.cpp
Workspace ws_view(const ethcnn_ctx* c, int p);         // the buffer set of pass parity p
int ensure_workspace(ethcnn_ctx* c, int n, int chunks);
int ensure_side_streams(ethcnn_ctx* c);  // copy_in / copy_out / s_tile: created on first pipelined use, not by ethcnn_create
hipEvent_t get_event(ethcnn_ctx* c);
// ethcnn_model.cpp
int ensure_fast_weights(ethcnn_ctx* c, int plan);
// ethcnn_pass.cpp: the measured stage of the plans' accuracy guard (max |dp| of the plan vs the exact plan on the calibration picture)
int calibrate_fast_plan(ethcnn_ctx* c, int plan, double* max_abs);
// ethcnn_pass.cpp
int make_geom(ethcnn_ctx* c, int w, int h, ptrdiff_t pitch, ptrdiff_t fstride, FrameGeom* g);
unsigned done_arm(ethcnn_ctx* c);
hipError_t stream_sync(ethcnn_ctx* c);
int serial_end(ethcnn_ctx* c);
int run_small_pass(ethcnn_ctx* c, const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, bool resi, const Workspace& w,
                   float* fc1_out, float qn, float* d_probs, int nchunks, bool pull = false, const unsigned* wait_rows = nullptr);
int run_pass(ethcnn_ctx* c, const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, int qp, float* d_probs_pass,
             hipEvent_t input_ready = nullptr);
struct Pass { long ctu0; int n; };
std::vector<Pass> plan_passes(int nctu, int nframes, int max_ctus);
// ethcnn_host.cpp
void free_staging(ethcnn_ctx* c);
int ensure_staging(ethcnn_ctx* c, size_t in_bytes, size_t out_bytes, int nbufs = 1);
bool in_pinned(const ethcnn_ctx* c, const void* p, size_t bytes);

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
