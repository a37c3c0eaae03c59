// ethcnn_kernels.h -- launch interface of the gfx950 kernels (ethcnn_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "ethcnn_spec.h"

namespace ethcnn {

// per-pass device workspace (sized for `cap` CTUs, rounded up to 16)
struct Workspace {
    int cap = 0;
    uint4* xs = nullptr;     // [cap][4][64]      u8 CTU pixels in trunk-lane order (S branch)
    uint4* xm = nullptr;     // [cap/4][8][64]    u16 2x2 pooled sums (M branch)
    uint4* xl = nullptr;     // [cap/16][8][64]   u16 4x4 pooled sums (L branch)
    float* feat = nullptr;   // [cap][2688]
    uint16_t* featb = nullptr; // plans 2 / 3 only: [cap/32][168][2][512] fp16 pieces of the features (ethcnn_spec.h, kFastPairBytes)
    float* h1 = nullptr;     // [cap][448]
    float* h2 = nullptr;     // [cap][336]
    float* logits = nullptr; // [cap][21]
    float* raw = nullptr;    // [cap][21] ungated probabilities
    int* flags = nullptr;    // the pass's gate predicates: [chunks][2] (+ 8 spare words), sync_words(chunks)
    int flags_cap = 0;       // in ints
};

struct FrameGeom {
    int width, height;
    long pitch, frame_stride;
    int cw, ch, nctu;  // CTUs per row / column / frame
};

// k0: luma frames -> xs/xm/xl for CTUs [ctu0, ctu0 + n) of the frame sequence; also clears the first
// n_flags ints of ws.flags (the pass's gate predicates; 0 = leave them alone)
// max_blocks > 0: slab-staged persistent form with at most that many blocks (one per CU when it runs beside FC1)
// wait_rows != null (streamed input, one frame): page-locked host words, one per CTU row; a group is loaded once the words of its
// CTU rows hold wait_seq (ethcnn_tile.hip, TileWait); *gave_up = wait_seq if a block waited ~1 s in vain
void launch_tile(const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, const Workspace& ws, int n_flags,
                 hipStream_t s, int max_blocks = 0, const unsigned* wait_rows = nullptr, unsigned wait_seq = 0, unsigned* gave_up = nullptr);
// k1: xs/xm/xl -> feat (fc1_plan 2: -> featb, every feature as two fp16 pieces in the 16-bit MFMA's operand order)
void launch_trunk(const Workspace& ws, const DeviceWeights& w, int n, bool resi, hipStream_t s, int fc1_plan = 0);
// k1 with the CTU-load stage folded in (A/B form, experiments build: ethcnn_trunk.hip); needs small_pass_ok-style 16-byte alignment
void launch_trunk_direct(const uint8_t* d_luma, const FrameGeom& g, long ctu0, const Workspace& ws, const DeviceWeights& w, int n, hipStream_t s);
// k1, plan 3 (ethcnn_trunk_fast.hip): the same trunk with its convolutions on the 16-bit matrix pipe (fp16 x 2 splits) -> featb in plan 2's form
void launch_trunk_f16(const Workspace& ws, const DeviceWeights& w, int n, hipStream_t s, bool ml_only = false);
// plan 3 with the CTU-load stage folded in: S tasks straight from the luma frames + the XM / XL records of the M / L tasks
// (which follow as launch_trunk_f16(..., ml_only = true)); clears the pass's n_flags sync words like launch_tile
void launch_trunk_f16_fold(const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, const Workspace& ws, const DeviceWeights& w, int n_flags,
                           hipStream_t s);
// the whole plan-3 trunk behind one pass over the frames (S, M and L tasks of a group in one block; no pixel records in HBM at all)
void launch_trunk_f16_foldall(const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, const Workspace& ws, const DeviceWeights& w, int n_flags,
                              hipStream_t s, int blocks_per_cu = 2);
// k2: feat -> h1 (bias + leaky-ReLU fused); out may be ws.h1 or a caller buffer (resi vectors)
void launch_fc1(const Workspace& ws, const DeviceWeights& w, int n, float* out, hipStream_t s);
// k2, plans 2 / 3 (ethcnn_fc1_fast.hip): featb -> h1 on the 16-bit matrix pipe: three fp16 products per fp32 product (two-way splits of
// the scaled operands), fp32 accumulate; same bias + leaky-ReLU epilogue, same h1 layout
void launch_fc1_fast(const Workspace& ws, const DeviceWeights& w, int n, float* out, int plan, hipStream_t s, int cus = 256);
// k3+k4 fused: h1 -> h2 -> logits, raw probs, probs (left ungated for launch_gate) and per-chunk gate predicates
void launch_heads(const Workspace& ws, const DeviceWeights& w, int n, float qn, int nctu_per_frame,
                  long ctu0, float thr1, float thr2, float* d_probs, hipStream_t s);
// the same heads on the 16-bit matrix pipe (plan 3; ethcnn_heads_fast.hip): fp16 x 2 splits of h1 / h2 and of W2 / W3 (w.heads16_w)
void launch_heads_f16(const Workspace& ws, const DeviceWeights& w, int n, float qn, int nctu_per_frame, long ctu0, float thr1, float thr2,
                      float* d_probs, hipStream_t s);
// ints of ws.flags a pass of `nchunks` gate sub-batches uses; ZERO on entry (the tile stage / the folded plan-3 trunk clears them)
// ---- hand-offs INSIDE one launch (the single-launch small pass, k_lstm_frame, the gate arrival counters): a producer block's results are
// read by a consumer block that may run on another XCD, i.e. behind another L2.  Shipped form ("lean"): results are stored, and read,
// with AGENT-SCOPE accesses (sc1: written through to / fetched from memory, never resident in an L2 as ordinary lines), the producer
// completes them (s_waitcnt vmcnt(0)) before the relaxed agent-scope atomic that announces them.  -DETHCNN_FULL_FENCE (A/B build,
// scripts/gpu_fence_ab.sh -> profiles/r06_handoff_fence_ab.txt) adds the two cache-maintenance instructions of LLVM's gfx942
// agent-scope release / acquire sequences -- buffer_wbl2 sc1 before the wait, buffer_inv sc1 behind the consumer's flag read --, which
// are vacuous for data that is only ever touched with sc1 accesses (the argument of DESIGN_history.md section 3b) and are measured
// there instead of argued.
#ifdef ETHCNN_FULL_FENCE
#define ETHCNN_HANDOFF_RELEASE() asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define ETHCNN_HANDOFF_ACQUIRE() asm volatile("buffer_inv sc1" ::: "memory")
#else
#define ETHCNN_HANDOFF_RELEASE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define ETHCNN_HANDOFF_ACQUIRE() \
    do {                         \
    } while (0)
#endif

inline int sync_words(int nchunks) { return 2 * nchunks + 8; }
// A small pass (<= kSmallPassMaxCtus CTUs: up to one 3840x2160 picture; 16-byte aligned rows) as ONE launch (ethcnn_small.hip): CTU load + trunk -> FC1 ->
// heads -> gates as a dataflow inside one grid (per-group / per-tile completion counters).  resi: the LDP front-end (stops
// after FC1, fc1_out = the 448-vectors).  d_sync: small_pass_sync_words(n, nchunks) ints, zeroed once; counters and flags are
// zero again when a launch has finished (each is reset by its last user), claim words keep the epoch of their last launch.
// epoch: a tag that differs from every earlier launch on this sync area and is never 0.
constexpr int kSmallPassMaxCtus = 2304;  // beyond ~2300 CTUs the grid no longer fits the GPU at once and five full launches win (measured)
bool small_pass_ok(const uint8_t* d_luma, const FrameGeom& g, int n);
int small_pass_sync_words(int n, int nchunks);
void launch_small_pass(const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, bool resi, const Workspace& ws,
                       const DeviceWeights& w, float* fc1_out, float qn, float thr1, float thr2, float* d_probs, int nchunks,
                       int* d_sync, int epoch, unsigned* done, unsigned done_seq, hipStream_t s, bool pull = false,
                       const unsigned* wait_rows = nullptr, unsigned wait_seq = 0, unsigned* gave_up = nullptr, float* host_probs = nullptr);
// host_probs (with done; page-locked host memory, 16-byte aligned): the launch's last block copies the probabilities there before it
// stores the completion word
// wait_rows (with pull; streamed input, as launch_tile's): a group is pulled once the caller has reported its CTU rows
// pull: d_luma is page-locked HOST memory; the launch's first blocks read it over PCIe into the workspace's pixel records (xs / xm /
// xl) and the trunk starts group by group as they land (ethcnn_small.hip, "PULL form")
// k5: apply the batch gates in place on d_probs
void launch_gate(const Workspace& ws, int n, int nctu_per_frame, long ctu0, float thr2, float* d_probs,
                 hipStream_t s);

// LDP: one ETH-LSTM step + heads + gates over the n CTUs of ONE frame (gates per 1024-CTU mini-batch, applied by the last
// block of the heads launch).  d_state_in may be null (zeros, i_frame <= 1).  d_gate: lstm_gate_words(n) ints -- two predicate
// words per mini-batch + a two-level ticket tree -- ZERO on entry and zero again on exit (every word is reset by its last user).
unsigned lstm_heads_blocks(int n);
int lstm_gate_words(int n);
void launch_lstm(const float* d_vec, const float* d_state_in, float* d_state_out, const float* d_lstm_blob, int n, int qp,
                 int i_frame, float thr1, float thr2, float* d_raw, float* d_probs, int* d_gate, unsigned* done, unsigned done_seq,
                 int one_launch, int epoch, hipStream_t s);
// done: completion word in page-locked host memory (null: none), stored by the frame's last heads block.  one_launch: cells and
// heads as ONE dataflow launch (k_lstm_frame; d_gate = lstm_frame_words(n) ints, zero except claim words tagged with earlier
// epochs; epoch = this launch's claim tag, never 0) instead of two launches (d_gate = lstm_gate_words(n) ints suffice)
int lstm_frame_words(int n);
// the one-launch form is used while (nearly) all of its blocks are resident at once: 28 cell blocks per 32 CTUs + 3 heads blocks
// per 16 on 2 x 256 slots -- up to a 1080p frame (544 blocks); a 2160p frame measures 10 us slower than the two launches
constexpr int kLstmOneLaunchMaxCtus = 640;

// box calibration (ethcnn_kernels.hip): pure v_mfma_f32_16x16x4_f32, blocks x 4 waves x iters x 32 MFMAs
void launch_mfma_rate(int blocks, int iters, float* d_sink, hipStream_t s);

int chunks_per_frame(int nctu);

// where the gate predicates of a pass live: chunk(c) = 1024-CTU sub-batch of the frame holding pass CTU c, counted from
// the chunk of the pass's first CTU (video_to_cu_depth.py:61-73).  32-bit arithmetic with a float reciprocal (+-1 fix-up)
// instead of two 64-bit divisions per wave.
struct GateIndex {
    int nctu, cpf, r0, c0;  // CTUs per frame, chunks per frame, ctu0 % nctu, chunk of ctu0 within its frame
    float inv_nctu;
};
__device__ __forceinline__ int gate_chunk(const GateIndex& gi, int ctu) {
    const int u = gi.r0 + ctu;  // < 2^24: exact in float
    int f = (int)((float)u * gi.inv_nctu);
    if (f * gi.nctu > u) --f;
    if ((f + 1) * gi.nctu <= u) ++f;
    return f * gi.cpf + (u - f * gi.nctu) / kSubBatch - gi.c0;
}

inline GateIndex make_gate_index(int nctu, long ctu0) {
    GateIndex gi;
    gi.nctu = nctu;
    gi.cpf = chunks_per_frame(nctu);
    gi.r0 = (int)(ctu0 % nctu);
    gi.c0 = gi.r0 / kSubBatch;
    gi.inv_nctu = 1.0f / (float)nctu;
    return gi;
}


}  // namespace ethcnn
