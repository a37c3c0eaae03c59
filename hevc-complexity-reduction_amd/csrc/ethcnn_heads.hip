// ethcnn_heads.hip -- the three QP-conditioned FC heads after FC1, fused in one kernel:
//   h2 = lrelu([h1, qp] W2 + b2)   (net_CNN.py:159,167,180)
//   y  = sigmoid([h2, qp] W3 + b3) (net_CNN.py:161,169,182)  + the gate predicates (:175,187)
// gfx950, v_mfma_f32_16x16x4_f32.
//
// One wave = 16 CTUs, "transposed": MFMA rows = output features, columns = CTUs.
//   FC2^T: A operand = W2 (16-k chunks by LDS-DMA, shared by the block's 4 waves, 2 stages),
//          B operand = this wave's h1 rows, one float4 per lane per chunk (element e feeds MFMA
//          step e -> k order 16c + 4g + e, the canonical FC order; head 16 fetches it straight into
//          registers a chunk ahead, heads 32 / 64 through the stage).  One block = one head of a
//          64-CTU tile (blockIdx.y = head, 16 first): short per-block latency, three times the blocks;
//          24 KB of LDS and 77 VGPRs: six blocks per CU cover each other's DMA latency.
//   The FC2^T accumulator of lane (ctu, g) holds h2[ctu][16t + 4g + r]: exactly the B operand
//   FC3^T needs for step (t, r) -- so FC2 -> FC3 chains in registers (same k order), no LDS,
//   no HBM round trip.  FC3^T's A operand (W3, 3525 floats in all) comes straight from L1/L2.
// Where the stage's time goes (phase stamps, device timeline, the rebuilt "wide" variant that was measured and dropped):
// profiles/r02_heads_timeline.txt, scripts/ubench/heads_probe.hip.  Round 6 built the leaner fetch plan for the short heads that the round-5
// timeline suggested (h1 of all chunks at block start, four W2 stages in the same 24 KB, W3 + scalars in one round trip; bit-identical):
// stage 0.150 -> 0.143 ms, 0.3 % of a C3 step, the tail of the launch unchanged -- below the 1 % bar, removed (profiles/r06_heads_short_ab.txt).
#include <hip/hip_runtime.h>

#include "ethcnn_kernels.h"
#include "ethcnn_heads_pass.h"

namespace ethcnn {

// (The tf.cond gates are applied by k5_gate behind this launch.  A form that applied them inside it, per sub-batch, by the block that
// completes the sub-batch, measured equal to 0.4 % slower and was removed in round 6; the single-launch small pass keeps that scheme:
// heads_gates_arrive in ethcnn_heads_pass.h.)
__global__ __launch_bounds__(256) void k_heads(const float* __restrict__ H1, HeadsParams hp, float qn, int N, GateIndex gi,
                                               float thr1, float thr2, float* __restrict__ H2,
                                               float* __restrict__ logits, float* __restrict__ raw,
                                               float* __restrict__ probs, int* __restrict__ flags) {
    __shared__ __attribute__((aligned(16))) float smem[kHeadsStages * kHeadsStage];  // 24 KB
    HEADS_STAMP(0);
    const int lane = threadIdx.x & 63;
    const unsigned wvu = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15;
    const int tile_ = blockIdx.x, head_ = blockIdx.y;
    const int ctu_raw = (tile_ * 4 + (int)wvu) * 16 + col;
    const bool valid = ctu_raw < N;
    const int ctu = min(ctu_raw, N - 1);  // clamped rows are loaded, never stored
    float* h2row = H2 ? H2 + (size_t)ctu * kNFc2 : nullptr;
    // only heads 64 / 32 raise predicates (blockIdx.y = 2 / 1); head 16 (most of the blocks) skips the index arithmetic
    int* fl = flags;
    if (head_ != 0) fl += 2 * gate_chunk(gi, ctu);
    // blockIdx.y selects the head: the three heads of a 64-CTU tile are independent (each reads its own
    // column slice of h1), so they run as separate blocks -- head 16 (16 K chunks) is dispatched first,
    // the short heads 32 / 64 fill in behind it.  A third of the per-block latency, three times the blocks.
    if (head_ == 0)
        head_pass<2>(smem, H1, hp, qn, lane, wvu, valid, ctu, h2row, logits, raw, probs, fl, fl + 1, thr1, thr2);
    else if (head_ == 1)
        head_pass<1>(smem, H1, hp, qn, lane, wvu, valid, ctu, h2row, logits, raw, probs, fl, fl + 1, thr1, thr2);
    else
        head_pass<0>(smem, H1, hp, qn, lane, wvu, valid, ctu, h2row, logits, raw, probs, fl, fl + 1, thr1, thr2);
    HEADS_STAMP(4);
}

void launch_heads(const Workspace& ws, const DeviceWeights& w, int n, float qn, int nctu, long ctu0, float thr1,
                  float thr2, float* d_probs, hipStream_t s) {
    HeadsParams hp;
    for (int h = 0; h < 3; ++h) {
        hp.w2[h] = w.fc2_w[h];
        hp.w2lane[h] = w.fc2_lane[h];
        hp.b2[h] = w.fc2_b[h];
        hp.w3[h] = w.fc3_w[h];
        hp.b3[h] = w.fc3_b[h];
    }
    const GateIndex gi = make_gate_index(nctu, ctu0);
    const dim3 grid((n + 63) / 64, 3);
    hipLaunchKernelGGL(k_heads, grid, dim3(256), 0, s, ws.h1, hp, qn, n, gi, thr1, thr2, ws.h2, ws.logits, ws.raw, d_probs, ws.flags);
}

}  // namespace ethcnn
