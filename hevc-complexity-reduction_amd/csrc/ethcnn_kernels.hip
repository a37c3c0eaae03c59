// ethcnn_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ETH-CNN path.
//
//   k0_tile   (ethcnn_tile.hip) luma frames -> zero-padded 64x64 CTUs in trunk-lane order +
//             exact integer 2x2 / 4x4 pooled sums.  HBM-bound, LDS-staged.
//   k1_trunk  (ethcnn_trunk.hip) block-mean removal + the three non-overlapping convs of all
//             21 units per CTU, register-chained MFMAs.
//   k_dense   (ethcnn_dense.hip) FC1 [N,2688]x[2688,448] + bias + leaky-ReLU (:156,164,177).
//   k_heads   (ethcnn_heads.hip) FC2 + FC3 + sigmoid of the three heads (:159-182), chained in
//             registers, and the per-sub-batch gate predicates.
//   k5_gate   tf.cond zero fill (:175,187).
//
// Arithmetic contract ("canonical order"): every dot product is a single
// k-ordered fmaf chain (what one fp32 MFMA accumulator computes) -- starting from the bias in
// the trunk convs, from 0 with the bias added afterwards in the FC layers -- leaky-ReLU =
// max(0.2f*h, h), pooling and block means come from exact integer pixel sums.  oracle/ethcnn_oracle.c (mode 0) restates exactly
// this, and the parity tests require bit-identical results.  Compile with
// -ffp-contract=off: the operation sequence below is the contract.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ethcnn_kernels.h"

namespace ethcnn {

int chunks_per_frame(int nctu) { return (nctu + kSubBatch - 1) / kSubBatch; }

// (k4: the fused FC2 + FC3 + sigmoid heads kernel lives in ethcnn_heads.hip)

// =========================================================================== k5 ======
// net_CNN.py:175  y32 = y32_tmp if any(y64 > thr1 over the fed sub-batch) else zeros
// net_CNN.py:187  y16 = y16_tmp if any(y32 > thr2) else zeros      (uses the GATED y32)
__global__ __launch_bounds__(256) void k5_gate(float* __restrict__ probs, const int* __restrict__ flags, int N,
                                               GateIndex gi, float thr2) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int i = idx / kNOut, j = idx % kNOut;
    if (i >= N || j == 0) return;
    const int chunk = gate_chunk(gi, i);
    const bool open32 = flags[2 * chunk] != 0;
    const bool open16 = open32 ? (flags[2 * chunk + 1] != 0) : (0.0f > thr2);
    if (j < 5 ? !open32 : !open16) probs[(size_t)i * kNOut + j] = 0.0f;
}

void launch_gate(const Workspace& ws, int n, int nctu, long ctu0, float thr2, float* d_probs, hipStream_t s) {
    const long total = (long)n * kNOut;
    hipLaunchKernelGGL(k5_gate, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_probs, ws.flags, n,
                       make_gate_index(nctu, ctu0), thr2);
}

// ============================================================ box calibration ======
// What THIS GPU sustains in exact-fp32 MFMAs when nothing else is issued (ethcnn_measure_mfma_rate): three waves per SIMD, four
// independent accumulators each, v_mfma_f32_16x16x4_f32 back to back.  bench.py prints it beside the roofline: boxes of one pool
// differ by several per cent in sustained clock, and a fraction of the data-sheet peak says nothing about that.
typedef float f32x4_cal __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma_rate(int iters, float* sink) {
    f32x4_cal acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const float a = 1.0f + threadIdx.x * 1e-7f, b = 0.999f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    }
    const float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (s == 123.456f) *sink = s;  // (never: keeps the chains alive)
}
// one launch: `blocks` blocks of 4 waves, `iters` x 32 MFMAs per wave; flops = blocks * 4 * iters * 32 * 2048
void launch_mfma_rate(int blocks, int iters, float* d_sink, hipStream_t s) {
    hipLaunchKernelGGL(k_mfma_rate, dim3(blocks), dim3(256), 0, s, iters, d_sink);
}

}  // namespace ethcnn
