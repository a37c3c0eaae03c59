// ethcnn_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ETH-CNN path.
//
//   k0_tile   (ethcnn_tile.hip) luma frames -> zero-padded 64x64 CTUs in trunk-lane order +
//             exact integer 2x2 / 4x4 pooled sums.  HBM-bound, LDS-staged.
//   k1_trunk  block-mean removal (net_CNN.py:78-84) + the three non-overlapping convs
//             (:86-92,127-141) of all 21 units per CTU, written as `h_conv_flat` (:143-150).
//             v_mfma_f32_16x16x4_f32, "transposed" (rows = output channels, columns = 16
//             units) so each layer's accumulator registers ARE the next layer's B operand:
//             no LDS, no cross-lane traffic between layers.
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

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float lrelu(float h) { return fmaxf(0.2f * h, h); }

int chunks_per_frame(int nctu) { return (nctu + kSubBatch - 1) / kSubBatch; }

// =========================================================================== k1 ======
// One wave = one task = the SAME unit position of 16 consecutive CTUs (a group): 16 S tasks,
// 4 M tasks and 1 L task per group.  lane = col + 16 g: col = CTU within the group (MFMA
// column), g = MFMA k-group.  Features are written as feat[group][k/4][16][4] so that every
// store here and every FC1 operand load is a full 256-byte run per k-group.
// MFMA D[row][col] (row = output channel) lives in lane (col, g) as rows 4g..4g+3, which is
// exactly B[k = g][col] for the 4 k-steps r = 0..3 of the next layer when that layer's K
// is enumerated as (patch, r, g) with ci = 4g + r.  240 MFMAs per task, 0 LDS bytes.
template <bool RESI>
__device__ __forceinline__ float px_value(int s, int cnt) {
    if (RESI) return ((float)(s - 128 * cnt) / 255.0f) * 10.0f;  // (x-128)/255.0*10, LSTM net :153
    return (float)s * (1.0f / 255.0f);                           // x * 1/255, net_CNN.py:105
}

template <int BR, bool RESI>
__device__ __forceinline__ void trunk_tasks(const uint4* __restrict__ X, int ntasks, int wave, int nwaves,
                                            const float* __restrict__ wfrag, const float* __restrict__ bfrag,
                                            float* __restrict__ F, int N) {
    const int lane = threadIdx.x & 63;
    const int col = lane & 15, g = lane >> 4;
    if (wave >= ntasks) return;

    // weights of this branch -> registers (A operands), once per wave
    float A1[4], A2[2][16], A3[2][24], B1[4], B2[2][4], B3[2][4];
    {
        const float* wf = wfrag + (size_t)BR * kTrunkWFrags * 64 + lane;
#pragma unroll
        for (int s = 0; s < 4; ++s) A1[s] = wf[s * 64];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 16; ++s) A2[t][s] = wf[(4 + t * 16 + s) * 64];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 24; ++s) A3[t][s] = wf[(36 + t * 24 + s) * 64];
        const float* bf = bfrag + (size_t)BR * kTrunkBFrags * 64 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) B1[r] = bf[r * 64];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                B2[t][r] = bf[(4 + t * 4 + r) * 64];
                B3[t][r] = bf[(12 + t * 4 + r) * 64];
            }
    }
    constexpr int POOL = (BR == 0) ? 1 : (BR == 1 ? 2 : 4);
    constexpr float SCALE = 1.0f / (float)(POOL * POOL);
    constexpr int NB = (BR == 0) ? 4 : (BR == 1 ? 2 : 1);
    constexpr int OFF2 = (BR == 0) ? 672 : (BR == 1 ? 2208 : 2592);
    constexpr int OFF3 = (BR == 0) ? 0 : (BR == 1 ? 512 : 640);

    // pixel records are prefetched one task ahead: the raw registers are dead as soon as they
    // are decoded, so the next task's loads fly under this task's 240 MFMAs at no VGPR cost
    constexpr int NJ = (BR == 0) ? 4 : 8;
    uint4 raw[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) raw[j] = X[((size_t)wave * NJ + j) * 64 + lane];

    for (int task = wave; task < ntasks; task += nwaves) {
        // ---- pixels: x[d][kx], d = 4 q2 + q1 (patch), this lane's row g of each patch
        float x[16][4];
        int T = 0;
        if (BR == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4 d = raw[j];
                const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int q1 = 0; q1 < 4; ++q1) {
                    T = (int)__builtin_amdgcn_udot4(w[q1], 0x01010101u, (unsigned)T, false);  // exact byte sum
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const int s = (int)((w[q1] >> (8 * kx)) & 0xff);
                        x[4 * j + q1][kx] = RESI ? px_value<true>(s, 1) : (float)s;
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint4 d = raw[j];
                const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) T += (int)((w[i] & 0xffffu) + (w[i] >> 16));
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const int s = (int)((w[2 * hh + (kx >> 1)] >> (16 * (kx & 1))) & 0xffff);
                        x[2 * j + hh][kx] = RESI ? px_value<true>(s, POOL * POOL) * SCALE : (float)s;
                    }
            }
        }
        if (task + nwaves < ntasks) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) raw[j] = X[((size_t)(task + nwaves) * NJ + j) * 64 + lane];
        }
        T += __shfl_xor(T, 16);
        T += __shfl_xor(T, 32);
        // canonical centring: AI  v = fma(float(sum), c255 * 2^-p, -mean)  (one rounding);
        //                                 resi v = x - mean with x = ((s - 128 cnt) / 255 * 10) * 2^-p
        const float mean = px_value<RESI>(T, 256 * POOL * POOL) * (SCALE * (1.0f / 256.0f));
        const float negmean = -mean;
        constexpr float C255S = (1.0f / 255.0f) * SCALE;  // exact: SCALE is a power of two

        // ---- where this column's outputs go
        int grp, by, bx;  // wave-uniform
        if (BR == 0) { grp = task >> 4; by = (task >> 2) & 3; bx = task & 3; }
        else if (BR == 1) { grp = task >> 2; by = (task >> 1) & 1; bx = task & 1; }
        else { grp = task; by = 0; bx = 0; }
        const bool valid = grp * 16 + col < N;
        // feature k of this lane's CTU: Fg[(k/4) * 64 + (k%4)]  (k % 4 == 0 for every f32x4 below)
        float* Fg = F + (size_t)grp * kNFeat * 16 + col * 4;

        f32x4 a2[4][2];
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
            // conv1: 4 patches (q1) x 4 k-steps (s = kx); lane supplies v[patch][ky=g][kx=s]
            // accumulators start at the bias (MFMA C operand): one rounding chain, no separate add
            f32x4 c1[4];
#pragma unroll
            for (int q1 = 0; q1 < 4; ++q1) c1[q1] = (f32x4){B1[0], B1[1], B1[2], B1[3]};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q1 = 0; q1 < 4; ++q1) {
                    const float xv = x[4 * q2 + q1][s];
                    c1[q1] = MFMA16(A1[s], RESI ? xv - mean : fmaf(xv, C255S, negmean), c1[q1]);
                }
#pragma unroll
            for (int q1 = 0; q1 < 4; ++q1)
#pragma unroll
                for (int r = 0; r < 4; ++r) c1[q1][r] = lrelu(c1[q1][r]);
            // conv2: K = (q1, r, g) with ci = 4g + r; two M tiles (channels 0-15, 16-23 + pad)
            f32x4 c2[2] = {(f32x4){B2[0][0], B2[0][1], B2[0][2], B2[0][3]}, (f32x4){B2[1][0], B2[1][1], B2[1][2], B2[1][3]}};
#pragma unroll
            for (int q1 = 0; q1 < 4; ++q1)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    c2[0] = MFMA16(A2[0][4 * q1 + r], c1[q1][r], c2[0]);
                    c2[1] = MFMA16(A2[1][4 * q1 + r], c1[q1][r], c2[1]);
                }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) c2[t][r] = lrelu(c2[t][r]);
            a2[q2][0] = c2[0];
            a2[q2][1] = c2[1];
            if (valid) {
                const int slot = (2 * by + (q2 >> 1)) * (2 * NB) + 2 * bx + (q2 & 1);
                const int k0 = OFF2 + slot * 24 + 4 * g;
                *reinterpret_cast<f32x4*>(Fg + (k0 >> 2) * 64) = c2[0];
                if (g < 2) *reinterpret_cast<f32x4*>(Fg + ((k0 + 16) >> 2) * 64) = c2[1];
            }
        }
        // conv3: phase A = channels 0..15 of the 4 positions, phase B = channels 16..23 with
        // positions (2j, 2j+1) packed into the lower / upper lane halves.
        f32x4 c3[2] = {(f32x4){B3[0][0], B3[0][1], B3[0][2], B3[0][3]}, (f32x4){B3[1][0], B3[1][1], B3[1][2], B3[1][3]}};
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c3[0] = MFMA16(A3[0][4 * q2 + r], a2[q2][0][r], c3[0]);
                c3[1] = MFMA16(A3[1][4 * q2 + r], a2[q2][0][r], c3[1]);
            }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float hi = __shfl(a2[2 * j + 1][1][r], lane & 31);  // lanes 32..63 <- lanes 0..31 of position 2j+1
                const float z = (lane < 32) ? a2[2 * j][1][r] : hi;
                c3[0] = MFMA16(A3[0][16 + 4 * j + r], z, c3[0]);
                c3[1] = MFMA16(A3[1][16 + 4 * j + r], z, c3[1]);
            }
        if (valid) {
            const int k0 = OFF3 + (by * NB + bx) * 32 + 4 * g;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = lrelu(c3[t][r]);
                *reinterpret_cast<f32x4*>(Fg + ((k0 + 16 * t) >> 2) * 64) = o;
            }
        }
    }
}

template <bool RESI>
__global__ __launch_bounds__(256) void k1_trunk(const uint4* __restrict__ XS, const uint4* __restrict__ XM,
                                                const uint4* __restrict__ XL, int N, int bS, int bM,
                                                const float* __restrict__ wfrag, const float* __restrict__ bfrag,
                                                float* __restrict__ F) {
    const int w = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const int groups = (N + 15) / 16;
    if (b < bS) trunk_tasks<0, RESI>(XS, groups * 16, b * 4 + w, bS * 4, wfrag, bfrag, F, N);
    else if (b < bS + bM) trunk_tasks<1, RESI>(XM, groups * 4, (b - bS) * 4 + w, bM * 4, wfrag, bfrag, F, N);
    else trunk_tasks<2, RESI>(XL, groups, (b - bS - bM) * 4 + w, (int)(gridDim.x - bS - bM) * 4, wfrag, bfrag, F, N);
}

void launch_trunk(const Workspace& ws, const DeviceWeights& w, int n, bool resi, hipStream_t s) {
    // tasks: S n, M n/4, L n/16 -- all 240 MFMAs each.  Persistent-ish grid: ~2 blocks/CU.
    const int groups = (n + 15) / 16, tS = groups * 16, tM = groups * 4, tL = groups;
    auto blocks = [](int tasks, int budget) { int b = (tasks + 3) / 4; return b < budget ? b : budget; };
    const int bS = blocks(tS, 390), bM = blocks(tM, 98), bL = blocks(tL, 24);  // 512 blocks = 2 per CU (219 VGPRs -> 2 waves per SIMD)
    if (resi)
        hipLaunchKernelGGL(k1_trunk<true>, dim3(bS + bM + bL), dim3(256), 0, s, ws.xs, ws.xm, ws.xl, n, bS, bM,
                           w.trunk_w, w.trunk_b, ws.feat);
    else
        hipLaunchKernelGGL(k1_trunk<false>, dim3(bS + bM + bL), dim3(256), 0, s, ws.xs, ws.xm, ws.xl, n, bS, bM,
                           w.trunk_w, w.trunk_b, ws.feat);
}

// (k4: the fused FC2 + FC3 + sigmoid heads kernel lives in ethcnn_heads.hip)

__device__ __forceinline__ long global_chunk(long gn, int nctu, int cpf) {
    const long f = gn / nctu;
    return f * cpf + (gn - f * nctu) / kSubBatch;
}

// =========================================================================== k5 ======
// net_CNN.py:175  y32 = y32_tmp if any(y64 > thr1 over the fed sub-batch) else zeros
// net_CNN.py:187  y16 = y16_tmp if any(y32 > thr2) else zeros      (uses the GATED y32)
__global__ __launch_bounds__(256) void k5_gate(float* __restrict__ probs, const int* __restrict__ flags, int N,
                                               int nctu, int cpf, long ctu0, float thr2) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int i = idx / kNOut, j = idx % kNOut;
    if (i >= N || j == 0) return;
    const long chunk = global_chunk(ctu0 + i, nctu, cpf) - global_chunk(ctu0, nctu, cpf);
    const bool open32 = flags[2 * chunk] != 0;
    const bool open16 = open32 ? (flags[2 * chunk + 1] != 0) : (0.0f > thr2);
    if (j < 5 ? !open32 : !open16) probs[(size_t)i * kNOut + j] = 0.0f;
}

void launch_gate(const Workspace& ws, int n, int nctu, long ctu0, float thr2, float* d_probs, hipStream_t s) {
    const long total = (long)n * kNOut;
    hipLaunchKernelGGL(k5_gate, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_probs, ws.flags, n, nctu,
                       chunks_per_frame(nctu), ctu0, thr2);
}

}  // namespace ethcnn
