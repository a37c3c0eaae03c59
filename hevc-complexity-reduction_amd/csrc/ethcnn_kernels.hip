// ethcnn_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ETH-CNN path.
//
//   k0_tile   luma frames -> zero-padded 64x64 CTUs (video_to_cu_depth.py:46-59,88-106) in
//             trunk-lane order + exact integer 2x2 / 4x4 pooled sums (aver_pool,
//             net_CNN.py:62-63).  HBM-bound, LDS-staged.
//   k1_trunk  block-mean removal (net_CNN.py:78-84) + the three non-overlapping convs
//             (:86-92,127-141) of all 21 units per CTU, written as `h_conv_flat` (:143-150).
//             v_mfma_f32_16x16x4_f32, "transposed" (rows = output channels, columns = 16
//             units) so each layer's accumulator registers ARE the next layer's B operand:
//             no LDS, no cross-lane traffic between layers.
//   k_dense   LDS-tiled fp32 MFMA GEMM with fused (qp row) + bias + leaky-ReLU epilogue:
//             FC1 [N,2688]x[2688,448] (:156,164,177) and the three FC2 layers (:159,167,180).
//   k4_head   FC3 + sigmoid (:161,169,182) and the per-sub-batch gate predicates.
//   k5_gate   tf.cond zero fill (:175,187).
//
// Arithmetic contract ("canonical order"): every dot product is a single
// k-ordered fmaf chain starting from 0 (what one fp32 MFMA accumulator computes), bias is
// added afterwards with one rounding, leaky-ReLU = max(0.2f*h, h), pooling and block means
// come from exact integer pixel sums.  oracle/ethcnn_oracle.c (mode 0) restates exactly
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

// =========================================================================== k0 ======
// One block = 4 consecutive CTUs (global raster index n0 = 4*blockIdx.x over the frame
// sequence).  Stage: 4 x (64 rows x 64 B) coalesced 16-B loads -> LDS (row pitch 17 dwords)
// -> three outputs laid out so that every k1 load is one fully coalesced dwordx4 per lane:
//   XS[n][j][lane]      uint4: dwords q1=0..3 = 4 pixels of row g of patch (q2=j, q1) of unit u
//                       (lane = u + 16 g).  patch row Y = 16uy + 8(q2>>1) + 4(q1>>1) + g,
//                       X = 16ux + 8(q2&1) + 4(q1&1) + 0..3.
//   XM[n/4][j][lane]    uint4: patch rows d = 2j, 2j+1 (d = 4 q2 + q1), each 4 x u16 sums of
//                       2x2 raw pixels; column = 4*(n%4) + unit(2x2).
//   XL[n/16][j][lane]   same with 4x4 sums; column = n % 16.
constexpr int kTilePitch = 17;

template <bool FAST>
__global__ __launch_bounds__(256) void k0_tile(const uint8_t* __restrict__ luma, int width, int height, long pitch,
                                               long frame_stride, int cw, int nctu, long ctu0, int n_total,
                                               uint4* __restrict__ XS, uint4* __restrict__ XM,
                                               uint4* __restrict__ XL) {
    __shared__ uint32_t tile[4][64][kTilePitch];
    const int t = threadIdx.x;
    const int n0 = blockIdx.x * 4;

    // ---- load 4 CTUs (zero outside the frame / beyond n_total)
    {
        const int row = t >> 2, seg = t & 3;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = n0 + c;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (n < n_total) {
                const long gn = ctu0 + n;
                const long f = gn / nctu;
                const int rr = (int)(gn - f * nctu);
                const int cy = rr / cw, cx = rr - cy * cw;
                const int y = cy * 64 + row, x = cx * 64 + seg * 16;
                if (y < height && x < width) {
                    const uint8_t* p = luma + f * frame_stride + (long)y * pitch + x;
                    if (FAST) {
                        v = *reinterpret_cast<const uint4*>(p);
                    } else {
                        uint32_t w4[4] = {0u, 0u, 0u, 0u};
                        const int lim = min(16, width - x);
                        for (int i = 0; i < lim; ++i) w4[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3));
                        v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                    }
                }
            }
            uint32_t* dst = &tile[c][row][seg * 4];
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
    }
    __syncthreads();

    // ---- XS: 4 CTUs x 256 uint4, thread t -> (j, lane)
    {
        const int j = t >> 6, lane = t & 63, u = lane & 15, g = lane >> 4;
        const int uy = u >> 2, ux = u & 3;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (n0 + c < n_total) {
                uint32_t d[4];
#pragma unroll
                for (int q1 = 0; q1 < 4; ++q1) {
                    const int Y = 16 * uy + 8 * (j >> 1) + 4 * (q1 >> 1) + g;
                    const int Xd = 4 * ux + 2 * (j & 1) + (q1 & 1);
                    d[q1] = tile[c][Y][Xd];
                }
                XS[(size_t)(n0 + c) * 256 + t] = make_uint4(d[0], d[1], d[2], d[3]);
            }
        }
    }
    // ---- XM: one task record (4 CTUs), 512 uint4 -> 2 per thread
    {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int e = t + 256 * rep;
            const int j = e >> 6, lane = e & 63, col = lane & 15, g = lane >> 4;
            const int c = col >> 2, unit = col & 3, uy = unit >> 1, ux = unit & 1;
            uint32_t out[4];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int d = 2 * j + hh, q2 = d >> 2, q1 = d & 3;
                const int Yp = 16 * uy + 8 * (q2 >> 1) + 4 * (q1 >> 1) + g;  // pooled row (0..31)
                const int Xp = 16 * ux + 8 * (q2 & 1) + 4 * (q1 & 1);        // pooled col of px 0
                const uint32_t a0 = tile[c][2 * Yp][Xp >> 1], a1 = tile[c][2 * Yp][(Xp >> 1) + 1];
                const uint32_t b0 = tile[c][2 * Yp + 1][Xp >> 1], b1 = tile[c][2 * Yp + 1][(Xp >> 1) + 1];
                // pooled px i uses bytes 2i, 2i+1 of the 8-byte row pair
                const uint32_t s0 = (a0 & 0xff) + ((a0 >> 8) & 0xff) + (b0 & 0xff) + ((b0 >> 8) & 0xff);
                const uint32_t s1 = ((a0 >> 16) & 0xff) + (a0 >> 24) + ((b0 >> 16) & 0xff) + (b0 >> 24);
                const uint32_t s2 = (a1 & 0xff) + ((a1 >> 8) & 0xff) + (b1 & 0xff) + ((b1 >> 8) & 0xff);
                const uint32_t s3 = ((a1 >> 16) & 0xff) + (a1 >> 24) + ((b1 >> 16) & 0xff) + (b1 >> 24);
                out[2 * hh] = s0 | (s1 << 16);
                out[2 * hh + 1] = s2 | (s3 << 16);
            }
            XM[(size_t)(n0 >> 2) * 512 + e] = make_uint4(out[0], out[1], out[2], out[3]);
        }
    }
    // ---- XL: this block's 4 columns of the 16-CTU task record: 8 j x 4 c x 4 g = 128 uint4
    if (t < 128) {
        const int j = t >> 4, c = (t >> 2) & 3, g = t & 3;
        uint32_t out[4];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int d = 2 * j + hh, q2 = d >> 2, q1 = d & 3;
            const int Yp = 8 * (q2 >> 1) + 4 * (q1 >> 1) + g;  // pooled row (0..15)
            const int Xp = 8 * (q2 & 1) + 4 * (q1 & 1);        // pooled col == dword col
            uint32_t s[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t acc = 0;
#pragma unroll
                for (int ry = 0; ry < 4; ++ry) {
                    const uint32_t w = tile[c][4 * Yp + ry][Xp + i];
                    acc += (w & 0xff) + ((w >> 8) & 0xff) + ((w >> 16) & 0xff) + (w >> 24);
                }
                s[i] = acc;
            }
            out[2 * hh] = s[0] | (s[1] << 16);
            out[2 * hh + 1] = s[2] | (s[3] << 16);
        }
        const int lane = ((n0 & 15) + c) + 16 * g;
        XL[((size_t)(n0 >> 4) * 8 + j) * 64 + lane] = make_uint4(out[0], out[1], out[2], out[3]);
    }
}

void launch_tile(const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, const Workspace& ws, hipStream_t s) {
    const int blocks = (n + 3) / 4;
    const bool fast = (g.width % 16 == 0) && (g.pitch % 16 == 0) && (g.frame_stride % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(d_luma) % 16 == 0);
    if (fast)
        hipLaunchKernelGGL(k0_tile<true>, dim3(blocks), dim3(256), 0, s, d_luma, g.width, g.height, g.pitch,
                           g.frame_stride, g.cw, g.nctu, ctu0, n, ws.xs, ws.xm, ws.xl);
    else
        hipLaunchKernelGGL(k0_tile<false>, dim3(blocks), dim3(256), 0, s, d_luma, g.width, g.height, g.pitch,
                           g.frame_stride, g.cw, g.nctu, ctu0, n, ws.xs, ws.xm, ws.xl);
}

// =========================================================================== k1 ======
// One wave = one task = 16 units of one branch (S: the 16 units of one CTU; M: 4 CTUs x 4
// units; L: 16 CTUs).  lane = col + 16 g: col = unit (MFMA column), g = MFMA k-group.
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

    for (int task = wave; task < ntasks; task += nwaves) {
        // ---- pixels: x[d][kx], d = 4 q2 + q1 (patch), this lane's row g of each patch
        float x[16][4];
        int T = 0;
        if (BR == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4 d = X[((size_t)task * 4 + j) * 64 + lane];
                const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int q1 = 0; q1 < 4; ++q1)
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const int s = (int)((w[q1] >> (8 * kx)) & 0xff);
                        T += s;
                        x[4 * j + q1][kx] = px_value<RESI>(s, 1);
                    }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint4 d = X[((size_t)task * 8 + j) * 64 + lane];
                const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const int s = (int)((w[2 * hh + (kx >> 1)] >> (16 * (kx & 1))) & 0xffff);
                        T += s;
                        x[2 * j + hh][kx] = px_value<RESI>(s, POOL * POOL) * SCALE;
                    }
            }
        }
        T += __shfl_xor(T, 16);
        T += __shfl_xor(T, 32);
        const float mean = px_value<RESI>(T, 256 * POOL * POOL) * (SCALE * (1.0f / 256.0f));

        // ---- where this column's outputs go
        int n, by, bx;
        if (BR == 0) { n = task; by = col >> 2; bx = col & 3; }
        else if (BR == 1) { n = task * 4 + (col >> 2); by = (col >> 1) & 1; bx = col & 1; }
        else { n = task * 16 + col; by = 0; bx = 0; }
        const bool valid = n < N;
        float* Fn = F + (size_t)n * kNFeat;

        f32x4 a2[4][2];
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
            // conv1: 4 patches (q1) x 4 k-steps (s = kx); lane supplies v[patch][ky=g][kx=s]
            f32x4 c1[4];
#pragma unroll
            for (int q1 = 0; q1 < 4; ++q1) c1[q1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q1 = 0; q1 < 4; ++q1) c1[q1] = MFMA16(A1[s], x[4 * q2 + q1][s] - mean, c1[q1]);
#pragma unroll
            for (int q1 = 0; q1 < 4; ++q1)
#pragma unroll
                for (int r = 0; r < 4; ++r) c1[q1][r] = lrelu(c1[q1][r] + B1[r]);
            // conv2: K = (q1, r, g) with ci = 4g + r; two M tiles (channels 0-15, 16-23 + pad)
            f32x4 c2[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
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
                for (int r = 0; r < 4; ++r) c2[t][r] = lrelu(c2[t][r] + B2[t][r]);
            a2[q2][0] = c2[0];
            a2[q2][1] = c2[1];
            if (valid) {
                const int slot = (2 * by + (q2 >> 1)) * (2 * NB) + 2 * bx + (q2 & 1);
                float* dst = Fn + OFF2 + slot * 24;
                *reinterpret_cast<f32x4*>(dst + 4 * g) = c2[0];
                if (g < 2) *reinterpret_cast<f32x4*>(dst + 16 + 4 * g) = c2[1];
            }
        }
        // conv3: phase A = channels 0..15 of the 4 positions, phase B = channels 16..23 with
        // positions (2j, 2j+1) packed into the lower / upper lane halves.
        f32x4 c3[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
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
            float* dst = Fn + OFF3 + (by * NB + bx) * 32;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = lrelu(c3[t][r] + B3[t][r]);
                *reinterpret_cast<f32x4*>(dst + 16 * t + 4 * g) = o;
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
    if (b < bS) trunk_tasks<0, RESI>(XS, N, b * 4 + w, bS * 4, wfrag, bfrag, F, N);
    else if (b < bS + bM) trunk_tasks<1, RESI>(XM, (N + 3) / 4, (b - bS) * 4 + w, bM * 4, wfrag, bfrag, F, N);
    else trunk_tasks<2, RESI>(XL, (N + 15) / 16, (b - bS - bM) * 4 + w, (int)(gridDim.x - bS - bM) * 4, wfrag, bfrag, F, N);
}

void launch_trunk(const Workspace& ws, const DeviceWeights& w, int n, bool resi, hipStream_t s) {
    // tasks: S n, M n/4, L n/16 -- all 240 MFMAs each.  Persistent-ish grid: ~2 blocks/CU.
    const int tS = n, tM = (n + 3) / 4, tL = (n + 15) / 16;
    auto blocks = [](int tasks, int budget) { int b = (tasks + 3) / 4; return b < budget ? b : budget; };
    const int bS = blocks(tS, 768), bM = blocks(tM, 192), bL = blocks(tL, 48);
    if (resi)
        hipLaunchKernelGGL(k1_trunk<true>, dim3(bS + bM + bL), dim3(256), 0, s, ws.xs, ws.xm, ws.xl, n, bS, bM,
                           w.trunk_w, w.trunk_b, ws.feat);
    else
        hipLaunchKernelGGL(k1_trunk<false>, dim3(bS + bM + bL), dim3(256), 0, s, ws.xs, ws.xm, ws.xl, n, bS, bM,
                           w.trunk_w, w.trunk_b, ws.feat);
}

// ====================================================================== k_dense ======
// out[m][n] = lrelu( sum_k A[m][k] W[k][n]  (+ qn * W[K][n])  + bias[n] )
// Block = WM x WN waves; wave tile = (16 MS) x (16 NS); block tile BM = 16 MS WM rows,
// BN = 16 NS WN columns.  NSPLIT blocks share one M tile, each owning BN of the layer's
// NSPLIT*BN columns; the column block is blockIdx.x % NSPLIT, so (dispatcher places block b
// on XCD b % 8) every XCD only ever touches 1/NSPLIT of W and that slice stays resident in
// its private 4 MiB L2 (the full FC1 matrix, 4.8 MB, does not fit).  Speed only: results do
// not depend on placement.  K is consumed in BK-wide chunks staged through LDS
// (register-staged double buffer, one barrier per chunk); each accumulator is ONE
// ascending-k MFMA chain (no split-K), which is the canonical order.
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int MS, int NS, int WM, int WN, int BK, int NSPLIT, bool QP>
__global__ __launch_bounds__(64 * WM * WN) void k_dense(const float* __restrict__ A, int lda, int K,
                                                        const float* __restrict__ W, const float* __restrict__ bias,
                                                        float qn, float* __restrict__ out, int ldo, int M) {
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int BM = 16 * MS * WM, BN = 16 * NS * WN;
    constexpr int LDW = BN * NSPLIT;  // row stride of W == number of columns of the layer
    constexpr int AP = BK + 1;        // A tile row pitch (floats): conflict-free column reads
    // B tile: LINEAR [BK][BN] image (what global_load_lds writes: wave-uniform base + 16 B x
    // lane).  When BN % 32 == 0 the two k-groups of a 32-lane half (g = 0,1) would hit the
    // same banks, so odd k rows are stored with adjacent 16-column groups swapped: the
    // permutation is applied to the per-lane GLOBAL address and undone on the ds_read.
    constexpr bool SWZ = (BN % 32 == 0);
    constexpr int A_F4 = BM * BK / 4, A_PER = (A_F4 + NT - 1) / NT;
    constexpr int B_INST = BK * BN / 256;               // 1 KiB wave-instructions per B tile
    constexpr int B_PER = (B_INST + NW - 1) / NW;       // per wave
    constexpr int A_FLOATS = BM * AP, B_FLOATS = BK * BN;
    constexpr int BUF = (A_FLOATS + B_FLOATS + 3) / 4 * 4;
    // one LDS object only (a second __shared__ array makes hipcc drain vmcnt before every ds_read)
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int col = lane & 15, g = lane >> 4;
    const int nb = (NSPLIT > 1) ? (int)(blockIdx.x % NSPLIT) : 0;
    const int m0 = (int)(blockIdx.x / NSPLIT) * BM;
    const int n0 = nb * BN;

    f32x4 acc[MS][NS];
#pragma unroll
    for (int i = 0; i < MS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // A tile: register staged (rows beyond M clamp to M-1: loaded, never stored)
    float4 ra[A_PER];
    const float* a_src[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int e = tid + i * NT;
        const int row = (e / (BK / 4)) % BM, k4 = e % (BK / 4);
        a_src[i] = A + (size_t)min(m0 + row, M - 1) * lda + k4 * 4;
    }
    // B tile: wave wv issues instructions wv, wv + NW, ...; instruction q covers float4
    // indices [64 q, 64 q + 63] of the linear tile
    const float* b_src[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int e = (wv + i * NW) * 64 + lane;  // float4 index in the tile
        const int row = (e / (BN / 4)) % BK;
        int c4 = e % (BN / 4);
        if (SWZ) c4 ^= (row & 1) << 2;
        b_src[i] = W + (size_t)row * LDW + n0 + c4 * 4;
    }
#define DENSE_GLOAD(kc, buf)                                                                          \
    {                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < A_PER; ++i)                                             \
            ra[i] = *reinterpret_cast<const float4*>(a_src[i] + (size_t)(kc) * BK);                   \
        _Pragma("unroll") for (int i = 0; i < B_PER; ++i) {                                           \
            if ((i + 1) * NW <= B_INST || wv + i * NW < B_INST)                                       \
                __builtin_amdgcn_global_load_lds((glb_void*)(b_src[i] + (size_t)(kc) * BK * LDW),     \
                                                 (lds_void*)(smem + (buf) * BUF + A_FLOATS + (wv + i * NW) * 256), \
                                                 16, 0, 0);                                           \
        }                                                                                             \
    }
#define DENSE_ASTORE(buf)                                                                             \
    {                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < A_PER; ++i) {                                           \
            const int e = tid + i * NT;                                                               \
            if ((i + 1) * NT <= A_F4 || e < A_F4) {                                                   \
                const int row = e / (BK / 4), k4 = e % (BK / 4);                                      \
                float* d = smem + (buf) * BUF + row * AP + k4 * 4;                                    \
                d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;                       \
            }                                                                                         \
        }                                                                                             \
    }

    const int nk = K / BK;
    DENSE_GLOAD(0, 0);
    DENSE_ASTORE(0);
    __syncthreads();  // (hipcc drains vmcnt here: the LDS-DMA of chunk 0 has landed)
    const int b_col = wn * 16 * NS + col;
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) DENSE_GLOAD(kc + 1, buf ^ 1);  // in flight during the MFMAs below
        const float* as = smem + buf * BUF + (wm * 16 * MS + col) * AP + g;
        const float* bs = smem + buf * BUF + A_FLOATS + g * BN;
#pragma unroll
        for (int kq = 0; kq < BK / 4; ++kq) {
            float a[MS], b[NS];
#pragma unroll
            for (int i = 0; i < MS; ++i) a[i] = as[i * 16 * AP + kq * 4];
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                int c = b_col + j * 16;
                if (SWZ) c ^= (g & 1) << 4;  // row k = 4 kq + g is odd iff g is odd
                b[j] = bs[kq * 4 * BN + c];
            }
#pragma unroll
            for (int i = 0; i < MS; ++i)
#pragma unroll
                for (int j = 0; j < NS; ++j) acc[i][j] = MFMA16(a[i], b[j], acc[i][j]);
        }
        if (kc + 1 < nk) DENSE_ASTORE(buf ^ 1);
        __syncthreads();
    }
#undef DENSE_GLOAD
#undef DENSE_ASTORE

    // epilogue: C layout row = 4g + r, col = lane & 15
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int n = n0 + wn * 16 * NS + j * 16 + col;
        const float bv = bias[n];
        float wq = 0.f;
        if (QP) wq = W[(size_t)K * LDW + n];
#pragma unroll
        for (int i = 0; i < MS; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 16 * MS + i * 16 + 4 * g + r;
                float v = acc[i][j][r];
                if (QP) v = fmaf(qn, wq, v);
                v = lrelu(v + bv);
                if (m < M) out[(size_t)m * ldo + n] = v;
            }
    }
}

template <int MS, int NS, int WM, int WN, int BK, int NSPLIT, bool QP>
static void launch_dense(const float* A, int lda, int K, const float* W, const float* bias, float qn, float* out,
                         int ldo, int M, hipStream_t s) {
    constexpr int BM = 16 * MS * WM;
    hipLaunchKernelGGL((k_dense<MS, NS, WM, WN, BK, NSPLIT, QP>), dim3(((M + BM - 1) / BM) * NSPLIT),
                       dim3(64 * WM * WN), 0, s, A, lda, K, W, bias, qn, out, ldo, M);
}

static int fc1_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ETHCNN_FC1_VARIANT");  // development knob: tile-shape A/B runs
        v = e ? atoi(e) : 6;  // measured best on MI355X (profiles/r01_fc1_variants.txt)
    }
    return v;
}

void launch_fc1(const Workspace& ws, const DeviceWeights& w, int n, float* out, hipStream_t s) {
    const float* A = ws.feat;
    switch (fc1_variant()) {
        case 0:  // BM 64 x BN 448, BK 16, 4 waves
            launch_dense<4, 7, 1, 4, 16, 1, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 1:  // BM 128 x BN 224 (N split 2), 4 waves
            launch_dense<4, 7, 2, 2, 16, 2, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 2:  // BM 256 x BN 224 (N split 2), 8 waves
            launch_dense<4, 7, 4, 2, 16, 2, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 3:  // BM 256 x BN 112 (N split 4), 4 waves
            launch_dense<4, 7, 4, 1, 16, 4, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 4:  // BM 128 x BN 448, 8 waves
            launch_dense<4, 7, 2, 4, 16, 1, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 5:  // BM 128 x BN 224, BK 32
            launch_dense<4, 7, 2, 2, 32, 2, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        default:  // 6: BM 128 x BN 112 (N split 4), wave 32 x 112, 4 waves, ~4 blocks / CU
            launch_dense<2, 7, 4, 1, 16, 4, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
    }
}

void launch_fc2(const Workspace& ws, const DeviceWeights& w, int n, float qn, hipStream_t s) {
    launch_dense<1, 3, 4, 1, 16, 1, true>(ws.h1 + kO1[0], kNVec, kN1[0], w.fc2_w[0], w.fc2_b[0], qn, ws.h2 + kO2[0], kNFc2, n, s);
    launch_dense<2, 3, 2, 2, 16, 1, true>(ws.h1 + kO1[1], kNVec, kN1[1], w.fc2_w[1], w.fc2_b[1], qn, ws.h2 + kO2[1], kNFc2, n, s);
    launch_dense<4, 3, 1, 4, 16, 1, true>(ws.h1 + kO1[2], kNVec, kN1[2], w.fc2_w[2], w.fc2_b[2], qn, ws.h2 + kO2[2], kNFc2, n, s);
}

// =========================================================================== k4 ======
// FC3 + sigmoid + gate predicates.
__device__ __forceinline__ float expf_canonical(float x) {
    x = fminf(x, 80.0f);
    x = fmaxf(x, -86.0f);
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    return __int_as_float(__float_as_int(p) + (((int)n) << 23));
}

struct HeadParams {
    const float* w3[3];
    const float* b3[3];
};

__device__ __forceinline__ long global_chunk(long gn, int nctu, int cpf) {
    const long f = gn / nctu;
    return f * cpf + (gn - f * nctu) / kSubBatch;
}

// One wave = 16 CTUs.  Per head: D[ctu][j] = sum_k H2[ctu][k] W3[k][j] as one 16x16 MFMA
// tile (n3 = 1 / 4 / 16 real columns, the rest zero), K = 48 / 96 / 192 ascending -- the
// canonical chain -- then the qp column, bias and sigmoid in the epilogue.  The three
// heads' chains are independent and interleaved.
__global__ __launch_bounds__(256) void k4_head(const float* __restrict__ H2, HeadParams hp, float qn, int N,
                                               int nctu, int cpf, long ctu0, float thr1, float thr2,
                                               float* __restrict__ logits, float* __restrict__ raw,
                                               float* __restrict__ probs, int* __restrict__ flags) {
    const int lane = threadIdx.x & 63, col = lane & 15, g = lane >> 4;
    const int group = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int base = group * 16;
    if (base >= N) return;
    const int arow = min(base + col, N - 1);  // A operand row (clamped: loaded, never stored)
    const float* x = H2 + (size_t)arow * kNFc2 + g;
    f32x4 acc[3] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    constexpr int N2[3] = {48, 96, 192}, N3[3] = {1, 4, 16}, O2[3] = {0, 48, 144}, O3[3] = {0, 1, 5};
#pragma unroll 4
    for (int s = 0; s < 48; ++s) {  // 48 k-steps cover head16; head32 uses the first 24, head64 the first 12
        const int k = 4 * s + g;
        {
            const float b = hp.w3[2][k * 16 + col];
            acc[2] = MFMA16(x[O2[2] + 4 * s], b, acc[2]);
        }
        if (s < 24) {
            const float b = (col < 4) ? hp.w3[1][k * 4 + col] : 0.0f;
            acc[1] = MFMA16(x[O2[1] + 4 * s], b, acc[1]);
        }
        if (s < 12) {
            const float b = (col < 1) ? hp.w3[0][k] : 0.0f;
            acc[0] = MFMA16(x[O2[0] + 4 * s], b, acc[0]);
        }
    }
    const long chunk0 = global_chunk(ctu0, nctu, cpf);
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        if (col < N3[h]) {
            const float wq = hp.w3[h][N2[h] * N3[h] + col], bv = hp.b3[h][col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = base + 4 * g + r;
                if (i < N) {
                    const float z = fmaf(qn, wq, acc[h][r]) + bv;
                    const float p = 1.0f / (1.0f + expf_canonical(-z));
                    const size_t o = (size_t)i * kNOut + O3[h] + col;
                    logits[o] = z;
                    raw[o] = p;
                    probs[o] = p;
                    // any(y64 > THR_L1_LOWER) / any(y32_tmp > THR_L2_LOWER) over the sub-batch
                    const bool hit = (h == 0 && p > thr1) || (h == 1 && p > thr2);
                    if (hit) {
                        int* f = flags + 2 * (global_chunk(ctu0 + i, nctu, cpf) - chunk0) + h;
                        if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                            __hip_atomic_store(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
    }
}

void launch_head(const Workspace& ws, const DeviceWeights& w, int n, float qn, int nctu, long ctu0, float thr1,
                 float thr2, float* d_probs, hipStream_t s) {
    HeadParams hp;
    for (int h = 0; h < 3; ++h) {
        hp.w3[h] = w.fc3_w[h];
        hp.b3[h] = w.fc3_b[h];
    }
    hipLaunchKernelGGL(k4_head, dim3((n + 63) / 64), dim3(256), 0, s, ws.h2, hp, qn, n, nctu,
                       chunks_per_frame(nctu), ctu0, thr1, thr2, ws.logits, ws.raw, d_probs, ws.flags);
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
