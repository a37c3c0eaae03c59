// ethcnn_trunk.hip -- k1: 16x16 block-mean removal (zero_mean_norm_local, net_CNN.py:78-84) and the
// three non-overlapping convs (non_overlap_conv, :86-92,127-141) of the 21 units of every CTU,
// written as `h_conv_flat` (:143-150).  gfx950, v_mfma_f32_16x16x4_f32.
//
// One wave = one task = the SAME unit position of 16 consecutive CTUs (a group): 16 S tasks, 4 M
// tasks and 1 L task per group, 240 MFMAs each.  lane = col + 16 g: col = CTU within the group
// (MFMA column), g = MFMA k-group.  The trunk runs "transposed" (rows = output channels): MFMA
// D[row][col] lives in lane (col, g) as rows 4g..4g+3, which is exactly B[k = g][col] for the 4
// k-steps r = 0..3 of the next layer when that layer's K is enumerated as (patch, r, g) with
// ci = 4g + r.  So each layer's accumulator registers ARE the next layer's B operand: no LDS, no
// cross-lane traffic (one lane-half swap for conv3's channels 16..23).  conv1's A operands and the
// biases stay in VGPRs, the 80 conv2 / conv3 A fragments of the branch in 21 KB of LDS (one
// conflict-free ds_read_b32 per MFMA costs the matrix pipe ~0.6 clocks).
//
// Three waves per SIMD (156 VGPRs) overlap one wave's VALU epilogues with the others' MFMAs; the
// next task's pixel record is prefetched a task ahead.  (A deeper in-wave software pipeline -- conv1(q+1) before leaky(conv1(q)) --
// was measured: it needs > 256 VGPRs, drops to one wave per SIMD and is 20 % slower.)  The task loop sits at 98 % of its
// instruction-mix bound: 240 MFMAs + 375 VALU (each already the cheapest instruction that computes it exactly) + 98 LDS
// + 14 VMEM = ~9960 predicted against 10,165 measured clocks per task.
// Features are written as feat[group][k/4][16][4]: every store is a 256-byte run per k-group.
//
// Arithmetic contract (DESIGN.md "canonical order"; oracle/ethcnn_oracle.c mode 0 restates it):
// each accumulator is one fmaf chain in MFMA k order starting from the bias; leaky = max(0.2h, h);
// v = fma(float(pixel sum), c255 * 2^-p, -mean) with exact integer pixel sums.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ethcnn_kernels.h"

namespace ethcnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifndef TRUNK_BUF_LOADS
#define TRUNK_BUF_LOADS 1
#endif
#ifndef TRUNK_BUF_STORES
#define TRUNK_BUF_STORES 1
#endif
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// max(0.2h, h) as v_mul + ONE v_max per value: fmaxf() makes hipcc canonicalise the raw MFMA
// output first (a second v_max per value; 104 values per task).  Hazards (guide 5.7): the
// compiler-generated multiplies read the MFMA result first, so the MFMA->VALU wait states are
// served before the asm issues; the asm's outputs feed MFMA operands, hence the trailing s_nop 1
// (VALU write -> MFMA read), paid once per four values.
__device__ __forceinline__ f32x4 lrelu4(f32x4 h) {
    const float t0 = 0.2f * h[0], t1 = 0.2f * h[1], t2 = 0.2f * h[2], t3 = 0.2f * h[3];
    float o0, o1, o2, o3;
    asm volatile("v_max_f32 %0, %4, %8\n\tv_max_f32 %1, %5, %9\n\tv_max_f32 %2, %6, %10\n\tv_max_f32 %3, %7, %11\n\ts_nop 1"
                 : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
                 : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]));
    return (f32x4){o0, o1, o2, o3};
}

template <bool RESI>
__device__ __forceinline__ float px_value(int s, int cnt) {
    if (RESI) return ((float)(s - 128 * cnt) / 255.0f) * 10.0f;  // (x-128)/255.0*10, LSTM net :153
    return (float)s * (1.0f / 255.0f);                           // x * 1/255, net_CNN.py:105
}

template <int BR, bool RESI>
struct Trunk {
    static constexpr int POOL = (BR == 0) ? 1 : (BR == 1 ? 2 : 4);
    static constexpr float SCALE = 1.0f / (float)(POOL * POOL);
    static constexpr float C255S = (1.0f / 255.0f) * SCALE;  // exact: SCALE is a power of two
    static constexpr int NB = (BR == 0) ? 4 : (BR == 1 ? 2 : 1);
    static constexpr int OFF2 = (BR == 0) ? 672 : (BR == 1 ? 2208 : 2592);
    static constexpr int OFF3 = (BR == 0) ? 0 : (BR == 1 ? 512 : 640);
    static constexpr int NJ = (BR == 0) ? 4 : 8;  // uint4 records per lane per task

    // exact integer sum of this lane's part of the record (the block mean needs it before conv1)
    static __device__ __forceinline__ int raw_sum(const uint4 (&raw)[NJ]) {
        int T = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const uint32_t w[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (BR == 0) T = (int)__builtin_amdgcn_udot4(w[i], 0x01010101u, (unsigned)T, false);  // exact byte sum
                else T += (int)((w[i] & 0xffffu) + (w[i] >> 16));
            }
        }
        return T;
    }
    // the 4 conv1 patches of position q2 -> x[q1][kx] (pixel sums as floats; resi: preprocessed values)
    static __device__ __forceinline__ void decode_q2(const uint4 (&raw)[NJ], int q2, float (&x)[4][4]) {
        if (BR == 0) {
            const uint32_t w[4] = {raw[q2].x, raw[q2].y, raw[q2].z, raw[q2].w};
#pragma unroll
            for (int q1 = 0; q1 < 4; ++q1)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int s = (int)((w[q1] >> (8 * kx)) & 0xff);
                    x[q1][kx] = RESI ? px_value<true>(s, 1) : (float)s;
                }
        } else {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const uint4 rw = raw[2 * q2 + jj];
                const uint32_t w[4] = {rw.x, rw.y, rw.z, rw.w};
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const int s = (int)((w[2 * hh + (kx >> 1)] >> (16 * (kx & 1))) & 0xffff);
                        x[2 * jj + hh][kx] = RESI ? px_value<true>(s, POOL * POOL) * SCALE : (float)s;
                    }
            }
        }
    }

    static __device__ __forceinline__ void run(const uint4* __restrict__ X, int ntasks, int wave, int nwaves,
                                               const float* __restrict__ wfrag, const float* __restrict__ bfrag,
                                               float* __restrict__ F, int N, float* wl) {
        const int lane = threadIdx.x & 63;
        const int col = lane & 15, g = lane >> 4;
        // conv2 / conv3 A-operand fragments of this branch -> LDS, once per block (80 of the 84
        // fragments; every MFMA fetches its A operand with one conflict-free ds_read_b32, which costs
        // the matrix pipe nothing, and frees 80 VGPRs: three waves per SIMD instead of two)
        {
            const float* wf = wfrag + (size_t)BR * kTrunkWFrags * 64;
            for (int i = threadIdx.x; i < kTrunkWFrags * 64; i += 256) wl[i] = wf[i];
        }
        __syncthreads();
        if (wave >= ntasks) return;
        const float* wA2 = wl + 4 * 64 + lane;   // A2[t][s] = wA2[(t * 16 + s) * 64]
        const float* wA3 = wl + 36 * 64 + lane;  // A3[t][s] = wA3[(t * 24 + s) * 64]

        float A1[4];
        f32x4 B1, B2[2], B3[2];
        {
#pragma unroll
            for (int s = 0; s < 4; ++s) A1[s] = wl[s * 64 + lane];
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

        // buffer addressing: SGPR resource + SGPR task / slot offset + ONE VGPR lane offset per instruction.
        // The 128-bit STORES keep their whole offset in the VGPR (one v_add each) and soffset = 0: with an SGPR soffset
        // hipcc (ROCm 7.2) places the next VALU write of the store-data registers directly behind the store -- its
        // hazard recognizer assumes a register soffset removes the ">64-bit store data" hazard -- and on gfx950 lanes
        // 12..15 of every row then stored the overwritten value (profiles/r02_fc1_variants.txt, "buffer_store hazard").
        const int lane16 = lane * 16;
        const int lane_off = (col * 4 + g * 64) * 4;  // feature stores: [k/4][16 CTUs][4] -> g, col
        const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(X), 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(F, 0, -1, 0x00020000);
        uint4 raw[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            raw[j] = TRUNK_BUF_LOADS ? __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rX, lane16, (wave * NJ + j) * 1024, 0))
                                     : X[((size_t)wave * NJ + j) * 64 + lane];

        for (int task = wave; task < ntasks; task += nwaves) {
            int T = raw_sum(raw);
            T += __shfl_xor(T, 16);
            T += __shfl_xor(T, 32);
            // canonical centring: AI  v = fma(float(sum), c255 * 2^-p, -mean)  (one rounding);
            //                     resi v = x - mean with x = ((s - 128 cnt) / 255 * 10) * 2^-p
            const float mean = px_value<RESI>(T, 256 * POOL * POOL) * (SCALE * (1.0f / 256.0f));
            const float negmean = -mean;

            int grp, by, bx;  // wave-uniform
            if (BR == 0) { grp = task >> 4; by = (task >> 2) & 3; bx = task & 3; }
            else if (BR == 1) { grp = task >> 2; by = (task >> 1) & 1; bx = task & 1; }
            else { grp = task; by = 0; bx = 0; }
            const bool valid = grp * 16 + col < N;
            // feature k of this lane's CTU: group image [(k/4)][16 CTUs][4]; k = k0 + 4 g with a uniform k0 % 4 == 0
            const int Fg = grp * (kNFeat * 16 * 4);  // uniform byte offset of the group image (< 2^31: <= 8192 groups)

            // conv1 of position q2: 4 patches (q1) x 4 k-steps (s = kx); lane supplies v[patch][ky=g][kx=s]
#define CONV1(q2, c1)                                                                                  \
    {                                                                                                  \
        _Pragma("unroll") for (int q1 = 0; q1 < 4; ++q1) c1[q1] = B1;                                  \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                  \
            _Pragma("unroll") for (int q1 = 0; q1 < 4; ++q1) {                                         \
                const float xv = x[q1][s];                                                  \
                c1[q1] = MFMA16(A1[s], RESI ? xv - mean : fmaf(xv, C255S, negmean), c1[q1]);           \
            }                                                                                          \
    }
            // conv2 of position q2: K = (q1, r, g) with ci = 4g + r; two M tiles (channels 0-15, 16-23 + pad)
#define CONV2(c1, c2)                                                                                  \
    {                                                                                                  \
        c2[0] = B2[0];                                                                                 \
        c2[1] = B2[1];                                                                                 \
        _Pragma("unroll") for (int q1 = 0; q1 < 4; ++q1)                                               \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                            \
                c2[0] = MFMA16(wA2[(4 * q1 + r) * 64], c1[q1][r], c2[0]);                                   \
                c2[1] = MFMA16(wA2[(16 + 4 * q1 + r) * 64], c1[q1][r], c2[1]);                                   \
            }                                                                                          \
    }
#define LEAKY1(c1) { _Pragma("unroll") for (int q1 = 0; q1 < 4; ++q1) c1[q1] = lrelu4(c1[q1]); }
            // leaky + store of conv2 position q2
#define FINISH2(q2, c2)                                                                                \
    {                                                                                                  \
        a2[q2][0] = lrelu4(c2[0]);                                                                     \
        a2[q2][1] = lrelu4(c2[1]);                                                                     \
        if (valid) {                                                                                   \
            const int slot = (2 * by + ((q2) >> 1)) * (2 * NB) + 2 * bx + ((q2) & 1);                  \
            const int dst = Fg + ((OFF2 + slot * 24) >> 2) * 256;                                      \
            if (TRUNK_BUF_STORES) {                                                                    \
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a2[q2][0]), rF, lane_off + dst, 0, 0);  \
                if (g < 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a2[q2][1]), rF, lane_off + dst + 1024, 0, 0); \
            } else {                                                                                   \
                char* p_ = reinterpret_cast<char*>(F) + (size_t)dst + lane_off;                        \
                *reinterpret_cast<f32x4*>(p_) = a2[q2][0];                                             \
                if (g < 2) *reinterpret_cast<f32x4*>(p_ + 1024) = a2[q2][1];                           \
            }                                                                                          \
        }                                                                                              \
    }
            f32x4 a2[4][2];
            f32x4 c3[2] = {B3[0], B3[1]};
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2) {
                f32x4 c1[4], c2[2];
                float x[4][4];
                decode_q2(raw, q2, x);
                CONV1(q2, c1);
                LEAKY1(c1);
                CONV2(c1, c2);
                FINISH2(q2, c2);
            }
            // the raw registers are dead now: the next task's record is fetched under conv3 (and the
            // other waves' work) at no VGPR cost
            if (task + nwaves < ntasks) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    raw[j] = TRUNK_BUF_LOADS ? __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rX, lane16, ((task + nwaves) * NJ + j) * 1024, 0))
                                             : X[((size_t)(task + nwaves) * NJ + j) * 64 + lane];
            }
#undef CONV1
#undef CONV2
#undef LEAKY1
#undef FINISH2
            // conv3 phase A: channels 0..15 of the 4 positions
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    c3[0] = MFMA16(wA3[(4 * q2 + r) * 64], a2[q2][0][r], c3[0]);
                    c3[1] = MFMA16(wA3[(24 + 4 * q2 + r) * 64], a2[q2][0][r], c3[1]);
                }
            // phase B: channels 16..23, positions (2j, 2j+1) packed into the lower / upper lane halves
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float hi = __shfl(a2[2 * j + 1][1][r], lane & 31);  // lanes 32..63 <- lanes 0..31 of position 2j+1
                    const float z = (lane < 32) ? a2[2 * j][1][r] : hi;
                    c3[0] = MFMA16(wA3[(16 + 4 * j + r) * 64], z, c3[0]);
                    c3[1] = MFMA16(wA3[(24 + 16 + 4 * j + r) * 64], z, c3[1]);
                }
            if (valid) {
                const int dst = Fg + ((OFF3 + (by * NB + bx) * 32) >> 2) * 256;
                if (TRUNK_BUF_STORES) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lrelu4(c3[0])), rF, lane_off + dst, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lrelu4(c3[1])), rF, lane_off + dst + 1024, 0, 0);
                } else {
                    char* p_ = reinterpret_cast<char*>(F) + (size_t)dst + lane_off;
                    *reinterpret_cast<f32x4*>(p_) = lrelu4(c3[0]);
                    *reinterpret_cast<f32x4*>(p_ + 1024) = lrelu4(c3[1]);
                }
            }
        }
    }
};

template <bool RESI>
__global__ __launch_bounds__(256) void k1_trunk(const uint4* __restrict__ XS, const uint4* __restrict__ XM,
                                                const uint4* __restrict__ XL, int N, int bS, int bM,
                                                const float* __restrict__ wfrag, const float* __restrict__ bfrag,
                                                float* __restrict__ F) {
    __shared__ float wl[kTrunkWFrags * 64];  // this block's branch: 84 A-operand fragments, 21 KB
    // wave-uniform on purpose (readfirstlane): task, group and unit indices then live in SGPRs, and every load / store
    // below is "SGPR base + one 32-bit VGPR lane offset" (saddr form) instead of a 64-bit per-lane address: a VMEM
    // instruction with VGPR addresses costs several times more matrix-pipe time (profiles/r02_fc1_variants.txt)
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const int groups = (N + 15) / 16;
    if (b < bS) Trunk<0, RESI>::run(XS, groups * 16, b * 4 + w, bS * 4, wfrag, bfrag, F, N, wl);
    else if (b < bS + bM) Trunk<1, RESI>::run(XM, groups * 4, (b - bS) * 4 + w, bM * 4, wfrag, bfrag, F, N, wl);
    else Trunk<2, RESI>::run(XL, groups, (b - bS - bM) * 4 + w, (int)(gridDim.x - bS - bM) * 4, wfrag, bfrag, F, N, wl);
}

// blocks per 256 of the S / M / L branches: tasks 16 : 4 : 1, weighted by their instruction-mix cost per task (M / L tasks
// carry ~80 more VALU for the 16-bit unpack: ~9700 vs ~9400 clocks) -> 193 : 50 : 13 (195 : 49 : 12 measured 0.6 % slower)
#ifndef TRUNK_SH_S
#define TRUNK_SH_S 193
#define TRUNK_SH_M 50
#define TRUNK_SH_L 13
#endif

void launch_trunk(const Workspace& ws, const DeviceWeights& w, int n, bool resi, hipStream_t s) {
    // tasks per group: 16 S, 4 M, 1 L -- all 240 MFMAs.  768 blocks = 3 per CU (156 VGPRs, 21 KB LDS).
    const int groups = (n + 15) / 16, tS = groups * 16, tM = groups * 4, tL = groups;
    auto blocks = [](int tasks, int budget) { int b = (tasks + 3) / 4; return b < budget ? b : budget; };
    static int per_cu = 0;
    if (!per_cu) { const char* e = getenv("ETHCNN_TRUNK_BLOCKS_PER_CU"); per_cu = e ? atoi(e) : 3; }  // development knob
    const int bS = blocks(tS, TRUNK_SH_S * per_cu), bM = blocks(tM, TRUNK_SH_M * per_cu), bL = blocks(tL, TRUNK_SH_L * per_cu);
    if (resi)
        hipLaunchKernelGGL(k1_trunk<true>, dim3(bS + bM + bL), dim3(256), 0, s, ws.xs, ws.xm, ws.xl, n, bS, bM,
                           w.trunk_w, w.trunk_b, ws.feat);
    else
        hipLaunchKernelGGL(k1_trunk<false>, dim3(bS + bM + bL), dim3(256), 0, s, ws.xs, ws.xm, ws.xl, n, bS, bM,
                           w.trunk_w, w.trunk_b, ws.feat);
}

}  // namespace ethcnn
