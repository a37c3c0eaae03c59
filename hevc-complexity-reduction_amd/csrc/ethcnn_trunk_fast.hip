// ethcnn_trunk_fast.hip -- plan 3 (opt-in, ethcnn_set_fc1_plan(ctx, 3)): the trunk of plan 2 -- 16x16 block-mean removal and the
// three non-overlapping convs of the 21 units of every CTU (net_CNN.py:78-92,126-150) -- with its convolutions on the 16-BIT
// matrix pipe as well, operands as fp16 x 2 splits like FC1's (ethcnn_fc1_fast.hip), fp32 accumulation.  Under plan 2 the exact-fp32
// trunk is the largest stage of a step (240 v_mfma_f32_16x16x4_f32 = 7680 matrix-pipe cycles per task at 1/16 of the 16-bit rate).
//
// Same task structure as ethcnn_trunk_task.h: one wave = one unit position of 16 CTUs (a group), lane = col + 16 g, everything
// "transposed" (MFMA rows = output channels, columns = CTUs) so that a layer's accumulator registers are the next layer's B operand
// without any cross-lane traffic -- which survives the move to the K = 32 shape: the B operand of v_mfma_f32_16x16x32_f16 is eight
// k values per lane, and lane (col, g) holds, as TWO accumulator quads of the previous layer, exactly channels 4 g .. 4 g + 3 of two
// positions.
//   conv1 (K = 16 taps): the pixel SUMS themselves are the B operand -- small integers, exact in fp16 (S: bytes, M: 2x2 sums
//          <= 1020; L: 4x4 sums <= 4080 as two digits 16 hi + lo) -- of v_mfma_f32_16x16x16_f16; weights as two fp16 pieces: every
//          product is exact, and  sum w (c s - mean) + b  =  c sum(w s) - mean sum(w) + b  is applied to the accumulator with one
//          fma per value (sum(w) per channel is a constant of the weights).  No per-pixel conversion arithmetic beyond building
//          the halves (one v_perm / v_or + one v_pk_add_f16 per two pixels).
//   conv2 (K = 64) / conv3 (K = 96): activations scaled by a power of two and split into two fp16 pieces (hi = fp16(v),
//          lo = fp16(v - hi): 2^-24 relative), weights likewise at load; three products per k step (hi W0, lo W0, hi W1).
//          conv2's outputs ARE features: their pieces (feature scale of plan 2) are at once conv3's B operand and what is
//          stored for FC1 -- the 128 VALU of plan 2's split epilogue are not paid twice.
// 98 matrix instructions of 16 cycles per task instead of 240 of 32; the task is VALU-bound now (~680 VALU: affine + leaky-ReLU +
// split of the 64 conv1 outputs per lane is half of it).
// Numerics: not bit-identical to the oracle (different rounding points), fp32-class: the features agree with the oracle's to
// ~1e-6 relative to their scale; probabilities within the north star's 1e-4 (tests/test_gpu_fast_plan.py, plan 3).
#include <hip/hip_runtime.h>

#include "ethcnn_kernels.h"
#include "ethcnn_trunk_task.h"

namespace ethcnn {

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define MFMA16H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16((a), (b), (c), 0, 0, 0)
#define MFMA32H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// eight (already scaled) fp32 values -> fp16 pieces hi = fp16(v), lo = fp16(v - hi), packed in pairs
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f16x2 h = __builtin_convertvector((f32x2){v[2 * i], v[2 * i + 1]}, f16x2);
        hi[i] = __builtin_bit_cast(unsigned, h);
        // v - hi with the half read straight out of the packed register (v_fma_mix_f32: hi * -1.0 + v, exact like the subtraction):
        // one instruction instead of v_cvt_f32_f16 + v_sub_f32
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi[i]), "v"(v[2 * i]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi[i]), "v"(v[2 * i + 1]));
        const f16x2 l = __builtin_convertvector((f32x2){r0, r1}, f16x2);
        lo[i] = __builtin_bit_cast(unsigned, l);
    }
}
__device__ __forceinline__ float lrelu1(float h) { return fmaxf(0.2f * h, h); }

// two packed 16-bit integers n < 1024 -> two halves holding n exactly: 0x6400 | n is the half 1024 + n
__device__ __forceinline__ unsigned minus_1024(unsigned two_halves) {
    typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
    const hh2 v = __builtin_bit_cast(hh2, two_halves) - (hh2){(_Float16)1024.0f, (_Float16)1024.0f};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned ints_to_halves(unsigned packed_u16) { return minus_1024(packed_u16 | 0x64006400u); }

template <int BR>
__device__ __forceinline__ void trunk16_run(const uint4* __restrict__ X, int ntasks, int wave, int nwaves, const uint16_t* __restrict__ wimg,
                                            const float* __restrict__ cfrag, float C1, float U2, float U3, char* __restrict__ F, int N,
                                            char* lds) {
    using T0 = Trunk<BR, false>;
    constexpr int POOL = T0::POOL, NB = T0::NB, NJ = T0::NJ;
    constexpr float SCALE = T0::SCALE;
    constexpr int FCH = 2 * 1024;  // plan 2's pair image: two pieces per chunk
    const int lane = threadIdx.x & 63;
    const int col = lane & 15, g = lane >> 4;

    // conv2 / conv3 A fragments -> LDS (20 KB), once per block
    {
        const uint4* src = reinterpret_cast<const uint4*>(wimg + kTrunk16Conv2At);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (int i = threadIdx.x; i < 20 * 64; i += 256) dst[i] = src[i];
    }
    // conv1 A pieces (piece 0, piece 1, and both x 16 for the L branch's high digit) and the per-channel constants: registers
    h4 A1[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) A1[f] = *reinterpret_cast<const h4*>(wimg + (f * 64 + lane) * 4);
    float m1[4], b1s[4], b2s[2][4], b3s[2][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        m1[r] = cfrag[(0 + r) * 64 + lane];
        b1s[r] = cfrag[(4 + r) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            b2s[t][r] = cfrag[(8 + t * 4 + r) * 64 + lane];
            b3s[t][r] = cfrag[(16 + t * 4 + r) * 64 + lane];
        }
    }
    __syncthreads();
    if (wave >= ntasks) return;
    const char* a2_lds = lds + lane * 16;           // conv2 fragment (t, s, p) at ((t * 2 + s) * 2 + p) KiB
    const char* a3_lds = lds + 8 * 1024 + lane * 16;  // conv3 fragment (t, s, p) at ((t * 3 + s) * 2 + p) KiB

    const int lane16 = lane * 16;
    const int lane_off = (g >> 1) * FCH + (g & 1) * 512 + col * 16;  // [chunk][piece][k half][row][8 x 16 bit] (ethcnn_spec.h)
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(X), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(F, 0, -1, 0x00020000);
    uint4 raw[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) raw[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rX, lane16, (wave * NJ + j) * 1024, 0));

    for (int task = wave; task < ntasks; task += nwaves) {
        int T = T0::raw_sum(raw);
        T += __shfl_xor(T, 16);
        T += __shfl_xor(T, 32);
        const float mean = px_value<false>(T, 256 * POOL * POOL) * (SCALE * (1.0f / 256.0f));  // the exact plan's block mean
        float hb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) hb[r] = fmaf(mean, m1[r], b1s[r]);  // S1 (b1 - mean sum(w))

        int grp, by, bx;  // wave-uniform
        if (BR == 0) { grp = task >> 4; by = (task >> 2) & 3; bx = task & 3; }
        else if (BR == 1) { grp = task >> 2; by = (task >> 1) & 1; bx = task & 1; }
        else { grp = task; by = 0; bx = 0; }
        (void)by; (void)bx; (void)NB;
        const bool valid = grp * 16 + col < N;
        const int Tpos = (BR == 0) ? (task & 15) : (BR == 1 ? 16 + (task & 3) : 20);
        const int Fb = (grp >> 1) * (kFastChunks * FCH) + Tpos * 8 * FCH + (grp & 1) * 256;

        f32x4 a2f[4][2];
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
            // ---- conv1 of the four patches of position q2: B operand = the pixel sums as halves
            f32x4 acc1[4];
#pragma unroll
            for (int q1 = 0; q1 < 4; ++q1) {
                acc1[q1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (BR == 0) {
                    const unsigned wq = q1 == 0 ? raw[q2].x : (q1 == 1 ? raw[q2].y : (q1 == 2 ? raw[q2].z : raw[q2].w));
                    // bytes kx = 0..3 of row ky = g -> halves 1024 + byte (v_perm with the constant 0x64 as high bytes), then - 1024
                    const unsigned p01 = __builtin_amdgcn_perm(0x64646464u, wq, 0x04010400u), p23 = __builtin_amdgcn_perm(0x64646464u, wq, 0x04030402u);
                    const h4 b = __builtin_bit_cast(h4, (u32x2){minus_1024(p01), minus_1024(p23)});
                    acc1[q1] = MFMA16H(A1[0], b, acc1[q1]);
                    acc1[q1] = MFMA16H(A1[1], b, acc1[q1]);
                } else {
                    const uint4 rw = raw[2 * q2 + (q1 >> 1)];
                    const unsigned w0 = (q1 & 1) ? rw.z : rw.x, w1 = (q1 & 1) ? rw.w : rw.y;  // (kx 0, 1), (kx 2, 3) as packed u16 sums
                    if (BR == 1) {
                        const h4 b = __builtin_bit_cast(h4, (u32x2){ints_to_halves(w0), ints_to_halves(w1)});
                        acc1[q1] = MFMA16H(A1[0], b, acc1[q1]);
                        acc1[q1] = MFMA16H(A1[1], b, acc1[q1]);
                    } else {  // sums up to 4080 = 16 hi + lo: two exact digits, the high one against the x 16 weight pieces
                        const h4 bh = __builtin_bit_cast(h4, (u32x2){ints_to_halves((w0 >> 4) & 0x00ff00ffu), ints_to_halves((w1 >> 4) & 0x00ff00ffu)});
                        const h4 bl = __builtin_bit_cast(h4, (u32x2){ints_to_halves(w0 & 0x000f000fu), ints_to_halves(w1 & 0x000f000fu)});
                        acc1[q1] = MFMA16H(A1[2], bh, acc1[q1]);
                        acc1[q1] = MFMA16H(A1[0], bl, acc1[q1]);
                        acc1[q1] = MFMA16H(A1[3], bh, acc1[q1]);
                        acc1[q1] = MFMA16H(A1[1], bl, acc1[q1]);
                    }
                }
            }
            // ---- affine (mean removal folded in) + leaky-ReLU + split: conv2's B operands of the two k steps (patches 2 s, 2 s + 1)
            u32x4 bhi[2], blo[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = lrelu1(fmaf(acc1[2 * s + (i >> 2)][i & 3], C1, hb[i & 3]));
                split8(v, bhi[s], blo[s]);
            }
            // ---- conv2 of position q2: two M tiles (channels 0..15, 16..23 + pad), two k steps, three products each
            f32x4 acc2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const h8 w0 = *reinterpret_cast<const h8*>(a2_lds + ((t * 2 + s) * 2 + 0) * 1024);
                    const h8 w1 = *reinterpret_cast<const h8*>(a2_lds + ((t * 2 + s) * 2 + 1) * 1024);
                    acc2[t] = MFMA32H(w0, __builtin_bit_cast(h8, bhi[s]), acc2[t]);
                    acc2[t] = MFMA32H(w0, __builtin_bit_cast(h8, blo[s]), acc2[t]);
                    acc2[t] = MFMA32H(w1, __builtin_bit_cast(h8, bhi[s]), acc2[t]);
                }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) a2f[q2][t][r] = lrelu1(fmaf(acc2[t][r], U2, b2s[t][r]));  // features, in plan 2's feature scale
        }
        // the raw registers are dead now: the next task's record is fetched under the rest
        if (task + nwaves < ntasks) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                raw[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rX, lane16, ((task + nwaves) * NJ + j) * 1024, 0));
        }
        // ---- the task's four register pairs (ethcnn_weights.cpp::fast_feature_k): split once -- stored for FC1 AND fed to conv3
        u32x4 phi[3], plo[3];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float v[8] = {a2f[2 * p][0][0], a2f[2 * p][0][1], a2f[2 * p][0][2], a2f[2 * p][0][3],
                                a2f[2 * p + 1][0][0], a2f[2 * p + 1][0][1], a2f[2 * p + 1][0][2], a2f[2 * p + 1][0][3]};
            split8(v, phi[p], plo[p]);
        }
        {   // channels 16..23: positions (2 j, 2 j + 1) packed into the lower / upper lane halves
            float v[8];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float up = __shfl(a2f[2 * j + 1][1][r], lane & 31);  // lanes 32..63 <- lanes 0..31 of position 2 j + 1
                    v[4 * j + r] = (lane < 32) ? a2f[2 * j][1][r] : up;
                }
            split8(v, phi[2], plo[2]);
        }
        if (valid) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                __builtin_amdgcn_raw_buffer_store_b128(phi[p], rF, lane_off + Fb + p * 2 * FCH, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(plo[p], rF, lane_off + Fb + p * 2 * FCH + 1024, 0, 0);
            }
        }
        // ---- conv3: k steps 0, 1 = channels 0..15 of positions (0, 1), (2, 3); k step 2 = the packed quads
        f32x4 acc3[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const h8 w0 = *reinterpret_cast<const h8*>(a3_lds + ((t * 3 + s) * 2 + 0) * 1024);
                const h8 w1 = *reinterpret_cast<const h8*>(a3_lds + ((t * 3 + s) * 2 + 1) * 1024);
                acc3[t] = MFMA32H(w0, __builtin_bit_cast(h8, phi[s]), acc3[t]);
                acc3[t] = MFMA32H(w0, __builtin_bit_cast(h8, plo[s]), acc3[t]);
                acc3[t] = MFMA32H(w1, __builtin_bit_cast(h8, phi[s]), acc3[t]);
            }
        {
            float v[8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * t + r] = lrelu1(fmaf(acc3[t][r], U3, b3s[t][r]));
            u32x4 h3, l3;
            split8(v, h3, l3);
            if (valid) {
                __builtin_amdgcn_raw_buffer_store_b128(h3, rF, lane_off + Fb + 3 * 2 * FCH, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(l3, rF, lane_off + Fb + 3 * 2 * FCH + 1024, 0, 0);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k1_trunk_f16(const uint4* __restrict__ XS, const uint4* __restrict__ XM, const uint4* __restrict__ XL,
                                                    int N, int bS, int bM, const uint16_t* __restrict__ wimg, const float* __restrict__ cfrag,
                                                    Trunk16Scalars sc, char* __restrict__ F) {
    __shared__ __attribute__((aligned(16))) char lds[20 * 1024];  // conv2 + conv3 A fragments of this block's branch
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const int groups = (N + 15) / 16;
    if (b < bS)
        trunk16_run<0>(XS, groups * 16, b * 4 + w, bS * 4, wimg, cfrag, sc.C1[0], sc.U2[0], sc.U3[0], F, N, lds);
    else if (b < bS + bM)
        trunk16_run<1>(XM, groups * 4, (b - bS) * 4 + w, bM * 4, wimg + kTrunk16Halves, cfrag + kTrunk16Consts, sc.C1[1], sc.U2[1], sc.U3[1], F, N, lds);
    else
        trunk16_run<2>(XL, groups, (b - bS - bM) * 4 + w, (int)(gridDim.x - bS - bM) * 4, wimg + 2 * kTrunk16Halves, cfrag + 2 * kTrunk16Consts,
                       sc.C1[2], sc.U2[2], sc.U3[2], F, N, lds);
}

void launch_trunk_f16(const Workspace& ws, const DeviceWeights& w, int n, hipStream_t s) {
    // tasks per group: 16 S, 4 M, 1 L; blocks per CU by registers (see the resource remark of the build); same branch shares as k1_trunk
    const int groups = (n + 15) / 16, tS = groups * 16, tM = groups * 4, tL = groups;
    auto blocks = [](int tasks, int budget) { int b = (tasks + 3) / 4; return b < budget ? b : budget; };
    constexpr int per_cu = 4;
    const int bS = blocks(tS, 193 * per_cu), bM = blocks(tM, 50 * per_cu), bL = blocks(tL, 13 * per_cu);
    hipLaunchKernelGGL(k1_trunk_f16, dim3(bS + bM + bL), dim3(256), 0, s, ws.xs, ws.xm, ws.xl, n, bS, bM, w.trunk16_w, w.trunk16_c, w.trunk16_s,
                       reinterpret_cast<char*>(ws.featb));
}

}  // namespace ethcnn
