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
//          product is exact, and  sum w (c s - mean) + b  =  sum((c w) s) + (b - mean sum(w)):  the pieces are those of c w (one
//          rounding of the weight) and the accumulator starts at the per-task constant, so conv1 costs no VALU per output at all
//          (sum(w) per channel is a constant of the weights).  No per-pixel conversion arithmetic beyond building the halves (one
//          v_perm / v_or + one v_pk_add_f16 per two pixels).
//   conv2 (K = 64) / conv3 (K = 96): activations scaled by a power of two and split into two fp16 pieces (hi = fp16(v),
//          lo = fp16(v - hi): 2^-24 relative), weights likewise at load; three products per k step (hi W0, lo W0, hi W1).
//          conv2's outputs ARE features: their pieces (feature scale of plan 2) are at once conv3's B operand and what is
//          stored for FC1 -- the 128 VALU of plan 2's split epilogue are not paid twice.
// 98 matrix instructions of 16 cycles per task instead of 240 of 32; the task is bound by instruction ISSUE now (~590 VALU of ~800
// instructions: leaky-ReLU + split of the 64 conv1 outputs per lane is half of them; profiles/r05_trunk16_issue_bound.txt).
// Numerics: not bit-identical to the oracle (different rounding points), fp32-class: the features agree with the oracle's to
// ~1e-6 relative to their scale; probabilities within the north star's 1e-4 (tests/test_gpu_fast_plan.py, plan 3).
#include <hip/hip_runtime.h>

#include "ethcnn_kernels.h"
#include "ethcnn_tile_group.h"
#include "ethcnn_trunk_task.h"

namespace ethcnn {

#ifdef TRUNK16_STAMPS  // probe build only (scripts/ubench/trunk16_probe.hip): shader-clock stamps of one wave's way through the kernel
__device__ unsigned long long g_t16_stamps[2][4][512];
__device__ int g_t16_block[2] = {0, 1};
struct T16Stamper {
    unsigned long long* p = nullptr;
    int n = 0;
    __device__ __forceinline__ void init() {
        const int sel = (int)blockIdx.x == g_t16_block[0] ? 0 : ((int)blockIdx.x == g_t16_block[1] ? 1 : -1);
        if (sel >= 0 && (threadIdx.x & 63) == 0) p = g_t16_stamps[sel][threadIdx.x >> 6];
    }
    __device__ __forceinline__ void operator()(int tag) {
        if (p && n < 512) p[n++] = ((unsigned long long)tag << 56) | (__builtin_readcyclecounter() & 0x00ffffffffffffffull);
    }
};
#define T16_STAMP(st, tag) (st)(tag)
#define T16_TASK_STAMP(tag) do { if (stp) (*stp)(tag); } while (0)
#else
#define T16_STAMP(st, tag)
#define T16_TASK_STAMP(tag)
#endif

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
        // one instruction instead of v_cvt_f32_f16 + v_sub_f32.  (v_fma_mixlo_f16 / v_fma_mixhi_f16 -- the difference rounded to half in the
        // same instruction, one instruction less per pair, same bits -- measured SLOWER: trunk 0.425 against 0.40 ms, profiles/r05_trunk16_issue_bound.txt)
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi[i]), "v"(v[2 * i]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi[i]), "v"(v[2 * i + 1]));
        const f16x2 l = __builtin_convertvector((f32x2){r0, r1}, f16x2);
        lo[i] = __builtin_bit_cast(unsigned, l);
    }
}
__device__ __forceinline__ float lrelu1(float h) { return fmaxf(0.2f * h, h); }
// four accumulator values -> leaky-ReLU(acc * u + b) with the affine step and the 0.2 h product as packed fp32 instructions
// (v_pk_fma_f32 / v_pk_mul_f32: two values per issue slot; same IEEE results as the scalar forms) -- the task is bound by VALU issue
// v_max_f32 as it is: fmaxf() on a raw MFMA result makes the compiler canonicalise the operand first (one more v_max per value)
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void affine_lrelu4(const f32x4 acc, const float u, const float (&b)[4], float* v) {
    const f32x2 uu = {u, u};
    const f32x2 h01 = __builtin_elementwise_fma((f32x2){acc[0], acc[1]}, uu, (f32x2){b[0], b[1]});
    const f32x2 h23 = __builtin_elementwise_fma((f32x2){acc[2], acc[3]}, uu, (f32x2){b[2], b[3]});
    const f32x2 t01 = h01 * 0.2f, t23 = h23 * 0.2f;
    v[0] = vmax(t01[0], h01[0]); v[1] = vmax(t01[1], h01[1]); v[2] = vmax(t23[0], h23[0]); v[3] = vmax(t23[1], h23[1]);
}
__device__ __forceinline__ void lrelu4(const f32x4 h, float* v) {
    const f32x2 t01 = (f32x2){h[0], h[1]} * 0.2f, t23 = (f32x2){h[2], h[3]} * 0.2f;
    v[0] = vmax(t01[0], h[0]); v[1] = vmax(t01[1], h[1]); v[2] = vmax(t23[0], h[2]); v[3] = vmax(t23[1], h[3]);
}

// two packed 16-bit integers n < 1024 -> two halves holding n exactly: 0x6400 | n is the half 1024 + n
__device__ __forceinline__ unsigned minus_1024(unsigned two_halves) {
    typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
    const hh2 v = __builtin_bit_cast(hh2, two_halves) - (hh2){(_Float16)1024.0f, (_Float16)1024.0f};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned ints_to_halves(unsigned packed_u16) { return minus_1024(packed_u16 | 0x64006400u); }

// One wave's view of branch BR: weights / constants in registers + LDS (setup, once per block), then task() per unit position.
template <int BR>
struct Trunk16 {
    using T0 = Trunk<BR, false>;
    static constexpr int POOL = T0::POOL, NJ = T0::NJ;
    static constexpr float SCALE = T0::SCALE;
    static constexpr int FCH = 2 * 1024;  // plan 2's pair image: two pieces per chunk

    h4 A1[4];
    float m1[4], b1s[4], b2s[2][4], b3s[2][4];
    const char *a2_lds, *a3_lds;
    int lane, col, g, lane_off, N;
    float C1, U2, U3;
    __amdgpu_buffer_rsrc_t rF;
#ifdef TRUNK16_STAMPS
    T16Stamper* stp = nullptr;
#endif

    // conv2 / conv3 A fragments of the branch -> LDS (20 KB); every thread of the 256-thread block; the caller puts a barrier behind it
    static __device__ __forceinline__ void stage(const uint16_t* __restrict__ wimg, char* lds) {
        const uint4* src = reinterpret_cast<const uint4*>(wimg + kTrunk16Conv2At);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (int i = threadIdx.x; i < 20 * 64; i += 256) dst[i] = src[i];
    }
    // conv1 A pieces (piece 0, piece 1, and both x 16 for the L branch's high digit) and the per-channel constants: registers
    __device__ __forceinline__ void load_consts(const uint16_t* __restrict__ wimg, const float* __restrict__ cfrag, float c1, float u2, float u3,
                                                char* __restrict__ F, int n, char* lds) {
        lane = threadIdx.x & 63;
        col = lane & 15;
        g = lane >> 4;
        N = n; C1 = c1; U2 = u2; U3 = u3;
#pragma unroll
        for (int f = 0; f < 4; ++f) A1[f] = *reinterpret_cast<const h4*>(wimg + (f * 64 + lane) * 4);
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
        a2_lds = lds + lane * 16;             // conv2 fragment (t, s, p) at ((t * 2 + s) * 2 + p) KiB
        a3_lds = lds + 8 * 1024 + lane * 16;  // conv3 fragment (t, s, p) at ((t * 3 + s) * 2 + p) KiB
        lane_off = (g >> 1) * FCH + (g & 1) * 512 + col * 16;  // [chunk][piece][k half][row][8 x 16 bit] (ethcnn_spec.h)
        rF = __builtin_amdgcn_make_buffer_rsrc(F, 0, -1, 0x00020000);
    }
    // every thread of the 256-thread block; ends with a barrier (the fragments are in `lds`, 20 KB)
    __device__ __forceinline__ void setup(const uint16_t* __restrict__ wimg, const float* __restrict__ cfrag, float c1, float u2, float u3,
                                          char* __restrict__ F, int n, char* lds) {
        stage(wimg, lds);
        load_consts(wimg, cfrag, c1, u2, u3, F, n, lds);
        __syncthreads();
    }

    // one task (unit position `task` of its group; `raw` = the lane's pixel record).  mid(): called once the record registers are
    // dead -- the caller's prefetch of the next record lands in them, under the rest of the task.
    template <class Mid>
    __device__ __forceinline__ void task(uint4 (&raw)[NJ], int task, Mid mid) const {
        int T = T0::raw_sum(raw);
        T += __shfl_xor(T, 16);
        T += __shfl_xor(T, 32);
        const float mean = px_value<false>(T, 256 * POOL * POOL) * (SCALE * (1.0f / 256.0f));  // the exact plan's block mean
        f32x4 hb;  // S1 (b1 - mean sum(w)): conv1's accumulators START here (the weight pieces carry c255 S1, ethcnn_weights.cpp), so that
                   // what the matrix pipe leaves in them is conv1's output before the leaky-ReLU -- no affine step per value
#pragma unroll
        for (int r = 0; r < 4; ++r) hb[r] = fmaf(mean, m1[r], b1s[r]);

        const int grp = (BR == 0) ? (task >> 4) : (BR == 1 ? (task >> 2) : task);  // wave-uniform
#ifdef TRUNK16_PROBE_NO_STORE  // probe build only (scripts/build_variant.sh): the whole task without its feature stores (N < 0 never holds)
        const bool valid = grp * 16 + col < N && N < 0;
#else
        const bool valid = grp * 16 + col < N;
#endif
        const int Tpos = (BR == 0) ? (task & 15) : (BR == 1 ? 16 + (task & 3) : 20);
        const int Fb = (grp >> 1) * (kFastChunks * FCH) + Tpos * 8 * FCH + (grp & 1) * 256;

        f32x4 a2f[4][2];
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
            // ---- conv1 of the four patches of position q2: B operand = the pixel sums as halves
            f32x4 acc1[4];
#pragma unroll
            for (int q1 = 0; q1 < 4; ++q1) {
                acc1[q1] = hb;
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
                lrelu4(acc1[2 * s], v);
                lrelu4(acc1[2 * s + 1], v + 4);
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
            for (int t = 0; t < 2; ++t) {  // features, in plan 2's feature scale
                float v[4];
                affine_lrelu4(acc2[t], U2, b2s[t], v);
                a2f[q2][t] = (f32x4){v[0], v[1], v[2], v[3]};
            }
        }
        T16_TASK_STAMP(3);
        mid();  // the raw registers are dead now: the next record is fetched under the rest
        T16_TASK_STAMP(4);
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
        T16_TASK_STAMP(5);
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
            affine_lrelu4(acc3[0], U3, b3s[0], v);
            affine_lrelu4(acc3[1], U3, b3s[1], v + 4);
            u32x4 h3, l3;
            split8(v, h3, l3);
            if (valid) {
                __builtin_amdgcn_raw_buffer_store_b128(h3, rF, lane_off + Fb + 3 * 2 * FCH, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(l3, rF, lane_off + Fb + 3 * 2 * FCH + 1024, 0, 0);
            }
        }
    }
};

// records from the tile stage's buffers (XS / XM / XL): tasks wave, wave + nwaves, ...
template <int BR>
__device__ __forceinline__ void trunk16_run(const uint4* __restrict__ X, int ntasks, int wave, int nwaves, const uint16_t* __restrict__ wimg,
                                            const float* __restrict__ cfrag, float C1, float U2, float U3, char* __restrict__ F, int N,
                                            char* lds) {
    Trunk16<BR> tk;
    constexpr int NJ = Trunk16<BR>::NJ;
    tk.setup(wimg, cfrag, C1, U2, U3, F, N, lds);
    if (wave >= ntasks) return;
    const int lane16 = tk.lane * 16;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(X), 0, -1, 0x00020000);
    uint4 raw[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) raw[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rX, lane16, (wave * NJ + j) * 1024, 0));
    for (int task = wave; task < ntasks; task += nwaves)
        tk.task(raw, task, [&]() {
            if (task + nwaves < ntasks) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    raw[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rX, lane16, ((task + nwaves) * NJ + j) * 1024, 0));
            }
        });
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

// ---- plan 3 with the CTU-load stage FOLDED IN (round 5).  Under plan 3 a step moves ~4.6 GB through HBM (profiles/r04_pmc_c3.txt),
// 1.36 GB of it the pixel records' round trip (k0_tile_slab writes 6,656 B per CTU, the trunk reads them back), and the tile stage
// beside FC1 costs the 16-bit FC1 0.15 ms.  Here the S branch (16 of a group's 21 tasks, 4,096 of the 6,656 record bytes) takes its
// records straight out of the LDS slab the loader role fills -- the same coalesced 1 KiB-per-wave-instruction frame reads and the
// same exact integer record words as k0_tile_slab (ethcnn_tile_group.h) -- and the block writes only the slab's XM / XL records
// (2,560 B per CTU) for the M / L tasks, which run as a second, small launch (k1_trunk_f16 with bS = 0).  No XS buffer, no tile
// launch, no side stream.
//   work item = one 16-row slab of one group (4 per group): load -> LDS -> barrier -> wave w reads the record of unit (uy = slab,
//   ux = w), the block emits the slab's 2.5 KiB of XM / XL records -> barrier -> S task; the next item's pixels are requested at the
//   task's mid point into the dead record registers.
template <bool FAST>
__global__ __launch_bounds__(256) void k1_trunk_f16_fold(const uint8_t* __restrict__ luma, int width, int height, long pitch, long frame_stride,
                                                         int cw, int nctu, int f0, int r0, int N, uint4* __restrict__ XM, uint4* __restrict__ XL,
                                                         int* __restrict__ gate_flags, int n_flags, const uint16_t* __restrict__ wimg,
                                                         const float* __restrict__ cfrag, Trunk16Scalars sc, char* __restrict__ F) {
    __shared__ __attribute__((aligned(16))) char lds[20 * 1024];
    __shared__ uint32_t tile[16 * kSlabCtuPitch];
    if (blockIdx.x == 0)  // (what the tile stage does on the way: the pass's sync area, read by the heads / gate launches behind us)
        for (int i = threadIdx.x; i < n_flags; i += 256) gate_flags[i] = 0;
    const int t = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    Trunk16<0> tk;
    tk.setup(wimg, cfrag, sc.C1[0], sc.U2[0], sc.U3[0], F, N, lds);
    const int nitems = ((N + 15) >> 4) * 4;
    int it = blockIdx.x;
    if (it >= nitems) return;
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(XM, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc(XL, 0, -1, 0x00020000);
    SlabLoader L;
    uint4 pre[4];
    L.init32(t, luma, width, frame_stride, cw, nctu, f0, r0, N, (it >> 2) * 16);
    L.template load<FAST>(it & 3, pre, width, height, pitch);
#pragma unroll 1
    for (; it < nitems; it += gridDim.x) {
        const int grp = it >> 2, s = it & 3;
        L.to_lds(tile, pre);
        __syncthreads();
        uint4 raw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = slab_xs_record(tile, tk.col, tk.g, j, w);
        // the slab's XM / XL records (same index arithmetic as tile_group): 512 + 128 uint4, 2.5 per thread
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int e = t + 256 * rep;
            const int lane = e & 63, jj = (e >> 6) & 3, ux = e >> 8;
            const int j = 4 * (s & 1) + jj, unit = 2 * (s >> 1) + ux;
            const uint4 v = slab_xm_record(tile, lane & 15, lane >> 4, j, ux);
            // (whole offset in the VGPR, soffset = 0: see the store-hazard remark in ethcnn_trunk_task.h)
            __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){v.x, v.y, v.z, v.w}, rM, (grp * 2048 + 512 * unit + 64 * j + lane) * 16, 0, 0);
        }
        if (t < 128) {
            const int lane = t & 63, m = t >> 6;
            const int j = 4 * (s >> 1) + 2 * m + (s & 1);
            const uint4 v = slab_xl_record(tile, lane & 15, lane >> 4, j);
            __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){v.x, v.y, v.z, v.w}, rL, (grp * 512 + 64 * j + lane) * 16, 0, 0);
        }
        __syncthreads();  // the slab is consumed: the next item may overwrite it
        tk.task(raw, grp * 16 + 4 * s + w, [&]() {
            const int nx = it + (int)gridDim.x;
            if (nx < nitems) {
                L.init32(t, luma, width, frame_stride, cw, nctu, f0, r0, N, (nx >> 2) * 16);
                L.template load<FAST>(nx & 3, pre, width, height, pitch);
            }
        });
    }
}

// ---- the whole trunk of plan 3 behind ONE pass over the luma frames (round 5).  k1_trunk_f16_fold above still writes and re-reads
// 2,560 B of XM / XL records per CTU and pays a second launch for the M / L tasks.  Here a block takes a whole GROUP: it walks the four
// slabs, every wave runs the slab's S task of its unit column, and the pooled records of the M / L units are built from the same LDS slab
// straight into the registers of the wave that will run that task -- waves 0 / 1 the M units (uy, ux = w) after slabs 1 and 3, wave 2
// the L unit after slab 3.  HBM traffic of the trunk: 4,096 B in, 10,752 B of feature pieces out per CTU (1.51 GB per C3 step against
// 2.87 GB for tile stage + trunk in round 4).  21 tasks in 24 wave slots per group; 77.5 KB of LDS (three branches' fragments + the
// slab), 255 VGPRs: two blocks per CU.
// (Measured and not kept, profiles/r05_plan3_fold.txt: one M task per wave -- 20 tasks in 20 slots -- with the L unit's records written
// to XL and its tasks as a small second launch: the second launch costs more than the idle slots, trunk 0.436 against 0.410 ms.)
template <bool FAST>
__global__ __launch_bounds__(256, 2) void k1_trunk_f16_foldall(const uint8_t* __restrict__ luma, int width, int height, long pitch, long frame_stride,
                                                            int cw, int nctu, int f0, int r0, int N, int* __restrict__ gate_flags, int n_flags,
                                                            const uint16_t* __restrict__ wimg, const float* __restrict__ cfrag, Trunk16Scalars sc,
                                                            char* __restrict__ F) {
    __shared__ __attribute__((aligned(16))) char lds[3][20 * 1024];
    __shared__ uint32_t tile[16 * kSlabCtuPitch];
    if (blockIdx.x == 0)  // (what the tile stage does on the way: the pass's sync area, read by the heads / gate launches behind us)
        for (int i = threadIdx.x; i < n_flags; i += 256) gate_flags[i] = 0;
    const int t = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    Trunk16<0>::stage(wimg, lds[0]);
    Trunk16<1>::stage(wimg + kTrunk16Halves, lds[1]);
    Trunk16<2>::stage(wimg + 2 * kTrunk16Halves, lds[2]);
    Trunk16<0> tk;
    tk.load_consts(wimg, cfrag, sc.C1[0], sc.U2[0], sc.U3[0], F, N, lds[0]);
#ifdef TRUNK16_STAMPS
    T16Stamper st;
    st.init();
    tk.stp = &st;
#endif
    __syncthreads();
    const int ngroups = (N + 15) >> 4;
    int grp = blockIdx.x;
    if (grp >= ngroups) return;
    SlabLoader L;
    uint4 pre[4];
    L.init32(t, luma, width, frame_stride, cw, nctu, f0, r0, N, grp * 16);
    L.template load<FAST>(0, pre, width, height, pitch);
#pragma unroll 1
    for (; grp < ngroups; grp += gridDim.x) {
        const int r = w;  // (rotating the M / L roles over the waves from group to group -- SIMDs 0 / 1 run 6 tasks per group, SIMD 3 four -- was
                          // measured: no change, profiles/r05_trunk16_issue_bound.txt)
        uint4 rawx[8];  // waves 0 / 1: the record of M unit (uy, w), rebuilt per slab pair; wave 2: the L unit's, over the four slabs
#pragma unroll 1
        for (int uy = 0; uy < 2; ++uy) {
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                const int s = 2 * uy + sh;
                T16_STAMP(st, 0);
                L.to_lds(tile, pre);
                T16_STAMP(st, 1);
                __syncthreads();
                T16_STAMP(st, 2);
                uint4 raw[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) raw[j] = slab_xs_record(tile, tk.col, tk.g, j, w);
                if (r < 2) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) rawx[4 * sh + jj] = slab_xm_record(tile, tk.col, tk.g, 4 * sh + jj, r);
                } else if (r == 2) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {  // record j = 4 uy + 2 m + sh (the columns depend on m only)
                        const uint4 rec = slab_xl_record(tile, tk.col, tk.g, 2 * m + sh);
                        if (uy == 0) rawx[2 * m + sh] = rec;
                        else rawx[4 + 2 * m + sh] = rec;
                    }
                }
                __syncthreads();  // the slab is consumed: the next one may overwrite it
                T16_STAMP(st, 6);
                tk.task(raw, grp * 16 + 4 * s + w, [&]() {
#ifdef TRUNK16_PROBE_NO_LOAD  // probe build only: the pixels of the first slab over and over (how much of the task waits for the next slab?)
                    if (N < 0)
#endif
                    if (s < 3) {
                        L.template load<FAST>(s + 1, pre, width, height, pitch);
                    } else {
                        const int nx = grp + (int)gridDim.x;
                        if (nx < ngroups) {
                            L.init32(t, luma, width, frame_stride, cw, nctu, f0, r0, N, nx * 16);
                            L.template load<FAST>(0, pre, width, height, pitch);
                        }
                    }
                });
            }
            if (r < 2) {
                Trunk16<1> tm;
                tm.load_consts(wimg + kTrunk16Halves, cfrag + kTrunk16Consts, sc.C1[1], sc.U2[1], sc.U3[1], F, N, lds[1]);
                T16_STAMP(st, 7);
                tm.task(rawx, grp * 4 + 2 * uy + r, []() {});
                T16_STAMP(st, 8);
            }
        }
        if (r == 2) {
            Trunk16<2> tl;
            tl.load_consts(wimg + 2 * kTrunk16Halves, cfrag + 2 * kTrunk16Consts, sc.C1[2], sc.U2[2], sc.U3[2], F, N, lds[2]);
            T16_STAMP(st, 9);
            tl.task(rawx, grp, []() {});
            T16_STAMP(st, 10);
        }
    }
}

void launch_trunk_f16(const Workspace& ws, const DeviceWeights& w, int n, hipStream_t s, bool ml_only) {
    // tasks per group: 16 S, 4 M, 1 L; blocks per CU by registers (see the resource remark of the build); same branch shares as k1_trunk
    const int groups = (n + 15) / 16, tS = groups * 16, tM = groups * 4, tL = groups;
    auto blocks = [](int tasks, int budget) { int b = (tasks + 3) / 4; return b < budget ? b : budget; };
    constexpr int per_cu = 4;
    // ml_only (the S branch ran in k1_trunk_f16_fold): the whole grid to the M / L tasks, 4 : 1, three resident blocks per CU (~8 tasks
    // per wave: a fourth, non-resident block per CU would be a second round for a quarter of the work)
    const int bS = ml_only ? 0 : blocks(tS, 193 * per_cu), bM = blocks(tM, ml_only ? 205 * 3 : 50 * per_cu), bL = blocks(tL, ml_only ? 51 * 3 : 13 * per_cu);
    hipLaunchKernelGGL(k1_trunk_f16, dim3(bS + bM + bL), dim3(256), 0, s, ws.xs, ws.xm, ws.xl, n, bS, bM, w.trunk16_w, w.trunk16_c, w.trunk16_s,
                       reinterpret_cast<char*>(ws.featb));
}

void launch_trunk_f16_foldall(const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, const Workspace& ws, const DeviceWeights& w, int n_flags,
                              hipStream_t s, int blocks_per_cu) {
    const int groups = (n + 15) / 16;
    const int cap = 256 * (blocks_per_cu > 0 ? blocks_per_cu : 2);
    const int blocks = groups < cap ? groups : cap;
    const bool fast = (g.width % 16 == 0) && (g.pitch % 16 == 0) && (g.frame_stride % 16 == 0) && (reinterpret_cast<uintptr_t>(d_luma) % 16 == 0);
    const int f0 = (int)(ctu0 / g.nctu), r0 = (int)(ctu0 % g.nctu);
    if (fast)
        hipLaunchKernelGGL(k1_trunk_f16_foldall<true>, dim3(blocks), dim3(256), 0, s, d_luma, g.width, g.height, g.pitch, g.frame_stride, g.cw, g.nctu,
                           f0, r0, n, ws.flags, n_flags, w.trunk16_w, w.trunk16_c, w.trunk16_s, reinterpret_cast<char*>(ws.featb));
    else
        hipLaunchKernelGGL(k1_trunk_f16_foldall<false>, dim3(blocks), dim3(256), 0, s, d_luma, g.width, g.height, g.pitch, g.frame_stride, g.cw, g.nctu,
                           f0, r0, n, ws.flags, n_flags, w.trunk16_w, w.trunk16_c, w.trunk16_s, reinterpret_cast<char*>(ws.featb));
}

void launch_trunk_f16_fold(const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, const Workspace& ws, const DeviceWeights& w, int n_flags,
                           hipStream_t s) {
    const int items = ((n + 15) / 16) * 4;
    const int blocks = items < 768 ? items : 768;  // 3 per CU (registers), persistent over the slab items
    const bool fast = (g.width % 16 == 0) && (g.pitch % 16 == 0) && (g.frame_stride % 16 == 0) && (reinterpret_cast<uintptr_t>(d_luma) % 16 == 0);
    const int f0 = (int)(ctu0 / g.nctu), r0 = (int)(ctu0 % g.nctu);  // first frame of the pass / first CTU inside it: 32-bit geometry in the kernel
    if (fast)
        hipLaunchKernelGGL(k1_trunk_f16_fold<true>, dim3(blocks), dim3(256), 0, s, d_luma, g.width, g.height, g.pitch, g.frame_stride, g.cw, g.nctu,
                           f0, r0, n, ws.xm, ws.xl, ws.flags, n_flags, w.trunk16_w, w.trunk16_c, w.trunk16_s, reinterpret_cast<char*>(ws.featb));
    else
        hipLaunchKernelGGL(k1_trunk_f16_fold<false>, dim3(blocks), dim3(256), 0, s, d_luma, g.width, g.height, g.pitch, g.frame_stride, g.cw, g.nctu,
                           f0, r0, n, ws.xm, ws.xl, ws.flags, n_flags, w.trunk16_w, w.trunk16_c, w.trunk16_s, reinterpret_cast<char*>(ws.featb));
}

}  // namespace ethcnn
