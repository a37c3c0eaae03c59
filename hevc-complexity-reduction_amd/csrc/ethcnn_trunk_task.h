// ethcnn_trunk_task.h -- device code of the trunk (one wave = one unit position of 16 CTUs; see ethcnn_trunk.hip for the
// design notes), shared by k1_trunk (ethcnn_trunk.hip) and the single-launch small-pass kernel (ethcnn_small.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "ethcnn_fc1_tile.h"   // f32x4, MFMA16, kAuxSc1
#include "ethcnn_heads_pass.h" // u32x4
#include "ethcnn_kernels.h"

namespace ethcnn {

#ifndef TRUNK_BUF_LOADS
#define TRUNK_BUF_LOADS 1
#endif
#ifndef TRUNK_BUF_STORES
#define TRUNK_BUF_STORES 1
#endif

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

// DIRECT mode (the single-launch small-pass kernel): the task's pixel record is gathered straight from the luma frame --
// the CTU-load stage (ethcnn_tile.hip) folded into the trunk wave that consumes it.  Produces exactly the uint4 records
// k0_tile_slab would have written (same exact integer 2x2 / 4x4 sums, zero padding beyond the frame edge), so everything
// downstream is bit-identical.  Needs 16-byte aligned rows (width, pitch, frame stride and base all multiples of 16).
struct DirectSrc {
    const uint8_t* luma;
    int width, height;
    long pitch, frame_stride;
    int cw, nctu;   // CTUs per frame row / per frame
    long ctu0;      // global raster index of the pass's first CTU
    int n_total;    // CTUs in the pass
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
// FC1 plan 2: a register pair of feature quads (8 values per lane) as two fp16 pieces of the SCALED value (scale = a power of two chosen at weight load so that
// no feature can overflow fp16, ethcnn_weights.cpp::fast_feature_bound): h0 = fp16(a s), h1 = fp16(a s - h0), round to nearest
// even: a s = h0 + h1 to 2^-24 relative.  ~32 VALU per pair.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <bool SC1>
__device__ __forceinline__ void store_pair_f16x2(const f32x4& qa, const f32x4& qb, float scale, __amdgpu_buffer_rsrc_t rF, int voff) {
    float r[8] = {qa[0] * scale, qa[1] * scale, qa[2] * scale, qa[3] * scale, qb[0] * scale, qb[1] * scale, qb[2] * scale, qb[3] * scale};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        u32x4 P;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f16x2 h = __builtin_convertvector((f32x2){r[2 * i], r[2 * i + 1]}, f16x2);
            P[i] = __builtin_bit_cast(unsigned, h);
            if (p == 0) {
                r[2 * i] -= (float)h[0];
                r[2 * i + 1] -= (float)h[1];
            }
        }
        __builtin_amdgcn_raw_buffer_store_b128(P, rF, voff + p * 1024, 0, SC1 ? kAuxSc1 : 0);
    }
}

// FAST = FC1 plan the features are written for: 0 fp32 group images (feat), 2 fp16 x 2 pieces (featb)
template <int BR, bool RESI, bool DIRECT = false, bool SC1 = false, int FAST = 0>
struct Trunk {
    static constexpr int FCH = 2 * 1024;  // bytes of one chunk record of a pair image (2 pieces x 1 KiB)
    static constexpr int POOL = (BR == 0) ? 1 : (BR == 1 ? 2 : 4);
    static constexpr float SCALE = 1.0f / (float)(POOL * POOL);
    static constexpr float C255S = (1.0f / 255.0f) * SCALE;  // exact: SCALE is a power of two
    static constexpr int NB = (BR == 0) ? 4 : (BR == 1 ? 2 : 1);
    static constexpr int OFF2 = (BR == 0) ? 672 : (BR == 1 ? 2208 : 2592);
    static constexpr int OFF3 = (BR == 0) ? 0 : (BR == 1 ? 512 : 640);
    static constexpr int NJ = (BR == 0) ? 4 : 8;  // uint4 records per lane per task
    // resi: the preprocessed value of a pixel sum, (s - 128 cnt) / 255.0 * 10 (* SCALE), is a true fp32 DIVISION per value
    // (~12 VALU instructions; 64 of them per lane and task: 1.5 us of a single-picture launch's trunk phase).  A pixel sum has
    // only 255 cnt + 1 values, so the block tabulates them once -- with exactly that expression -- in LDS behind its weight
    // fragments and the gather area, and the task loop looks them up: bit-identical by construction.
    static constexpr int TAB = 255 * POOL * POOL + 1;  // 256 / 1021 / 4081 entries

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
    static __device__ __forceinline__ void decode_q2(const uint4 (&raw)[NJ], int q2, float (&x)[4][4], const float* tab) {
        if (BR == 0) {
            const uint32_t w[4] = {raw[q2].x, raw[q2].y, raw[q2].z, raw[q2].w};
#pragma unroll
            for (int q1 = 0; q1 < 4; ++q1)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int s = (int)((w[q1] >> (8 * kx)) & 0xff);
                    x[q1][kx] = RESI ? tab[s] : (float)s;
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
                        x[2 * jj + hh][kx] = RESI ? tab[s] : (float)s;
                    }
            }
        }
    }

    // ---- DIRECT gather.  Lane (c = CTU in group, g) of task `task`: where its CTU lies in the frame
    struct LaneCtu {
        const uint8_t* base;  // pixel (0, 0) of the lane's CTU; null = CTU beyond the pass: all zero
        int y0, x0, by, bx;
    };
    static __device__ __forceinline__ LaneCtu lane_ctu(const DirectSrc& S, int task, int lane) {
        const int c = lane & 15;
        int grp;
        LaneCtu L;
        if (BR == 0) { grp = task >> 4; L.by = (task >> 2) & 3; L.bx = task & 3; }
        else if (BR == 1) { grp = task >> 2; L.by = (task >> 1) & 1; L.bx = task & 1; }
        else { grp = task; L.by = 0; L.bx = 0; }
        const int n = grp * 16 + c;
        L.base = nullptr;
        L.y0 = L.x0 = 0;
        if (n < S.n_total) {
            const long gn = S.ctu0 + n;
            const long f = gn / S.nctu;
            const int rr = (int)(gn - f * S.nctu);
            const int cy = rr / S.cw, cx = rr - cy * S.cw;
            L.y0 = cy * 64;
            L.x0 = cx * 64;
            L.base = S.luma + f * S.frame_stride + (long)L.y0 * S.pitch + L.x0;
        }
        return L;
    }
    // one uint4 record j of the task (layouts: ethcnn_tile.hip); whole 8 / 16-byte chunks are inside or outside the frame
    // because width % 16 == 0; outside = zero padding (video_to_cu_depth.py:54-57)
    static __device__ __forceinline__ uint4 direct_record(const DirectSrc& S, const LaneCtu& L, int g, int j) {
        auto in = [&](int Y, int X) { return L.base != nullptr && L.y0 + Y < S.height && L.x0 + X < S.width; };
        auto ld8 = [&](int Y, int X) -> uint2 {
            return in(Y, X) ? *reinterpret_cast<const uint2*>(L.base + (long)Y * S.pitch + X) : make_uint2(0u, 0u);
        };
        auto ld16 = [&](int Y, int X) -> uint4 {
            return in(Y, X) ? *reinterpret_cast<const uint4*>(L.base + (long)Y * S.pitch + X) : make_uint4(0u, 0u, 0u, 0u);
        };
        if (BR == 0) {
            const int Y = 16 * L.by + 8 * (j >> 1) + g, X = 16 * L.bx + 8 * (j & 1);
            const uint2 a = ld8(Y, X), b = ld8(Y + 4, X);  // q1 = 0, 1 | q1 = 2, 3
            return make_uint4(a.x, a.y, b.x, b.y);
        }
        uint32_t out[4];
        if (BR == 1) {  // both patch rows of record j (q1 & 1 = 0, 1) are the two halves of ONE 16-byte run per raw row
            const int q2 = j >> 1, pr = 8 * (q2 >> 1) + 4 * (j & 1) + g, pc = 8 * (q2 & 1);  // pooled row / first pooled col
            const uint4 a = ld16(32 * L.by + 2 * pr, 32 * L.bx + 2 * pc), b = ld16(32 * L.by + 2 * pr + 1, 32 * L.bx + 2 * pc);
            const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const uint32_t a0 = aw[2 * hh], a1 = aw[2 * hh + 1], b0 = bw[2 * hh], b1 = bw[2 * hh + 1];
                const uint32_t s0 = __builtin_amdgcn_udot4(a0, 0x00000101u, __builtin_amdgcn_udot4(b0, 0x00000101u, 0u, false), false);
                const uint32_t s1 = __builtin_amdgcn_udot4(a0, 0x01010000u, __builtin_amdgcn_udot4(b0, 0x01010000u, 0u, false), false);
                const uint32_t s2 = __builtin_amdgcn_udot4(a1, 0x00000101u, __builtin_amdgcn_udot4(b1, 0x00000101u, 0u, false), false);
                const uint32_t s3 = __builtin_amdgcn_udot4(a1, 0x01010000u, __builtin_amdgcn_udot4(b1, 0x01010000u, 0u, false), false);
                out[2 * hh] = s0 | (s1 << 16);
                out[2 * hh + 1] = s2 | (s3 << 16);
            }
            return make_uint4(out[0], out[1], out[2], out[3]);
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int dd = 2 * j + hh, q2 = dd >> 2, q1 = dd & 3;
            const int pr = 8 * (q2 >> 1) + 4 * (q1 >> 1) + g, pc = 8 * (q2 & 1) + 4 * (q1 & 1);  // pooled row / first pooled col
            uint32_t sacc[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int ry = 0; ry < 4; ++ry) {
                const uint4 v = ld16(4 * pr + ry, 4 * pc);
                sacc[0] = __builtin_amdgcn_udot4(v.x, 0x01010101u, sacc[0], false);
                sacc[1] = __builtin_amdgcn_udot4(v.y, 0x01010101u, sacc[1], false);
                sacc[2] = __builtin_amdgcn_udot4(v.z, 0x01010101u, sacc[2], false);
                sacc[3] = __builtin_amdgcn_udot4(v.w, 0x01010101u, sacc[3], false);
            }
            out[2 * hh] = sacc[0] | (sacc[1] << 16);
            out[2 * hh + 1] = sacc[2] | (sacc[3] << 16);
        }
        return make_uint4(out[0], out[1], out[2], out[3]);
    }
    // S / M tasks: the wave gathers its own record.  L task (1 KiB of pixels per lane): the block's four waves gather two
    // records each and hand them to wave 0 through LDS (`xl`, 8 KB behind the weight fragments); waves 1..3 then leave
    // (return value false).  All four waves of an L block are called with the SAME task.
    static __device__ __forceinline__ bool load_direct(const DirectSrc& S, int task, int lane, uint4 (&raw)[NJ], uint4* xl) {
        const LaneCtu L = lane_ctu(S, task, lane);
        const int g = lane >> 4;
        if (BR != 2) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) raw[j] = direct_record(S, L, g, j);
            return true;
        }
        const int q = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        xl[(2 * q) * 64 + lane] = direct_record(S, L, g, 2 * q);
        xl[(2 * q + 1) * 64 + lane] = direct_record(S, L, g, 2 * q + 1);
        __syncthreads();
        if (q != 0) return false;
#pragma unroll
        for (int j = 0; j < NJ; ++j) raw[j] = xl[j * 64 + lane];
        return true;
    }

    // ready (PULL form of the single-launch pass): called by every thread once the weights are staged and before the task's pixel
    // records are loaded -- it waits until they exist.  Returns true when the block's LDS was lent out meanwhile (stage again).
    struct NoWait { __device__ __forceinline__ bool operator()() const { return false; } };
    template <class Ready = NoWait>
    static __device__ __forceinline__ void run(const uint4* __restrict__ X, int ntasks, int wave, int nwaves,
                                               const float* __restrict__ wfrag, const float* __restrict__ bfrag,
                                               float* __restrict__ F, int N, float* wl, const DirectSrc* src = nullptr,
                                               int* claim = nullptr, int tag = 0, int* s_owned = nullptr, float fscale = 1.0f,
                                               Ready ready = Ready()) {
        int lane = threadIdx.x & 63;
        // (single-launch forms inline this function into several roles: an opaque copy of the lane index keeps the compiler from
        // computing the lane geometry once at kernel entry and carrying it through all of them -- VGPRs the register-fed heads need)
        if (SC1) asm volatile("" : "+v"(lane));
        const int col = lane & 15, g = lane >> 4;
        // single-launch pass: the block CLAIMS its work item (one exchange on the item's own word; `tag` = this launch's epoch).
        // The answer is needed only after the weight-staging barrier below, so its round trip hides under the gather.  A block
        // that finds the item already claimed (a waiting consumer executed it: ethcnn_small.hip, "claim or execute") leaves.
        if (claim != nullptr && threadIdx.x == 0)
            *s_owned = (__hip_atomic_exchange(claim, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag);
        // DIRECT (latency path): the pixel gather is requested FIRST, so that its memory round trip overlaps the weight staging
        // below.  (L blocks: all four waves take part in the gather's LDS exchange; waves 1..3 leave after the barriers.)
        uint4 raw[NJ];
        bool active = wave < ntasks;
        if (DIRECT && active) active = load_direct(*src, wave, lane, raw, reinterpret_cast<uint4*>(wl + kTrunkWFrags * 64));
        // bias fragments: requested BEFORE the staging barrier (loads cannot be moved across it by the compiler, and behind it
        // they are one more exposed memory round trip of a block that has nothing else to do yet: single-launch pass)
        f32x4 B1, B2[2], B3[2];
        {
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
        // conv2 / conv3 A-operand fragments of this branch -> LDS, once per block (80 of the 84
        // fragments; every MFMA fetches its A operand with one conflict-free ds_read_b32, which costs
        // the matrix pipe nothing, and frees 80 VGPRs: three waves per SIMD instead of two)
        auto stage = [&]() {
            const float* wf = wfrag + (size_t)BR * kTrunkWFrags * 64;
            for (int i = threadIdx.x; i < kTrunkWFrags * 64; i += 256) wl[i] = wf[i];
            if (RESI)  // (same expression as px_value's callers used per value: the table IS those values)
                for (int i = threadIdx.x; i < TAB; i += 256) wl[kTrunkResiTabAt + i] = px_value<true>(i, POOL * POOL) * (BR == 0 ? 1.0f : SCALE);
            __syncthreads();
        };
        stage();
        if (ready()) stage();  // (block-uniform answer)
        if (claim != nullptr && !*s_owned) return;  // (block-uniform)
        if (!active) return;
        const float* tab = wl + kTrunkResiTabAt;  // (resi only)
        const float* wA2 = wl + 4 * 64 + lane;   // A2[t][s] = wA2[(t * 16 + s) * 64]
        const float* wA3 = wl + 36 * 64 + lane;  // A3[t][s] = wA3[(t * 24 + s) * 64]

        float A1[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) A1[s] = wl[s * 64 + lane];

        // buffer addressing: SGPR resource + SGPR task / slot offset + ONE VGPR lane offset per instruction.
        // The 128-bit STORES keep their whole offset in the VGPR (one v_add each) and soffset = 0: with an SGPR soffset
        // hipcc (ROCm 7.2) places the next VALU write of the store-data registers directly behind the store -- its
        // hazard recognizer assumes a register soffset removes the ">64-bit store data" hazard -- and on gfx950 lanes
        // 12..15 of every row then stored the overwritten value (profiles/r02_fc1_variants.txt, "buffer_store hazard").
        const int lane16 = lane * 16;
        const int lane_off = FAST ? (g >> 1) * FCH + (g & 1) * 512 + col * 16  // plan 2: [chunk][piece][k half][row][8 x 16 bit]
                                  : (col * 4 + g * 64) * 4;  // feature stores: [k/4][16 CTUs][4] -> g, col
        const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(X), 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(F, 0, -1, 0x00020000);
        if (!DIRECT) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                raw[j] = (TRUNK_BUF_LOADS || SC1) ? __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rX, lane16, (wave * NJ + j) * 1024, SC1 ? kAuxSc1 : 0))
                                         : X[((size_t)wave * NJ + j) * 64 + lane];  // (SC1: records written inside this launch, by its PULL role)
        }

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
            // plan 2: this task's 8 chunks inside its pair image (unit position T: 16 S, 4 M, 1 L), rows 16 .. 31 for the odd group
            const int Tpos = (BR == 0) ? (task & 15) : (BR == 1 ? 16 + (task & 3) : 20);
            const int Fb = (grp >> 1) * (kFastChunks * FCH) + Tpos * 8 * FCH + (grp & 1) * 256;
            auto store_pair = [&](const f32x4& qa, const f32x4& qb, int pair) {
                store_pair_f16x2<SC1>(qa, qb, fscale, rF, lane_off + Fb + pair * 2 * FCH);
            };

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
        if (FAST) {                                                                                    \
            if (((q2) & 1) && valid) store_pair(a2[(q2) - 1][0], a2[q2][0], (q2) >> 1);                \
        } else if (valid) {                                                                            \
            const int slot = (2 * by + ((q2) >> 1)) * (2 * NB) + 2 * bx + ((q2) & 1);                  \
            const int dst = Fg + ((OFF2 + slot * 24) >> 2) * 256;                                      \
            if (TRUNK_BUF_STORES) {                                                                    \
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a2[q2][0]), rF, lane_off + dst, 0, SC1 ? kAuxSc1 : 0);  \
                if (g < 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a2[q2][1]), rF, lane_off + dst + 1024, 0, SC1 ? kAuxSc1 : 0); \
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
                decode_q2(raw, q2, x, tab);
                CONV1(q2, c1);
                LEAKY1(c1);
                CONV2(c1, c2);
                FINISH2(q2, c2);
            }
            // the raw registers are dead now: the next task's record is fetched under conv3 (and the
            // other waves' work) at no VGPR cost
            if (task + nwaves < ntasks) {
                if (DIRECT) {  // (L blocks run exactly one task: the cooperative gather has no second round)
                    if (BR != 2) (void)load_direct(*src, task + nwaves, lane, raw, nullptr);
                } else {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        raw[j] = (TRUNK_BUF_LOADS || SC1) ? __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rX, lane16, ((task + nwaves) * NJ + j) * 1024, SC1 ? kAuxSc1 : 0))
                                                 : X[((size_t)(task + nwaves) * NJ + j) * 64 + lane];
                }
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
            f32x4 zq[2];  // (plan 2: the packed quads are also the third register pair of the task)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float hi = __shfl(a2[2 * j + 1][1][r], lane & 31);  // lanes 32..63 <- lanes 0..31 of position 2j+1
                    const float z = (lane < 32) ? a2[2 * j][1][r] : hi;
                    if (FAST) zq[j][r] = z;
                    c3[0] = MFMA16(wA3[(16 + 4 * j + r) * 64], z, c3[0]);
                    c3[1] = MFMA16(wA3[(24 + 16 + 4 * j + r) * 64], z, c3[1]);
                }
            if (FAST) {
                if (valid) {
                    store_pair(zq[0], zq[1], 2);
                    store_pair(lrelu4(c3[0]), lrelu4(c3[1]), 3);
                }
            } else if (valid) {
                const int dst = Fg + ((OFF3 + (by * NB + bx) * 32) >> 2) * 256;
                if (TRUNK_BUF_STORES) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lrelu4(c3[0])), rF, lane_off + dst, 0, SC1 ? kAuxSc1 : 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lrelu4(c3[1])), rF, lane_off + dst + 1024, 0, SC1 ? kAuxSc1 : 0);
                } else {
                    char* p_ = reinterpret_cast<char*>(F) + (size_t)dst + lane_off;
                    *reinterpret_cast<f32x4*>(p_) = lrelu4(c3[0]);
                    *reinterpret_cast<f32x4*>(p_ + 1024) = lrelu4(c3[1]);
                }
            }
        }
    }
};


}  // namespace ethcnn
