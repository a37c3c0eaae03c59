// ethcnn_fc1_tile.h -- device code of the FC1 tile (see ethcnn_dense.hip for the design notes): shared by the FC1 kernels
// (ethcnn_dense.hip) and the single-launch small pass (ethcnn_small.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "ethcnn_kernels.h"

namespace ethcnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
#ifndef MFMA16
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif
// buffer-instruction cache policy (aux) bit of an AGENT-scope access on gfx940+: sc1 (bit 4); sc0 is bit 0, nt bit 1
constexpr int kAuxSc1 = 16;

// ---- k_fc1_p3.  Everything arrives by LDS-DMA (W1 chunk shared by the block, each wave's own
// 1 KiB feature sub-chunks: read back with one conflict-free ds_read_b128), 3 LDS stages,
// prefetch distance 2.
// All VMEM operations of a wave are DMA instructions issued in a fixed number per iteration, so
// the data of chunk kc+1 is known to have landed when `vmcnt` has drained down to this
// iteration's own issue count: a COUNTED s_waitcnt, never vmcnt(0), and a raw s_barrier (a
// __syncthreads() would drain the queue, ROCm 7.2).  WAR: stage (kc+3)%3 == kc%3 is refilled in
// iteration kc+1, after every wave has passed the barrier that ends iteration kc.
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most `keep` (0..MAXK) groups of ISSUE DMA instructions are still in flight
template <int ISSUE, int MAXK>
__device__ __forceinline__ void vm_wait_groups(int keep) {
    if constexpr (MAXK == 0) {
        vm_wait<0>();
    } else {
        if (keep >= MAXK) vm_wait<MAXK * ISSUE>();
        else vm_wait_groups<ISSUE, MAXK - 1>(keep);
    }
}

// NST = LDS stages (prefetch distance NST - 1).  3 everywhere: deeper rings (4..6 stages, distance up to 5) were measured on
// the short-range shapes, which run one block per CU or less, and change nothing (profiles/r02_fc1_variants.txt section 5):
// those launches are bound by the per-chunk barrier + ds_read -> MFMA latency of a lone wave per SIMD, not by DMA latency.
template <int MS, int NS, int WM, int NSUB, int NST>
struct Fc1Shape {
    static constexpr int B_FLOATS = 16 * NSUB * 16 * NS;
    static constexpr int STAGE = B_FLOATS + WM * NSUB * MS * 256;
    static constexpr int LDS_FLOATS = NST * STAGE;
};

// block -> (M tile, column block).  Spread: column block = b % NSPLIT, an XCD (b % 8) streams one slice of W1 but every M
// tile's features are fetched by NSPLIT different XCDs.  GROUP: the NSPLIT column blocks of an M tile are consecutive slots
// of ONE XCD, so its features come from HBM once and are shared through that XCD's L2 (W1 is then swept whole per XCD).
template <int NSPLIT, bool GROUP>
__device__ __forceinline__ void fc1_block_to_tile(unsigned bid, int& mt, int& nb) {
    if (GROUP) {
        const int xcd = bid & 7, slot = bid >> 3;
        nb = slot % NSPLIT;
        mt = (slot / NSPLIT) * 8 + xcd;
    } else {
        nb = (int)(bid % NSPLIT);
        mt = (int)(bid / NSPLIT);
    }
}

// one output tile: M tile `mt` (rows mt * BM ..), column block `nb`; smem: >= Fc1Shape<...>::LDS_FLOATS floats.
// COHERENT: the h1 stores are agent-scope (sc1: written through this XCD's L2), for a consumer that runs in the SAME launch
// on another XCD (the heads blocks of the single-launch small pass, ethcnn_small.hip); results are identical.
// A_SC1: the features were written earlier in the SAME launch (single-launch small pass): fetched with agent-scope loads.
template <int MS, int NS, int WM, int NSUB, int NST, bool COHERENT = false, bool A_SC1 = false>
__device__ __forceinline__ void fc1_tile_at(float* __restrict__ smem, const float* __restrict__ feat, const float* __restrict__ Wimg,
                                            const float* __restrict__ bias, float* __restrict__ out, int M, const int mt, const int nb) {
    constexpr int BK = 16 * NSUB, BN = 16 * NS, BM = 16 * MS * WM;
    constexpr int NK = kNFeat / BK;
    constexpr int B_FLOATS = BK * BN;
    constexpr bool COLSWZ = (BN % 32 == 0);
    constexpr int B_INST = B_FLOATS / 256;
    constexpr int B_PER = (B_INST + WM - 1) / WM;   // every wave issues exactly B_PER (tail duplicates the last piece)
    constexpr int A_PER = NSUB * MS;                // 1 KiB feature pieces per wave per chunk
    constexpr int ISSUE = B_PER + A_PER;            // VMEM ops per wave per iteration
    constexpr int A_FLOATS = WM * A_PER * 256;
    constexpr int STAGE = B_FLOATS + A_FLOATS;
    constexpr int DIST = NST - 1;
    static_assert(NK % NST == 0 && NK >= NST, "K chunks must come in whole rounds of the stage ring");
    static_assert(DIST * ISSUE <= 63, "vmcnt is a 6-bit counter");
    static_assert(STAGE == Fc1Shape<MS, NS, WM, NSUB, NST>::STAGE, "LDS budget of the shape");

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 15, g = lane >> 4;
    const int m0 = mt * BM + wv * 16 * MS;
    const int n0 = nb * BN;

    f32x4 acc[MS][NS];
#pragma unroll
    for (int i = 0; i < MS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ngroups = (M + 15) >> 4;
    // DMA addressing: every source is (wave-uniform 64-bit base in SGPRs) + (lane * 16 B in ONE VGPR), the
    // `global_load_lds_dwordx4 vOffset, s[base:base+1]` form.  The per-chunk advance is scalar arithmetic (free beside
    // MFMAs); the 64-bit per-lane addresses of the `vAddr, off` form cost two v_lshl_add_u64 per DMA and two VGPR
    // address reads -- VALU / VGPR-port time that comes straight out of the matrix pipe on gfx950
    // (profiles/r01_ubench_gfx950_issue_costs.txt).  LDS destinations are wave-uniform too (M0).
    const unsigned wvu = __builtin_amdgcn_readfirstlane(wv);
    const unsigned lane16 = (unsigned)lane * 16u;
    const int mg0 = __builtin_amdgcn_readfirstlane(m0 >> 4);
    const float* a_base[MS];  // this wave's group images [k/4][16][4]: 1 KiB per 16-k sub-chunk
#pragma unroll
    for (int i = 0; i < MS; ++i) a_base[i] = feat + (size_t)min(mg0 + i, ngroups - 1) * (kNFeat / 4) * 64;
    const float* b_base[B_PER];
    unsigned b_dst[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const unsigned piece = min(wvu + i * WM, (unsigned)(B_INST - 1));
        b_base[i] = Wimg + (size_t)nb * NK * B_FLOATS + (size_t)piece * 256;
        b_dst[i] = piece * 1024u;
    }
    int bcol[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) bcol[j] = (j * 16 + col) ^ (COLSWZ ? ((g & 1) << 4) : 0);
    int brow[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) brow[e] = (COLSWZ ? e : (e ^ (g & 1))) * BN;
    float* a_lds = smem + B_FLOATS + wv * A_PER * 256;  // + stage * STAGE

    // LDS-DMA through inline asm: hipcc does not model it, so it neither drains it at an LDS read
    // (as it does for the builtin: a vmcnt(0) before the first ds_read) nor counts it -- the
    // counted waits below are the only ordering.  M0 = wave-uniform LDS byte address, written in
    // the same statement that uses it (guide 5.7).
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)smem);
    const unsigned a_dst0 = lds_base + 4u * (B_FLOATS + wvu * A_PER * 256);
#define P3_DMA(sbase, lds_byte_addr)                                                                   \
    {                                                                                                  \
        unsigned keep_;                                                                                \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(lane16), "s"(sbase), "s"(lds_byte_addr) : "memory");          \
    }
#define P3_DMA_SC1(sbase, lds_byte_addr)                                                               \
    {                                                                                                  \
        unsigned keep_;                                                                                \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(lane16), "s"(sbase), "s"(lds_byte_addr) : "memory");          \
    }
#define P3_ISSUE(kc, st)                                                                               \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < B_PER; ++i)                                              \
            P3_DMA(b_base[i] + (size_t)(kc) * B_FLOATS, lds_base + 4u * (st) * STAGE + b_dst[i]);      \
        _Pragma("unroll") for (int u = 0; u < NSUB; ++u)                                               \
            _Pragma("unroll") for (int i = 0; i < MS; ++i) {                                           \
                if (A_SC1) P3_DMA_SC1(a_base[i] + ((size_t)(kc) * NSUB + u) * 256,                     \
                                      a_dst0 + 4u * ((st) * STAGE + (u * MS + i) * 256))               \
                else P3_DMA(a_base[i] + ((size_t)(kc) * NSUB + u) * 256,                               \
                            a_dst0 + 4u * ((st) * STAGE + (u * MS + i) * 256))                         \
            }                                                                                          \
    }
#define P3_COMPUTE(st)                                                                                 \
    {                                                                                                  \
        _Pragma("unroll") for (int u = 0; u < NSUB; ++u) {                                             \
            float4 av[MS];                                                                             \
            _Pragma("unroll") for (int i = 0; i < MS; ++i)                                             \
                av[i] = *reinterpret_cast<const float4*>(a_lds + (st) * STAGE + (u * MS + i) * 256 + lane * 4); \
            const float* bs = smem + (st) * STAGE + (16 * u + 4 * g) * BN;                             \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                            \
                float b[NS];                                                                           \
                _Pragma("unroll") for (int j = 0; j < NS; ++j) b[j] = bs[brow[e] + bcol[j]];           \
                _Pragma("unroll") for (int i = 0; i < MS; ++i) {                                       \
                    const float a = (e == 0) ? av[i].x : (e == 1) ? av[i].y : (e == 2) ? av[i].z : av[i].w; \
                    _Pragma("unroll") for (int j = 0; j < NS; ++j) acc[i][j] = MFMA16(a, b[j], acc[i][j]); \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
    }
    // iteration kc: refill the stage consumed in iteration kc-1 with chunk kc+DIST, compute chunk kc, then make sure chunk
    // kc+1 has landed: of the chunks issued so far only the newest min(DIST-1, NK-2-kc) may still be in flight
#define P3_STEP(kc, st)                                                                                \
    {                                                                                                  \
        if ((kc) + DIST < NK) { P3_ISSUE((kc) + DIST, ((st) + DIST) % NST); }                          \
        P3_COMPUTE(st);                                                                                \
        vm_wait_groups<ISSUE, DIST - 1>(NK - 2 - (kc));                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
        __builtin_amdgcn_s_barrier();                                                                  \
    }

    // wins the issue arbitration against the co-resident wave of the next pass's tile stage (priority 0), which then fills
    // the gaps this kernel leaves instead of taking slots from it: FC1 inside the timed region 1.712 -> 1.686 ms, the step
    // +0.2 % (the tile stage, now slower, becomes what the next trunk waits for); priority 3 measures the same
    __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int c0 = 0; c0 < DIST; ++c0) { P3_ISSUE(c0, c0); }
    vm_wait<(DIST - 1) * ISSUE>();  // chunk 0 landed (the younger ones may still be in flight)
    __builtin_amdgcn_s_barrier();
    for (int kc = 0; kc < NK; kc += NST) {
#pragma unroll
        for (int st = 0; st < NST; ++st) { P3_STEP(kc + st, st); }
    }
#undef P3_DMA
#undef P3_DMA_SC1
#undef P3_ISSUE
#undef P3_COMPUTE
#undef P3_STEP

    // epilogue: bias + leaky-ReLU, one buffer_store per value: SGPR resource + uniform column offset (soffset) + one
    // VGPR offset holding the ROW part.  The hardware range check covers voffset only (the SGPR offset is not part of
    // it), and the resource ends at row M: rows of a ragged last tile are dropped by the check, no exec masking.
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(out, 0, M * kNVec * 4, 0x00020000);
    const int lane_out = ((m0 + 4 * g) * kNVec + col) * 4;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const float bv = bias[n0 + j * 16 + col];
#pragma unroll
        for (int i = 0; i < MS; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float h = acc[i][j][r] + bv;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(0.2f * h, h)), rO,
                                                      lane_out + (i * 16 + r) * kNVec * 4, (n0 + j * 16) * 4, COHERENT ? kAuxSc1 : 0);
            }
    }
}

// the tile of block `bid` in a grid of this shape over M rows (blocks beyond the last tile leave before any barrier)
template <int MS, int NS, int WM, int NSUB, bool GROUP, int NST>
__device__ __forceinline__ void fc1_tile(float* __restrict__ smem, const float* __restrict__ feat, const float* __restrict__ Wimg,
                                         const float* __restrict__ bias, float* __restrict__ out, int M, const unsigned bid) {
    int mt, nb;
    fc1_block_to_tile<kNVec / (16 * NS), GROUP>(bid, mt, nb);
    if (mt * (16 * MS * WM) >= M) return;
    fc1_tile_at<MS, NS, WM, NSUB, NST>(smem, feat, Wimg, bias, out, M, mt, nb);
}


}  // namespace ethcnn
