// ethcnn_fc1_regs.h -- the register-fed FC1 tile (no LDS): device code shared by the single-launch small pass (ethcnn_small.hip)
// and the short-row-range FC1 kernels (ethcnn_dense.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "ethcnn_fc1_tile.h"
#include "ethcnn_kernels.h"

namespace ethcnn {

// ---- FC1 tile, REGISTER-FED (the single-launch pass and the short row ranges of ethcnn_dense.hip): 64 CTUs x 16 NS columns, one wave per 16 CTUs, NS accumulators per wave
// = NS dependent chains of 672 MFMAs (the canonical order: sub-chunk u ascending, inside it e = 0..3 -- exactly fc1_tile_at's).
// One picture's FC1 blocks are alone on their SIMDs, so their time is the chain's: 45 cycles per link (scripts/ubench/
// chain_probe.hip), 12.9 us for 672 links -- IF nothing else is exposed.  The LDS-staged tile (fc1_tile_at, built for
// throughput) adds a barrier + an LDS round trip per K chunk and cannot look further ahead than its ring (17 us measured,
// whatever the ring depth).  Here both operands of a sub-chunk are ONE dwordx4 load per lane each, straight from memory
// into a ring of D register slots (features: the trunk's [k/4][16][4] group image, agent-scope; weights: the same order per
// 16-column tile, DeviceWeights::fc1_lane16, shared by the block's four waves through the CU's vector cache): no LDS, no
// barrier, no address arithmetic, look-ahead D sub-chunks = D x 4 NS MFMAs.
// COHERENT (the single-launch pass): features were written and h1 is read by other blocks of the SAME launch -> agent-scope
// (sc1) loads / stores; separate launches use ordinary ones.  Results are identical.
template <int NS, int D, bool COHERENT>
__device__ __forceinline__ void fc1_tile_regs(const float* __restrict__ feat, const float* __restrict__ wlane, const float* __restrict__ bias,
                                              float* __restrict__ out, int M, const int mt, const int nb) {
    constexpr int NU = kNFeat / 16;  // 168 sub-chunks
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = lane & 15, g = lane >> 4;
    const int m0 = mt * 64 + wv * 16, n0 = nb * 16 * NS;
    const int grp = min(m0 >> 4, ((M + 15) >> 4) - 1);  // (a ragged tile's idle waves recompute the last group; never stored)
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(feat) + (size_t)grp * (kNFeat / 4) * 64, 0, kNFeat * 16 * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rB[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j)
        rB[j] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wlane) + (size_t)(nb * NS + j) * NU * 256, 0, NU * 1024, 0x00020000);
    const int voff = lane * 16;
    float bv[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) bv[j] = bias[n0 + j * 16 + col];
    f32x4 ra[D], rb[D][NS];
#pragma unroll
    for (int u = 0; u < D; ++u) {
        ra[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, voff, u * 1024, COHERENT ? kAuxSc1 : 0));
#pragma unroll
        for (int j = 0; j < NS; ++j) rb[u][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB[j], voff, u * 1024, 0));
    }
    f32x4 acc[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int slot = u % D;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < NS; ++j) acc[j] = MFMA16(ra[slot][e], rb[slot][j][e], acc[j]);
        if (u + D < NU) {
            ra[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, voff, (u + D) * 1024, COHERENT ? kAuxSc1 : 0));
#pragma unroll
            for (int j = 0; j < NS; ++j)
                rb[slot][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB[j], voff, (u + D) * 1024, 0));
        }
        // the order above IS the schedule: left alone, the machine scheduler sinks every load to just before its use (fewest
        // live registers) and the look-ahead is gone
        __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue as fc1_tile_at's: bias + leaky-ReLU, agent-scope stores, rows beyond M dropped by the range check
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(out, 0, M * kNVec * 4, 0x00020000);
    const int lane_out = ((m0 + 4 * g) * kNVec + col) * 4;
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float h = acc[j][r] + bv[j];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(0.2f * h, h)), rO, lane_out + r * kNVec * 4, (n0 + j * 16) * 4, COHERENT ? kAuxSc1 : 0);
        }
}

}  // namespace ethcnn
