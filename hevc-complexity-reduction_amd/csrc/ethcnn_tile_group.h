// ethcnn_tile_group.h -- the CTU-load stage's work item: ONE group of 16 consecutive CTUs -> the trunk's pixel records
// (layouts and the 16-row slab scheme: ethcnn_tile.hip).  Shared by the bulk tile kernel (k0_tile_slab, HBM -> HBM) and by the
// single-launch small pass's PULL role (ethcnn_small.hip), which reads the picture straight from page-locked HOST memory --
// one coalesced pass over PCIe, 16 bytes per lane -- while the rest of the launch is already working.
#pragma once
#include <hip/hip_runtime.h>

#include "ethcnn_fc1_tile.h"  // kAuxSc1
#include "ethcnn_kernels.h"

namespace ethcnn {

constexpr int kRowPitch = 17;                    // dwords per CTU row in LDS (16 + 1 pad)

// One block handles a GROUP of 16 consecutive CTUs, staged one 16-row slab at a time (17.5 KB of LDS; the r01 kernel staged
// the whole 64 rows: 69.7 KB, which no other kernel could sit beside).  Every output uint4 depends on exactly one slab:
//   XS  unit row uy = s                                   4 units x 4 j x 64 lanes   per slab
//   XM  unit row uy = s >> 1, j in [4 (s & 1), +4)        2 units x 4 j x 64 lanes
//   XL  j = 4 (s >> 1) + 2 m + (s & 1), m = 0, 1          2 j x 64 lanes
// so a block walks s = 0..3 with the next slab's pixels already in flight in registers.  Small enough to be co-resident
// with three FC1 blocks per CU (138 KB + 17.5 KB <= 160 KB): this is what lets the CTU-load stage of pass i+1 run UNDER
// the MFMA-bound FC1 of pass i (csrc/ethcnn_pass.cpp run_pass, profiles/r02_overlap_trace.txt).
constexpr int kSlabCtuPitch = 16 * kRowPitch + 1;  // 273 dwords: 17 c mod 32 puts the 16 CTUs on 16 different banks

// streaming accesses: the records are read by the NEXT pass's trunk, long after; kept out of the L2 lines FC1 (running
// beside this stage) shares between the column blocks of an M tile
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(uint4* p, uint4 v) {
    __builtin_nontemporal_store((u32x4_t){v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_t*>(p));
}
__device__ __forceinline__ uint4 nt_load(const uint4* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// WAIT (streamed input, ethcnn_ldp_step_begin): the picture lies in page-locked HOST memory that the caller is STILL FILLING when the
// kernel starts -- an encoder-side reader copying resi.yuv out of the page cache, CTU row by CTU row.  rows[cy] == seq says "the 64
// luma rows of CTU row cy of picture `seq` are in the buffer" (ethcnn_rows_ready: a release store by the filling thread; the
// device's reads of host memory snoop the CPU caches and x86 makes the stores visible in order, so data read AFTER the flag is the
// new picture's).  One thread per block polls the flags of its group's one or two CTU rows with system-scope loads.  A caller
// that dies between begin and end must not hang the GPU: after ~1 s of wall clock the block gives up, stores `seq` to *gave_up (the
// call then fails in ethcnn_ldp_step_end) and goes on with whatever is in the buffer.
struct TileWait {
    const unsigned* rows;  // page-locked host memory, one word per CTU row; null: no waiting (WAIT == false)
    unsigned seq;
    unsigned* gave_up;     // page-locked host memory
};

// every thread of the block; returns behind a barrier
__device__ __forceinline__ void tile_wait_rows(const TileWait& tw, long ctu0, int grp, int n_total, int nctu, int cw) {
    if (threadIdx.x == 0) {
        const int r0 = (int)((ctu0 + grp * 16) % nctu), r1 = (int)((ctu0 + min(grp * 16 + 15, n_total - 1)) % nctu);
        unsigned long long t0;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int cy = r0 / cw; cy <= r1 / cw; ++cy)
            while (__hip_atomic_load(tw.rows + cy, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != tw.seq) {
                __builtin_amdgcn_s_sleep(16);
                unsigned long long t;
                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
                if (t - t0 > 100000000ull) {  // 1 s at 100 MHz
                    __hip_atomic_store(tw.gave_up, tw.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
    }
    __syncthreads();
}

// ---- the pieces of a group's work, shared by tile_group (below) and by the plan-3 trunk that consumes the S records straight out
// of LDS (ethcnn_trunk_fast.hip, k1_trunk_f16_fold).  `tile`: 16 * kSlabCtuPitch dwords of LDS = one 16-row slab of the group.
#define ETHCNN_PXS(tile, c, Y, Xd) (tile)[(c) * kSlabCtuPitch + (Y) * kRowPitch + (Xd)]

// loader role of a 256-thread block: lane -> (CTU c, 16-B segment), wave -> 4 of the slab's 16 rows: one wave instruction covers a
// whole 1 KiB run of a frame row when the 16 CTUs are horizontally adjacent
struct SlabLoader {
    const uint8_t* lbase;  // first pixel of this thread's segment in row 0 of its CTU; null = all zero (beyond the pass / the frame)
    int ly0, lx, lc, lseg, lrow0;
    __device__ __forceinline__ void init(int t, const uint8_t* __restrict__ luma, int width, long frame_stride, int cw, int nctu, long ctu0,
                                         int n_total, int n0) {
        lc = (t >> 2) & 15; lseg = t & 3; lrow0 = (t >> 6) * 4;
        lbase = nullptr; ly0 = 0; lx = 0;
        const int n = n0 + lc;
        if (n < n_total) {
            const long gn = ctu0 + n;
            const long f = gn / nctu;
            const int rr = (int)(gn - f * nctu);
            const int cy = rr / cw, cx = rr - cy * cw;
            ly0 = cy * 64;
            lx = cx * 64 + lseg * 16;
            if (lx < width) lbase = luma + f * frame_stride + lx;
        }
    }
    // the same with the pass's first CTU given as (frame f0, CTU r0 inside it): 32-bit arithmetic only (the folded trunk re-derives
    // the geometry for every slab item it walks)
    __device__ __forceinline__ void init32(int t, const uint8_t* __restrict__ luma, int width, long frame_stride, int cw, int nctu, int f0, int r0,
                                           int n_total, int n0) {
        lc = (t >> 2) & 15; lseg = t & 3; lrow0 = (t >> 6) * 4;
        lbase = nullptr; ly0 = 0; lx = 0;
        const int n = n0 + lc;
        if (n < n_total) {
            const unsigned q = (unsigned)(r0 + n), df = q / (unsigned)nctu, rr = q - df * (unsigned)nctu;
            const unsigned cy = rr / (unsigned)cw, cx = rr - cy * (unsigned)cw;
            ly0 = (int)cy * 64;
            lx = (int)cx * 64 + lseg * 16;
            if (lx < width) lbase = luma + (long)(f0 + (int)df) * frame_stride + lx;
        }
    }
    // rows lrow0 .. lrow0 + 3 of slab s (zero beyond the frame: video_to_cu_depth.py:54-57)
    template <bool FAST>
    __device__ __forceinline__ void load(int s, uint4 (&v)[4], int width, int height, long pitch) const {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int y = ly0 + 16 * s + lrow0 + k;
            uint4 r = make_uint4(0u, 0u, 0u, 0u);
            if (lbase != nullptr && y < height) {
                const uint8_t* p = lbase + (long)y * pitch;
                if (FAST) {
                    r = nt_load(reinterpret_cast<const uint4*>(p));
                } else {
                    uint32_t w4[4] = {0u, 0u, 0u, 0u};
                    const int lim = min(16, width - lx);
                    for (int i = 0; i < lim; ++i) w4[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3));
                    r = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                }
            }
            v[k] = r;
        }
    }
    __device__ __forceinline__ void to_lds(uint32_t* tile, const uint4 (&v)[4]) const {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t* dst = &ETHCNN_PXS(tile, lc, lrow0 + k, lseg * 4);
            dst[0] = v[k].x; dst[1] = v[k].y; dst[2] = v[k].z; dst[3] = v[k].w;
        }
    }
};

// XS record j of S unit (uy = the slab, ux) for lane (c, g): the 4 pixels of row g of the patches (q2 = j, q1 = 0..3)
__device__ __forceinline__ uint4 slab_xs_record(const uint32_t* tile, int c, int g, int j, int ux) {
    uint32_t d[4];
#pragma unroll
    for (int q1 = 0; q1 < 4; ++q1) d[q1] = ETHCNN_PXS(tile, c, 8 * (j >> 1) + 4 * (q1 >> 1) + g, 4 * ux + 2 * (j & 1) + (q1 & 1));
    return make_uint4(d[0], d[1], d[2], d[3]);
}
// XM record j (j = 4 (s & 1) + jj for slab s) of M unit column ux: two patch rows of exact 2x2 sums
__device__ __forceinline__ uint4 slab_xm_record(const uint32_t* tile, int c, int g, int j, int ux) {
    uint32_t out[4];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int dd = 2 * j + hh, q2 = dd >> 2, q1 = dd & 3;
        const int Yl = 8 * (q1 >> 1) + 2 * g;                    // raw row of the pooled row inside the slab
        const int Xp = 16 * ux + 8 * (q2 & 1) + 4 * (q1 & 1);    // pooled col of px 0
        const uint32_t a0 = ETHCNN_PXS(tile, c, Yl, Xp >> 1), a1 = ETHCNN_PXS(tile, c, Yl, (Xp >> 1) + 1);
        const uint32_t b0 = ETHCNN_PXS(tile, c, Yl + 1, Xp >> 1), b1 = ETHCNN_PXS(tile, c, Yl + 1, (Xp >> 1) + 1);
        // exact 2x2 byte sums as masked v_dot4_u32_u8 pairs: 5 VALU per output dword instead of ~15 shifts / masks /
        // adds -- the stage runs beside FC1, where every VALU instruction costs matrix-pipe issue time
        const uint32_t s0 = __builtin_amdgcn_udot4(a0, 0x00000101u, __builtin_amdgcn_udot4(b0, 0x00000101u, 0u, false), false);
        const uint32_t s1 = __builtin_amdgcn_udot4(a0, 0x01010000u, __builtin_amdgcn_udot4(b0, 0x01010000u, 0u, false), false);
        const uint32_t s2 = __builtin_amdgcn_udot4(a1, 0x00000101u, __builtin_amdgcn_udot4(b1, 0x00000101u, 0u, false), false);
        const uint32_t s3 = __builtin_amdgcn_udot4(a1, 0x01010000u, __builtin_amdgcn_udot4(b1, 0x01010000u, 0u, false), false);
        out[2 * hh] = s0 | (s1 << 16);
        out[2 * hh + 1] = s2 | (s3 << 16);
    }
    return make_uint4(out[0], out[1], out[2], out[3]);
}
// XL record j (j = 4 (s >> 1) + 2 m + (s & 1), m = 0, 1, for slab s): two patch rows of exact 4x4 sums
__device__ __forceinline__ uint4 slab_xl_record(const uint32_t* tile, int c, int g, int j) {
    uint32_t out[4];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int dd = 2 * j + hh, q2 = dd >> 2, q1 = dd & 3;
        const int Xp = 8 * (q2 & 1) + 4 * (q1 & 1);  // pooled col == dword col
        uint32_t sacc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t acc = 0;
#pragma unroll
            for (int ry = 0; ry < 4; ++ry) acc = __builtin_amdgcn_udot4(ETHCNN_PXS(tile, c, 4 * g + ry, Xp + i), 0x01010101u, acc, false);
            sacc[i] = acc;
        }
        out[2 * hh] = sacc[0] | (sacc[1] << 16);
        out[2 * hh + 1] = sacc[2] | (sacc[3] << 16);
    }
    return make_uint4(out[0], out[1], out[2], out[3]);
}

// One group.  `tile`: 16 * kSlabCtuPitch dwords of LDS.  FAST: rows are 16-byte aligned (width, pitch, frame stride, base).
// SC1: the records are consumed INSIDE this launch (agent-scope stores; the caller completes them -- s_waitcnt vmcnt(0) --
// before it signals); otherwise streaming stores for the next launch.  ALL: the four slabs' loads are requested at once
// (64 KiB in flight per block: the PCIe round trip is paid once per group), otherwise one slab ahead (the bulk kernel beside
// FC1: registers and LDS kept small).  Every thread of the 256-thread block; ends with a barrier (the LDS tile is free).
template <bool FAST, bool SC1, bool ALL>
__device__ __forceinline__ void tile_group(uint32_t* tile, const uint8_t* __restrict__ luma, int width, int height, long pitch, long frame_stride,
                                           int cw, int nctu, long ctu0, int n_total, int grp, uint4* __restrict__ XS, uint4* __restrict__ XM,
                                           uint4* __restrict__ XL) {
    int t = threadIdx.x;
    // (inlined into several roles of the single-launch pass: an opaque copy of the thread index keeps the compiler from computing the
    // lane geometry below ONCE at kernel entry and carrying it through every role -- 8 VGPRs spilled in the register-fed heads)
    if (ALL) asm volatile("" : "+v"(t));
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(XS, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(XM, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc(XL, 0, -1, 0x00020000);
    auto put_rec = [&](uint4* base, const __amdgpu_buffer_rsrc_t& r, size_t idx, uint4 v) {
        if (SC1) __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){v.x, v.y, v.z, v.w}, r, (int)(idx * 16), 0, kAuxSc1);
        else nt_store(base + idx, v);
    };
    SlabLoader L;
    L.init(t, luma, width, frame_stride, cw, nctu, ctu0, n_total, grp * 16);
    uint4 pre[ALL ? 4 : 1][4];
    L.template load<FAST>(0, pre[0], width, height, pitch);
    if (ALL) {
#pragma unroll
        for (int s = 1; s < 4; ++s) L.template load<FAST>(s, pre[ALL ? s : 0], width, height, pitch);
    }
    constexpr int kSlabUnroll = ALL ? 4 : 1;  // (ALL: pre[s] must be a compile-time register choice)
#pragma unroll kSlabUnroll
    for (int s = 0; s < 4; ++s) {
        L.to_lds(tile, pre[ALL ? s : 0]);
        __syncthreads();
        if (!ALL && s < 3) L.template load<FAST>(s + 1, pre[0], width, height, pitch);  // in flight while this slab is turned into records

        // ---- XS: unit row uy = s: 4 units x 4 j x 64 lanes, 4 per thread; output index e + 1024 s (u = 4 s + ux)
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            const int e = t + 256 * rep;
            const int lane = e & 63, j = (e >> 6) & 3, ux = e >> 8;
            put_rec(XS, rS, (size_t)grp * 4096 + 1024 * s + e, slab_xs_record(tile, lane & 15, lane >> 4, j, ux));
        }
        // ---- XM: units (uy = s >> 1, ux = 0, 1), j = 4 (s & 1) + jj: 2 x 4 x 64, 2 per thread
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int e = t + 256 * rep;
            const int lane = e & 63, jj = (e >> 6) & 3, ux = e >> 8;
            const int j = 4 * (s & 1) + jj, unit = 2 * (s >> 1) + ux;
            put_rec(XM, rM, (size_t)grp * 2048 + 512 * unit + 64 * j + lane, slab_xm_record(tile, lane & 15, lane >> 4, j, ux));
        }
        // ---- XL: j = 4 (s >> 1) + 2 m + (s & 1): 2 x 64, threads 0..127
        if (t < 128) {
            const int lane = t & 63, m = t >> 6;
            const int j = 4 * (s >> 1) + 2 * m + (s & 1);
            put_rec(XL, rL, (size_t)grp * 512 + 64 * j + lane, slab_xl_record(tile, lane & 15, lane >> 4, j));
        }
        __syncthreads();  // the slab is consumed: the next one may overwrite it
    }
}

}  // namespace ethcnn
