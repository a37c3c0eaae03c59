// ethcnn_tile.hip -- k0: the CTU-load stage.  Luma frames -> zero-padded 64x64 CTUs in raster
// order (get_Y_for_one_frame + the tiling loop, video_to_cu_depth.py:46-59,88-106) with the
// integer 2x2 / 4x4 pooled sums of aver_pool (net_CNN.py:62-63,126,132), written in the lane
// order the trunk (k1) consumes.  HBM-bound: 4096 B/CTU in, 6656 B/CTU out.
//
// One block = one GROUP of 16 consecutive CTUs (global raster index over the frame sequence), staged through LDS one
// 16-row slab at a time.  Loads: 16-B per lane, a wave instruction covers a whole 1 KiB run of a frame row when the 16 CTUs
// are horizontally adjacent.  Every output record is a full, linear 1 KiB per wave instruction (lane = c + 16 g,
// c = CTU in the group, g = MFMA k-group):
//   XS[group*16 + u][j][lane]  uint4 = 4 dwords q1 = 0..3: the 4 pixels of row g of patch
//                       (q2 = j, q1) of S unit u.  Y = 16uy + 8(q2>>1) + 4(q1>>1) + g,
//                       X = 16ux + 8(q2&1) + 4(q1&1) + 0..3.
//   XM[group*4 + unit][j][lane] uint4 = patch rows d = 2j, 2j+1 (d = 4 q2 + q1) of M unit (2x2),
//                       each 4 x u16 exact sums of 2x2 raw pixels.
//   XL[group][j][lane]  same with 4x4 sums.
#include <hip/hip_runtime.h>

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
// the MFMA-bound FC1 of pass i (csrc/ethcnn_api.cpp run_pass, profiles/r02_overlap_trace.txt).
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

template <bool FAST>
__global__ __launch_bounds__(256) void k0_tile_slab(const uint8_t* __restrict__ luma, int width, int height, long pitch,
                                                    long frame_stride, int cw, int nctu, long ctu0, int n_total,
                                                    uint4* __restrict__ XS, uint4* __restrict__ XM,
                                                    uint4* __restrict__ XL, int* __restrict__ gate_flags, int n_flags) {
    __shared__ uint32_t tile[16 * kSlabCtuPitch];
    const int t = threadIdx.x;
    if (blockIdx.x == 0)
        for (int i = t; i < n_flags; i += 256) gate_flags[i] = 0;
    // persistent form: with fewer blocks than groups (one per CU when the stage runs beside FC1, so that its LDS never keeps
    // FC1's third block off a CU) a block walks several groups
    const int ngroups = (n_total + 15) >> 4;
#pragma unroll 1
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int n0 = grp * 16;

    // loader role: lane -> (CTU c, 16-B segment), wave -> 4 of the slab's 16 rows: one wave instruction covers a whole
    // 1 KiB run of a frame row when the 16 CTUs are horizontally adjacent
    const int lc = (t >> 2) & 15, lseg = t & 3, lrow0 = (t >> 6) * 4;
    const uint8_t* lbase = nullptr;  // first pixel of this thread's segment in row 0 of its CTU; null = all zero
    int ly0 = 0, lx = 0;
    {
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
    auto load_slab = [&](int s, uint4 (&v)[4]) {
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
    };
#define PXS(c, Y, Xd) tile[(c) * kSlabCtuPitch + (Y) * kRowPitch + (Xd)]
    uint4 pre[4];
    load_slab(0, pre);
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t* dst = &PXS(lc, lrow0 + k, lseg * 4);
            dst[0] = pre[k].x; dst[1] = pre[k].y; dst[2] = pre[k].z; dst[3] = pre[k].w;
        }
        __syncthreads();
        if (s < 3) load_slab(s + 1, pre);  // in flight while this slab is turned into records

        // ---- XS: unit row uy = s: 4 units x 4 j x 64 lanes, 4 per thread; output index e + 1024 s (u = 4 s + ux)
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            const int e = t + 256 * rep;
            const int lane = e & 63, j = (e >> 6) & 3, ux = e >> 8;
            const int c = lane & 15, g = lane >> 4;
            uint32_t d[4];
#pragma unroll
            for (int q1 = 0; q1 < 4; ++q1) d[q1] = PXS(c, 8 * (j >> 1) + 4 * (q1 >> 1) + g, 4 * ux + 2 * (j & 1) + (q1 & 1));
            nt_store(&XS[(size_t)grp * 4096 + 1024 * s + e], make_uint4(d[0], d[1], d[2], d[3]));
        }
        // ---- XM: units (uy = s >> 1, ux = 0, 1), j = 4 (s & 1) + jj: 2 x 4 x 64, 2 per thread
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int e = t + 256 * rep;
            const int lane = e & 63, jj = (e >> 6) & 3, ux = e >> 8;
            const int c = lane & 15, g = lane >> 4, j = 4 * (s & 1) + jj, unit = 2 * (s >> 1) + ux;
            uint32_t out[4];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int dd = 2 * j + hh, q2 = dd >> 2, q1 = dd & 3;
                const int Yl = 8 * (q1 >> 1) + 2 * g;                    // raw row of the pooled row inside the slab
                const int Xp = 16 * ux + 8 * (q2 & 1) + 4 * (q1 & 1);    // pooled col of px 0
                const uint32_t a0 = PXS(c, Yl, Xp >> 1), a1 = PXS(c, Yl, (Xp >> 1) + 1);
                const uint32_t b0 = PXS(c, Yl + 1, Xp >> 1), b1 = PXS(c, Yl + 1, (Xp >> 1) + 1);
                // exact 2x2 byte sums as masked v_dot4_u32_u8 pairs: 5 VALU per output dword instead of ~15 shifts / masks /
                // adds -- the stage runs beside FC1, where every VALU instruction costs matrix-pipe issue time
                const uint32_t s0 = __builtin_amdgcn_udot4(a0, 0x00000101u, __builtin_amdgcn_udot4(b0, 0x00000101u, 0u, false), false);
                const uint32_t s1 = __builtin_amdgcn_udot4(a0, 0x01010000u, __builtin_amdgcn_udot4(b0, 0x01010000u, 0u, false), false);
                const uint32_t s2 = __builtin_amdgcn_udot4(a1, 0x00000101u, __builtin_amdgcn_udot4(b1, 0x00000101u, 0u, false), false);
                const uint32_t s3 = __builtin_amdgcn_udot4(a1, 0x01010000u, __builtin_amdgcn_udot4(b1, 0x01010000u, 0u, false), false);
                out[2 * hh] = s0 | (s1 << 16);
                out[2 * hh + 1] = s2 | (s3 << 16);
            }
            nt_store(&XM[(size_t)grp * 2048 + 512 * unit + 64 * j + lane], make_uint4(out[0], out[1], out[2], out[3]));
        }
        // ---- XL: j = 4 (s >> 1) + 2 m + (s & 1): 2 x 64, threads 0..127
        if (t < 128) {
            const int lane = t & 63, m = t >> 6;
            const int c = lane & 15, g = lane >> 4, j = 4 * (s >> 1) + 2 * m + (s & 1);
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
                    for (int ry = 0; ry < 4; ++ry) acc = __builtin_amdgcn_udot4(PXS(c, 4 * g + ry, Xp + i), 0x01010101u, acc, false);
                    sacc[i] = acc;
                }
                out[2 * hh] = sacc[0] | (sacc[1] << 16);
                out[2 * hh + 1] = sacc[2] | (sacc[3] << 16);
            }
            nt_store(&XL[(size_t)grp * 512 + 64 * j + lane], make_uint4(out[0], out[1], out[2], out[3]));
        }
        __syncthreads();  // the slab is consumed: the next one may overwrite it
    }
    }  // groups of this block
#undef PXS
}

void launch_tile(const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, const Workspace& ws, int n_flags,
                 hipStream_t s, int max_blocks) {
    const int blocks = (n + 15) / 16;
    const bool fast = (g.width % 16 == 0) && (g.pitch % 16 == 0) && (g.frame_stride % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(d_luma) % 16 == 0);
    const int sb = max_blocks > 0 ? (blocks < max_blocks ? blocks : max_blocks) : blocks;
    if (fast)
        hipLaunchKernelGGL(k0_tile_slab<true>, dim3(sb), dim3(256), 0, s, d_luma, g.width, g.height, g.pitch,
                           g.frame_stride, g.cw, g.nctu, ctu0, n, ws.xs, ws.xm, ws.xl, ws.flags, n_flags);
    else
        hipLaunchKernelGGL(k0_tile_slab<false>, dim3(sb), dim3(256), 0, s, d_luma, g.width, g.height, g.pitch,
                           g.frame_stride, g.cw, g.nctu, ctu0, n, ws.xs, ws.xm, ws.xl, ws.flags, n_flags);
}

}  // namespace ethcnn
