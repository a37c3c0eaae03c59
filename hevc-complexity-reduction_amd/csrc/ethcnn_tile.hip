// ethcnn_tile.hip -- k0: the CTU-load stage.  Luma frames -> zero-padded 64x64 CTUs in raster
// order (get_Y_for_one_frame + the tiling loop, video_to_cu_depth.py:46-59,88-106) with the
// integer 2x2 / 4x4 pooled sums of aver_pool (net_CNN.py:62-63,126,132), written in the lane
// order the trunk (k1) consumes.  HBM-bound: 4096 B/CTU in, 6656 B/CTU out.
//
// One block = one GROUP of 16 consecutive CTUs (global raster index over the frame sequence).
// Stage 1: 16 x (64 rows x 64 B) 16-B loads, a frame row of horizontally adjacent CTUs is a
// contiguous run -> LDS.  Stage 2: every output record is a full, linear 1 KiB per wave
// instruction (lane = c + 16 g, c = CTU in the group, g = MFMA k-group):
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
constexpr int kCtuPitch = 64 * kRowPitch + 1;    // + 1: the 16 CTUs of a group start on 16 different banks

template <bool FAST>
__global__ __launch_bounds__(256) void k0_tile(const uint8_t* __restrict__ luma, int width, int height, long pitch,
                                               long frame_stride, int cw, int nctu, long ctu0, int n_total,
                                               uint4* __restrict__ XS, uint4* __restrict__ XM,
                                               uint4* __restrict__ XL, int* __restrict__ gate_flags, int n_flags) {
    __shared__ uint32_t tile[16 * kCtuPitch];
    const int t = threadIdx.x;
    const int grp = blockIdx.x, n0 = grp * 16;
    // first kernel of a pass: clear the pass's gate predicates (set by the heads kernel, two
    // kernels later) here instead of in a separate memset launch
    if (grp == 0)
        for (int i = t; i < n_flags; i += 256) gate_flags[i] = 0;

    // ---- load 16 CTUs (zero outside the frame / beyond n_total): thread -> (row, 16-B segment)
    {
        const int row = t >> 2, seg = t & 3;
        uint4 vv[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
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
            vv[c] = v;
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) {  // all 16 loads are in flight before the first LDS write
            uint32_t* dst = &tile[c * kCtuPitch + row * kRowPitch + seg * 4];
            dst[0] = vv[c].x; dst[1] = vv[c].y; dst[2] = vv[c].z; dst[3] = vv[c].w;
        }
    }
    __syncthreads();
#define PX(c, Y, Xd) tile[(c) * kCtuPitch + (Y) * kRowPitch + (Xd)]

    // ---- XS: 16 units x 4 j x 64 lanes = 4096 uint4, 16 per thread, linear in the output
#pragma unroll 8
    for (int rep = 0; rep < 16; ++rep) {
        const int e = t + 256 * rep;
        const int lane = e & 63, j = (e >> 6) & 3, u = e >> 8;
        const int c = lane & 15, g = lane >> 4, uy = u >> 2, ux = u & 3;
        uint32_t d[4];
#pragma unroll
        for (int q1 = 0; q1 < 4; ++q1)
            d[q1] = PX(c, 16 * uy + 8 * (j >> 1) + 4 * (q1 >> 1) + g, 4 * ux + 2 * (j & 1) + (q1 & 1));
        XS[(size_t)grp * 4096 + e] = make_uint4(d[0], d[1], d[2], d[3]);
    }
    // ---- XM: 4 units x 8 j x 64 lanes = 2048 uint4, 8 per thread
#pragma unroll 4
    for (int rep = 0; rep < 8; ++rep) {
        const int e = t + 256 * rep;
        const int lane = e & 63, j = (e >> 6) & 7, unit = e >> 9;
        const int c = lane & 15, g = lane >> 4, uy = unit >> 1, ux = unit & 1;
        uint32_t out[4];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int d = 2 * j + hh, q2 = d >> 2, q1 = d & 3;
            const int Yp = 16 * uy + 8 * (q2 >> 1) + 4 * (q1 >> 1) + g;  // pooled row (0..31)
            const int Xp = 16 * ux + 8 * (q2 & 1) + 4 * (q1 & 1);        // pooled col of px 0
            const uint32_t a0 = PX(c, 2 * Yp, Xp >> 1), a1 = PX(c, 2 * Yp, (Xp >> 1) + 1);
            const uint32_t b0 = PX(c, 2 * Yp + 1, Xp >> 1), b1 = PX(c, 2 * Yp + 1, (Xp >> 1) + 1);
            // pooled px i uses bytes 2i, 2i+1 of the 8-byte row pair
            const uint32_t s0 = (a0 & 0xff) + ((a0 >> 8) & 0xff) + (b0 & 0xff) + ((b0 >> 8) & 0xff);
            const uint32_t s1 = ((a0 >> 16) & 0xff) + (a0 >> 24) + ((b0 >> 16) & 0xff) + (b0 >> 24);
            const uint32_t s2 = (a1 & 0xff) + ((a1 >> 8) & 0xff) + (b1 & 0xff) + ((b1 >> 8) & 0xff);
            const uint32_t s3 = ((a1 >> 16) & 0xff) + (a1 >> 24) + ((b1 >> 16) & 0xff) + (b1 >> 24);
            out[2 * hh] = s0 | (s1 << 16);
            out[2 * hh + 1] = s2 | (s3 << 16);
        }
        XM[(size_t)grp * 2048 + e] = make_uint4(out[0], out[1], out[2], out[3]);
    }
    // ---- XL: 8 j x 64 lanes = 512 uint4, 2 per thread
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int e = t + 256 * rep;
        const int lane = e & 63, j = e >> 6;
        const int c = lane & 15, g = lane >> 4;
        uint32_t out[4];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int d = 2 * j + hh, q2 = d >> 2, q1 = d & 3;
            const int Yp = 8 * (q2 >> 1) + 4 * (q1 >> 1) + g;  // pooled row (0..15)
            const int Xp = 8 * (q2 & 1) + 4 * (q1 & 1);        // pooled col == dword col
            uint32_t sacc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t acc = 0;
#pragma unroll
                for (int ry = 0; ry < 4; ++ry)
                    acc = __builtin_amdgcn_udot4(PX(c, 4 * Yp + ry, Xp + i), 0x01010101u, acc, false);
                sacc[i] = acc;
            }
            out[2 * hh] = sacc[0] | (sacc[1] << 16);
            out[2 * hh + 1] = sacc[2] | (sacc[3] << 16);
        }
        XL[(size_t)grp * 512 + e] = make_uint4(out[0], out[1], out[2], out[3]);
    }
#undef PX
}

void launch_tile(const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, const Workspace& ws, int n_flags,
                 hipStream_t s) {
    const int blocks = (n + 15) / 16;
    const bool fast = (g.width % 16 == 0) && (g.pitch % 16 == 0) && (g.frame_stride % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(d_luma) % 16 == 0);
    if (fast)
        hipLaunchKernelGGL(k0_tile<true>, dim3(blocks), dim3(256), 0, s, d_luma, g.width, g.height, g.pitch,
                           g.frame_stride, g.cw, g.nctu, ctu0, n, ws.xs, ws.xm, ws.xl, ws.flags, n_flags);
    else
        hipLaunchKernelGGL(k0_tile<false>, dim3(blocks), dim3(256), 0, s, d_luma, g.width, g.height, g.pitch,
                           g.frame_stride, g.cw, g.nctu, ctu0, n, ws.xs, ws.xm, ws.xl, ws.flags, n_flags);
}

}  // namespace ethcnn
