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
#include "ethcnn_tile_group.h"

namespace ethcnn {

template <bool FAST, bool WAIT>
__global__ __launch_bounds__(256) void k0_tile_slab(const uint8_t* __restrict__ luma, int width, int height, long pitch,
                                                    long frame_stride, int cw, int nctu, long ctu0, int n_total,
                                                    uint4* __restrict__ XS, uint4* __restrict__ XM,
                                                    uint4* __restrict__ XL, int* __restrict__ gate_flags, int n_flags, TileWait tw) {
    __shared__ uint32_t tile[16 * kSlabCtuPitch];
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < n_flags; i += 256) gate_flags[i] = 0;
    // persistent form: with fewer blocks than groups (one per CU when the stage runs beside FC1, so that its LDS never keeps
    // FC1's third block off a CU) a block walks several groups
    const int ngroups = (n_total + 15) >> 4;
#pragma unroll 1
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        if (WAIT) tile_wait_rows(tw, ctu0, grp, n_total, nctu, cw);
        tile_group<FAST, false, false>(tile, luma, width, height, pitch, frame_stride, cw, nctu, ctu0, n_total, grp, XS, XM, XL);
    }  // groups of this block
}

void launch_tile(const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, const Workspace& ws, int n_flags,
                 hipStream_t s, int max_blocks, const unsigned* wait_rows, unsigned wait_seq, unsigned* gave_up) {
    const int blocks = (n + 15) / 16;
    const bool fast = (g.width % 16 == 0) && (g.pitch % 16 == 0) && (g.frame_stride % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(d_luma) % 16 == 0);
    const int sb = max_blocks > 0 ? (blocks < max_blocks ? blocks : max_blocks) : blocks;
    const TileWait tw{wait_rows, wait_seq, gave_up};
    if (wait_rows && fast)
        hipLaunchKernelGGL((k0_tile_slab<true, true>), dim3(sb), dim3(256), 0, s, d_luma, g.width, g.height, g.pitch,
                           g.frame_stride, g.cw, g.nctu, ctu0, n, ws.xs, ws.xm, ws.xl, ws.flags, n_flags, tw);
    else if (wait_rows)
        hipLaunchKernelGGL((k0_tile_slab<false, true>), dim3(sb), dim3(256), 0, s, d_luma, g.width, g.height, g.pitch,
                           g.frame_stride, g.cw, g.nctu, ctu0, n, ws.xs, ws.xm, ws.xl, ws.flags, n_flags, tw);
    else if (fast)
        hipLaunchKernelGGL((k0_tile_slab<true, false>), dim3(sb), dim3(256), 0, s, d_luma, g.width, g.height, g.pitch,
                           g.frame_stride, g.cw, g.nctu, ctu0, n, ws.xs, ws.xm, ws.xl, ws.flags, n_flags, tw);
    else
        hipLaunchKernelGGL((k0_tile_slab<false, false>), dim3(sb), dim3(256), 0, s, d_luma, g.width, g.height, g.pitch,
                           g.frame_stride, g.cw, g.nctu, ctu0, n, ws.xs, ws.xm, ws.xl, ws.flags, n_flags, tw);
}

}  // namespace ethcnn
