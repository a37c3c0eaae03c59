// ethcnn_dense.hip -- FC1: h1[N,448] = lrelu(feat[N,2688] . W1 + b1)  (net_CNN.py:156,164,177;
// the three heads' FC1 matrices side by side, 64 | 128 | 256 columns).  77.6 % of the path's
// MACs: the dominant kernel.  gfx950, v_mfma_f32_16x16x4_f32 (exact fp32, 157.3 TFLOP/s peak).
//
// Block = WM waves stacked along M; a wave owns MS groups of 16 CTUs x (16 NS) columns; the
// block's waves share one BK x BN slice of W1 per K chunk (BK = 16 NSUB, BN = 16 NS).
//   * W1 slice (B operand): global_load_lds (LDS-DMA, no staging VGPRs), three stages, one
//     barrier per chunk.  The host packs W1 per column block in exactly the LDS image order
//     (ethcnn_weights.cpp::pack_fc1_image), so every DMA instruction is a linear 1 KiB copy.
//     The image is bank-permuted: the two k-groups of a 32-lane half read rows e and e+4, so
//     rows with (k>>2) odd have adjacent 16-column groups swapped (BN % 32 == 0) or adjacent
//     rows swapped (BN % 32 == 16); undone on the ds_read.
//   * features (A operand): the trunk writes them as feat[group of 16 CTUs][k/4][16 CTUs][4 floats],
//     so a wave's operands of one 16-k sub-chunk are ONE linear 1 KiB piece: one DMA instruction
//     into the wave's private LDS slot, read back as one float4 per lane (ctu, g), whose element
//     e feeds MFMA step e.  That fixes the accumulation order inside a sub-chunk to k = 16c + 4g + e
//     (e outer, g inner): the canonical FC order of DESIGN.md, restated by the oracle.
//   * NSPLIT column blocks per M tile.  The dispatcher puts block b on XCD b % 8; the blocks of
//     one M tile are mapped to consecutive slots of the same XCD so the tile's features cross
//     the fabric once and are shared through that XCD's L2 (measured: FETCH_SIZE / NSPLIT, same
//     or better time).  Speed only; results are placement-independent.
//   * one ascending-chunk MFMA chain per accumulator, no split-K; bias + leaky-ReLU epilogue.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ethcnn_kernels.h"
#include "ethcnn_fc1_regs.h"
#include "ethcnn_fc1_tile.h"

namespace ethcnn {

template <int MS, int NS, int WM, int NSUB, bool GROUP, int NST = 3>
__global__ __launch_bounds__(64 * WM) void k_fc1_p3(const float* __restrict__ feat, const float* __restrict__ Wimg,
                                                    const float* __restrict__ bias, float* __restrict__ out, int M) {
    __shared__ __attribute__((aligned(16))) float smem[Fc1Shape<MS, NS, WM, NSUB, NST>::LDS_FLOATS];  // the ONLY LDS object
    fc1_tile<MS, NS, WM, NSUB, GROUP, NST>(smem, feat, Wimg, bias, out, M, blockIdx.x);
}

// The bulk launch of a big pass: rows [0, m_main) as 128 x 112 tiles (a whole number of blocks per CU) AND the remaining rows
// [m_main, m_total) as 64 x 112 tiles, in ONE grid.  The remainder's blocks come FIRST (rem_blocks of them, a multiple of 8 so
// that the XCD grouping of both parts holds), so they run beside two bulk blocks per CU from the start: their short
// per-chunk MFMA chains, latency-bound when such a launch has the GPU to itself (107 us for 3696 rows), are absorbed at the
// bulk rate instead (the matrix pipe just sees 4 % more MFMAs).  Same chains per accumulator: results are identical.
#ifdef FC1_STAMPS
// development probe (scripts/ubench/fc1_probe.hip): device-wide 100 MHz stamps at entry / exit of every block
__device__ unsigned long long g_fc1_stamps[1 << 14][2];
__device__ __forceinline__ void fc1_stamp(int slot) {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    if (threadIdx.x == 0) g_fc1_stamps[blockIdx.x & 0x3fff][slot] = t;
}
#define FC1_STAMP(i) fc1_stamp(i)
#else
#define FC1_STAMP(i)
#endif

#ifndef FC1_BULK_NST
#define FC1_BULK_NST 3  // LDS stages of the bulk shape (A/B builds: 2 = 30 KB per block, 4 blocks per CU)
#endif
template <int NS, int WM, int NSUB>
__global__ __launch_bounds__(64 * WM) void k_fc1_bulk(const float* __restrict__ feat, const float* __restrict__ Wimg,
                                                      const float* __restrict__ bias, float* __restrict__ out, int m_main,
                                                      int m_total, unsigned rem_blocks) {
    __shared__ __attribute__((aligned(16))) float smem[Fc1Shape<2, NS, WM, NSUB, FC1_BULK_NST>::LDS_FLOATS];  // the ONLY LDS object
    FC1_STAMP(0);
    if (blockIdx.x < rem_blocks)
        fc1_tile<1, NS, WM, NSUB, true, FC1_BULK_NST>(smem, feat + (size_t)(m_main / 16) * kNFeat * 16, Wimg, bias, out + (size_t)m_main * kNVec,
                                           m_total - m_main, blockIdx.x);
    else
        fc1_tile<2, NS, WM, NSUB, true, FC1_BULK_NST>(smem, feat, Wimg, bias, out, m_main, blockIdx.x - rem_blocks);
    FC1_STAMP(1);
}

template <int MS, int NS, int WM, int NSUB, bool GROUP = false, int NST = 3>
static void launch_fc1_p3(const float* feat, const float* wimg, const float* bias, float* out, int M, hipStream_t s) {
    constexpr int BM = 16 * MS * WM, NSPLIT = kNVec / (16 * NS);
    const int mtiles = (M + BM - 1) / BM;
    const int blocks = GROUP ? ((mtiles + 7) / 8) * 8 * NSPLIT : mtiles * NSPLIT;
    hipLaunchKernelGGL((k_fc1_p3<MS, NS, WM, NSUB, GROUP, NST>), dim3(blocks), dim3(64 * WM), 0, s, feat, wimg, bias, out, M);
}

// Short row ranges, register-fed (ethcnn_fc1_regs.h): 64 CTUs x 16 NS columns per block, no LDS, one wave per 16 CTUs with NS
// dependent chains.  A block's time is its chain's -- 672 links x max(45 cycles, 32 x the chains sharing its SIMD) -- not a
// per-chunk barrier + LDS round trip: few rows finish in one chain time (~13 us), many rows run at the matrix pipe's rate.
template <int NS, int D>
__global__ __launch_bounds__(256) void k_fc1_regs(const float* __restrict__ feat, const float* __restrict__ wlane,
                                                  const float* __restrict__ bias, float* __restrict__ out, int M) {
    int mt, nb;
    fc1_block_to_tile<kNVec / (16 * NS), true>(blockIdx.x, mt, nb);
    if (mt * 64 >= M) return;
    fc1_tile_regs<NS, D, false>(feat, wlane, bias, out, M, mt, nb);
}
template <int NS, int D>
static void launch_fc1_regs(const float* feat, const float* wlane, const float* bias, float* out, int M, hipStream_t s) {
    constexpr int NSPLIT = kNVec / (16 * NS);
    const int mtiles = (M + 63) / 64;
    hipLaunchKernelGGL((k_fc1_regs<NS, D>), dim3(((mtiles + 7) / 8) * 8 * NSPLIT), dim3(256), 0, s, feat, wlane, bias, out, M);
}

static int fc1_variant() {
    static int v = -2;
    if (v == -2) {
        const char* e = dev_env("ETHCNN_FC1_VARIANT");  // development knob: tile-shape A/B runs
        v = e ? atoi(e) : -1;
    }
    return v;
}

// Tile shape for a SHORT row range (the remainder of the split launch, or a whole small batch such
// as one LDP frame).  Here every block runs the full serial K loop once, so the time is the K-loop
// latency of the shape times the number of 256-block rounds, not the MFMA rate: measured
// (profiles/r01_fc1_rows.txt) t[us] ~= a + b * ceil(blocks / 256).  Narrow shapes have short chains
// (few MFMAs per K chunk per wave) but need more blocks.
static int fc1_short_variant(int n) {
    struct Shape { int variant, bm, nsplit; double a, b; };
    // a, b refitted in round 2 (gpurun_out/fc1_rows.txt after the SGPR-base addressing; r01: profiles/r01_fc1_rows.txt)
    static const Shape shapes[] = {{0, 128, 4, 15.0, 150.0}, {1, 64, 4, 30.0, 70.0},  {2, 64, 7, 9.0, 45.0},
                                   {3, 64, 14, 10.0, 22.5},  {4, 64, 28, 8.0, 14.4}, {5, 64, 14, 9.5, 22.6}};
    // register-fed shapes (k_fc1_regs; round 3, profiles/r03_fc1_rows.txt): 256-510 rows 15.9 us (LDS-staged best: 21.9), 924 rows
    // 26.6 (34.1), 1536 46.7 (48.2), 2040 48.5 (59.3), 3696 89.0 (98.2), 6000 138 (142); from 8000 rows the staged shapes win again
    if (n <= 576) return 7;
    if (n <= 1152) return 8;
    if (n <= 1792) return 7;   // (672 blocks: one round at three blocks per CU, 40 us at 1536 rows against 46-47)
    if (n <= 6400) return 9;
    int best = 2;
    double tbest = 1e30;
    for (const Shape& sh : shapes) {
        const int blocks = ((n + sh.bm - 1) / sh.bm) * sh.nsplit;
        const double t = sh.a + sh.b * ((blocks + 255) / 256);
        if (t < tbest) { tbest = t; best = sh.variant; }
    }
    return best;
}

// one shape over rows [row0, row0 + rows) (row0 a multiple of 16): the kernel sees shifted pointers
static void launch_fc1_rows(int variant, const Workspace& ws, const DeviceWeights& w, int row0, int rows, float* out,
                            hipStream_t s) {
    const float* feat = ws.feat + (size_t)(row0 / 16) * kNFeat * 16;
    float* o = out + (size_t)row0 * kNVec;
    switch (variant) {  // production shapes run XCD-grouped; 10..12 are the ungrouped A/B twins
        default:  // 0: 128 CTUs (4 waves x 2 groups) x 112 columns (N split 4), BK 16
            launch_fc1_p3<2, 7, 4, 1, true>(feat, w.fc1_img112, w.fc1_b, o, rows, s);
            break;
        case 1:  // 64 CTUs x 112 columns
            launch_fc1_p3<1, 7, 4, 1, true>(feat, w.fc1_img112, w.fc1_b, o, rows, s);
            break;
        case 2:  // 64 CTUs x 64 columns (N split 7), BK 32
            launch_fc1_p3<1, 4, 4, 2, true>(feat, w.fc1_img64, w.fc1_b, o, rows, s);
            break;
        case 3:  // 64 CTUs x 32 columns (N split 14), BK 32: short K-chain time for short row ranges
            launch_fc1_p3<1, 2, 4, 2, true>(feat, w.fc1_img32, w.fc1_b, o, rows, s);
            break;
        case 4:  // 64 CTUs x 16 columns (N split 28), BK 64
            launch_fc1_p3<1, 1, 4, 4, true>(feat, w.fc1_img16, w.fc1_b, o, rows, s);
            break;
        case 5:  // 64 CTUs x 32 columns, BK 64
            launch_fc1_p3<1, 2, 4, 4, true>(feat, w.fc1_img32, w.fc1_b, o, rows, s);
            break;
        case 6:  // 32 CTUs (2 waves) x 32 columns, BK 64
            launch_fc1_p3<1, 2, 2, 4, true>(feat, w.fc1_img32, w.fc1_b, o, rows, s);
            break;
        case 7: launch_fc1_regs<1, 16>(feat, w.fc1_lane16, w.fc1_b, o, rows, s); break;  // register-fed, 64 x 16
        case 8: launch_fc1_regs<2, 10>(feat, w.fc1_lane16, w.fc1_b, o, rows, s); break;  // register-fed, 64 x 32
        case 9: launch_fc1_regs<4, 6>(feat, w.fc1_lane16, w.fc1_b, o, rows, s); break;   // register-fed, 64 x 64
        case 10: launch_fc1_p3<2, 7, 4, 1, false>(feat, w.fc1_img112, w.fc1_b, o, rows, s); break;
        case 11: launch_fc1_p3<1, 7, 4, 1, false>(feat, w.fc1_img112, w.fc1_b, o, rows, s); break;
        case 12: launch_fc1_p3<1, 4, 4, 2, false>(feat, w.fc1_img64, w.fc1_b, o, rows, s); break;
    }
}

void launch_fc1(const Workspace& ws, const DeviceWeights& w, int n, float* out, hipStream_t s) {
    const int variant = fc1_variant();
    if (variant >= 100) {  // A/B knob: default split launch, remainder forced to shape variant - 100
        const int bt = ((n / 128) * 4 / 256) * 256 / 4;
        if (bt > 0) launch_fc1_rows(0, ws, w, 0, bt * 128, out, s);
        if (n > bt * 128) launch_fc1_rows(variant - 100, ws, w, bt * 128, n - bt * 128, out, s);
        return;
    }
    if (variant >= 0 && variant != 20 && variant != 30) {  // A/B knob: one fixed shape (20 = the best single shape for n)
        launch_fc1_rows(variant, ws, w, 0, n, out, s);
        return;
    }
    if (variant == 20) {
        launch_fc1_rows(fc1_short_variant(n), ws, w, 0, n, out, s);
        return;
    }
    // Default, split launch: the 128 x 112 shape (best plateau) on as many rows as give a whole number of
    // blocks per CU (a multiple of 256 blocks), the remaining rows (< 8192) with the lowest-latency shape.
    const int big_tiles = ((n / 128) * 4 / 256) * 256 / 4;
    const int row0 = big_tiles * 128;
    // bulk + remainder in one grid -- only when the bulk part runs for several rounds of resident blocks: the remainder's
    // blocks take a slot on some CUs for a third of the launch, and with a one-round bulk part (c2: 24,576 + 924 rows) that
    // pushes bulk blocks into a second round (measured: c3 +1.1 %, c4 +0.4 %, c2 -5 %; profiles/r02_fc1_variants.txt section 7)
    if (big_tiles >= 3 * 192 && n > row0 && fc1_variant() != 30) {  // 192 tiles of 128 x 4 column blocks = one round (30: A/B knob)
        const int rem_tiles = (n - row0 + 63) / 64;
        const unsigned rem_blocks = (unsigned)((rem_tiles + 7) / 8) * 8 * 4;   // GROUP mapping: tiles in eights, 4 column blocks
        const unsigned main_blocks = (unsigned)((big_tiles + 7) / 8) * 8 * 4;
        hipLaunchKernelGGL((k_fc1_bulk<7, 4, 1>), dim3(rem_blocks + main_blocks), dim3(256), 0, s, ws.feat, w.fc1_img112, w.fc1_b, out,
                           row0, n, rem_blocks);
        return;
    }
    if (big_tiles > 0) launch_fc1_rows(0, ws, w, 0, row0, out, s);
    if (n > row0) launch_fc1_rows(fc1_short_variant(n - row0), ws, w, row0, n - row0, out, s);
}

}  // namespace ethcnn
