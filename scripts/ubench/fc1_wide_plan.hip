// fc1_wide_plan.hip -- MEASURED AND NOT KEPT (round 3, profiles/r03_fc1_wide_plan.txt).  An alternative launch plan for FC1 of
// a big pass; bit-exact (the GPU parity + launch-plan tests ran through it), 2.5 % slower than the split plan stage-alone and
// 14 % slower inside the timed region (256 VGPRs x 2 waves per SIMD leave no registers for the co-resident CTU-load wave; hipcc
// ignores amdgpu_num_vgpr here).  Kept as a record of the experiment, not built: to rebuild, copy into
// hevc-complexity-reduction_amd/csrc/ as ethcnn_fc1_wide.hip, add it to HIPSRC, declare fc1_wide_ok / launch_fc1_wide in
// ethcnn_kernels.h and call launch_fc1_wide(w, c->dw, n, w.h1, cus, stream) in run_pass instead of launch_fc1.
// Knock-out builds (-DWIDE_KO=1|2|4, wrong results) time the K loop without its barrier / DMA / LDS reads.
//
// FC1 of a big pass (net_CNN.py:156,164,177), "wide" launch plan: ONE 8-wave block per CU that owns ALL
// 448 columns of its rows.
//
// The split plan (ethcnn_dense.hip: 128 x 112 tiles, 4 column blocks per M tile, 3 blocks per CU) reads every feature row
// through four different blocks.  They are mapped to one XCD so that three of the four reads hit its L2, but the blocks drift
// apart over the 168 K chunks and the features cross the fabric 1.75 times (profiles/r03_pmc_c3.txt).  It also comes in quanta
// of 768 blocks: 3328 blocks = 4.3 rounds at C3, 1.04 rounds at C2.  Here:
//   * grid = one block per CU (116 KB of LDS: a second one cannot land on the CU), 8 waves = (column quarter cq, row half rh);
//     the two waves of a SIMD share a column quarter;
//   * the row groups (16 CTUs each) of the pass are dealt out evenly, block b gets G / grid or one more; a block walks its share
//     in row tiles of 6..9 groups, each a full serial K loop (same ascending-k chain per accumulator as every other FC1
//     shape: results are bit-identical) -- C3: 25 groups = 9 + 8 + 8, three K loops per CU, no ragged last round;
//   * per 16-k chunk the block stages the W1 slice of ALL columns (28 KB, the four 112-column images of the split plan side by
//     side) and ONE 1 KB feature piece per row group by LDS-DMA: every feature byte is fetched exactly once per pass;
//   * per chunk and wave: 4 or 5 feature reads (ds_read_b128) + 28 W reads (ds_read_b32) for 112..140 MFMAs (split plan:
//     30 reads for 56), one barrier per 7168..8064 MFMA cycles per SIMD instead of one per 1792.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ethcnn_fc1_tile.h"
#include "ethcnn_kernels.h"

namespace ethcnn {

constexpr int kWideBFloats = 16 * kNVec;       // W1 rows of one 16-k chunk, all 448 columns: four images of 16 x 112
constexpr int kWideImg = 16 * 112;             // one column quarter's image of a chunk (pack_fc1_image(.., 112, 16, ..))
constexpr int kWideRMax = 9;                   // row groups per tile
constexpr int kWideStage = kWideBFloats + kWideRMax * 256;
constexpr int kWideNst = 3;
constexpr int kWideNK = kNFeat / 16;

struct WideParams {
    const float* feat;   // [groups][k/4][16][4]
    const float* wimg;   // fc1_img112: [4 column quarters][168 chunks][16 x 112 image]
    const float* bias;
    float* out;          // h1 [M][448]
    int M;               // rows (CTUs) of the pass
};

// one row tile: groups [g0, g0 + NR0 + NR1) of the pass, waves with rh = 0 own the first NR0 groups, rh = 1 the next NR1;
// rows at or beyond `row_end` are computed on clamped addresses and dropped by the store's range check
template <int NRW, int R>
__device__ __forceinline__ void wide_tile_wave(float* __restrict__ smem, const WideParams& P, const int g0, const int r0, const int glast,
                                               const int row_end, const unsigned wvu, const int cq) {
    constexpr int NS = 7, BN = 112;
    constexpr int NK = kWideNK, NST = kWideNst, DIST = NST - 1, STAGE = kWideStage;
    constexpr int PIECES = 28 + R;                 // 1 KB DMA pieces per chunk: 28 of W1, R of features
    constexpr int ISSUE = (PIECES + 7) / 8;        // every wave issues exactly ISSUE (the tail duplicates the last piece)
    static_assert(NK % NST == 0, "K chunks come in whole rounds of the stage ring");
    static_assert(DIST * ISSUE <= 63, "vmcnt is a 6-bit counter");

    const int lane = threadIdx.x & 63;
    const int col = lane & 15, g = lane >> 4;
    const unsigned lane16 = (unsigned)lane * 16u;

    f32x4 acc[NRW][NS];
#pragma unroll
    for (int i = 0; i < NRW; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // DMA pieces of this wave: wave-uniform 64-bit source bases + per-chunk strides in SGPRs, wave-uniform LDS destinations
    const float* src[ISSUE];
    unsigned step[ISSUE], dst[ISSUE];
#pragma unroll
    for (int i = 0; i < ISSUE; ++i) {
        const unsigned p = min(wvu + 8u * i, (unsigned)(PIECES - 1));
        if (p < 28u) {
            const unsigned nb = p / 7u, q = p % 7u;
            src[i] = P.wimg + (size_t)nb * NK * kWideImg + q * 256u;
            step[i] = kWideImg;
            dst[i] = 4u * (nb * kWideImg + q * 256u);
        } else {
            const int r = (int)p - 28;
            src[i] = P.feat + (size_t)min(g0 + r, glast) * (kNFeat / 4) * 64;
            step[i] = 256;
            dst[i] = 4u * (kWideBFloats + r * 256);
        }
    }
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)smem);

    int bcol[NS], brow[4];
#pragma unroll
    for (int j = 0; j < NS; ++j) bcol[j] = j * 16 + col;
#pragma unroll
    for (int e = 0; e < 4; ++e) brow[e] = (e ^ (g & 1)) * BN;   // the image's bank permutation (BN % 32 == 16), see ethcnn_dense.hip
    const float* a_lds = smem + kWideBFloats + r0 * 256 + lane * 4;
    const float* b_lds = smem + cq * kWideImg + 4 * g * BN;

#define WD_DMA(sbase, lds_byte_addr)                                                                   \
    {                                                                                                  \
        unsigned keep_;                                                                                \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(lane16), "s"(sbase), "s"(lds_byte_addr) : "memory");          \
    }
#define WD_ISSUE(kc, st)                                                                               \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < ISSUE; ++i)                                              \
            WD_DMA(src[i] + (size_t)(kc) * step[i], lds_base + 4u * (st) * STAGE + dst[i]);            \
    }
    // Operands of HALF a chunk (two of its four 4-k MFMA steps) live in registers, double buffered: the barrier of a chunk
    // sits in its MIDDLE -- behind it the wave still holds the operands of the second half, and it fetches the first half of
    // the next chunk (landed: every wave waited for its own DMAs before the barrier) under those MFMAs.  No LDS latency is
    // ever exposed at a chunk boundary, which a block that is alone on its CU could not hide behind another block.
    struct Half {
        float2 a[NRW];
        float b[2][NS];
    };
#define WD_LOAD(h, st, half)                                                                           \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NRW; ++i)                                                \
            (h).a[i] = *reinterpret_cast<const float2*>(a_lds + (st) * STAGE + i * 256 + 2 * (half));  \
        const float* bs = b_lds + (st) * STAGE;                                                        \
        _Pragma("unroll") for (int e2 = 0; e2 < 2; ++e2)                                               \
            _Pragma("unroll") for (int j = 0; j < NS; ++j) (h).b[e2][j] = bs[brow[2 * (half) + e2] + bcol[j]]; \
    }
#define WD_MFMA(h)                                                                                     \
    {                                                                                                  \
        _Pragma("unroll") for (int e2 = 0; e2 < 2; ++e2)                                               \
            _Pragma("unroll") for (int i = 0; i < NRW; ++i) {                                          \
                const float a = e2 ? (h).a[i].y : (h).a[i].x;                                          \
                _Pragma("unroll") for (int j = 0; j < NS; ++j) acc[i][j] = MFMA16(a, (h).b[e2][j], acc[i][j]); \
            }                                                                                          \
    }
    // iteration kc (stage st = kc % 3): second-half operands of chunk kc -> MFMAs of its first half -> chunk kc+1 landed
    // (its DMAs are this wave's only outstanding VMEM: vmcnt(0)) -> barrier -> refill the stage of chunk kc-1 (every wave
    // consumed it before it arrived here) with chunk kc+2 -> first-half operands of chunk kc+1 -> MFMAs of the second half
#ifndef WIDE_KO
#define WIDE_KO 0   // knock-out timing builds (WRONG results): 1 no barrier, 2 no DMA in the loop, 4 no LDS reads in the loop
#endif
#define WD_STEP(kc, st)                                                                                \
    {                                                                                                  \
        if (!(WIDE_KO & 4)) { WD_LOAD(Q, st, 1); }                                                     \
        WD_MFMA(Pq);                                                                                   \
        vm_wait<0>();                                                                                  \
        if (!(WIDE_KO & 1)) __builtin_amdgcn_s_barrier();                                              \
        if (!(WIDE_KO & 2) && (kc) + 2 < NK) { WD_ISSUE((kc) + 2, ((st) + 2) % NST); }                 \
        if (!(WIDE_KO & 4) && (kc) + 1 < NK) { WD_LOAD(Pq, ((st) + 1) % NST, 0); }                     \
        WD_MFMA(Q);                                                                                    \
    }

    Half Pq, Q;
#pragma unroll
    for (int c0 = 0; c0 < 2; ++c0) { WD_ISSUE(c0, c0); }
    vm_wait<ISSUE>();  // chunk 0 landed (chunk 1 may still be in flight)
    __builtin_amdgcn_s_barrier();
    WD_LOAD(Pq, 0, 0);
    if (WIDE_KO & 4) { WD_LOAD(Q, 0, 1); }
    for (int kc = 0; kc < NK; kc += NST) {
#pragma unroll
        for (int st = 0; st < NST; ++st) { WD_STEP(kc + st, st); }
    }
#undef WD_LOAD
#undef WD_MFMA
#undef WD_DMA
#undef WD_ISSUE
#undef WD_STEP

    // epilogue: bias + leaky-ReLU; the buffer resource ends at the tile's last row (or the pass's): rows beyond are dropped
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(P.out, 0, row_end * kNVec * 4, 0x00020000);
    const int n0 = cq * BN;
    const int lane_out = (((g0 + r0) * 16 + 4 * g) * kNVec + col) * 4;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const float bv = P.bias[n0 + j * 16 + col];
#pragma unroll
        for (int i = 0; i < NRW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float h = acc[i][j][r] + bv;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(0.2f * h, h)), rO,
                                                      lane_out + (i * 16 + r) * kNVec * 4, (n0 + j * 16) * 4, 0);
            }
    }
}

template <int NR0, int NR1>
__device__ __forceinline__ void wide_tile(float* __restrict__ smem, const WideParams& P, const int g0, const int ng, const unsigned wvu) {
    constexpr int R = NR0 + NR1;
#ifndef WIDE_MAP
#define WIDE_MAP 0
#endif
    // the two waves of a SIMD share a column quarter and split the tile's rows (4 + 5 groups on every SIMD, not 5 + 5 on two)
    const int cq = WIDE_MAP ? (int)(wvu >> 1) : (int)(wvu & 3u);
    const unsigned rh = WIDE_MAP ? (wvu & 1u) : (wvu >> 2);
    const int glast = g0 + ng - 1;                              // last group this tile may read
    const int row_end = min(P.M, (g0 + ng) * 16);               // first row this tile must not write
    if (NR0 == NR1) {
        wide_tile_wave<NR0, R>(smem, P, g0, (int)rh * NR0, glast, row_end, wvu, cq);
    } else if (rh == 0) {
        wide_tile_wave<NR0, R>(smem, P, g0, 0, glast, row_end, wvu, cq);
    } else {
        wide_tile_wave<NR1, R>(smem, P, g0, NR0, glast, row_end, wvu, cq);
    }
}

#ifdef WIDE_VGPRS  // A/B builds: leave registers for the co-resident CTU-load wave of the next pass (96 per SIMD)
#define WIDE_ATTR __attribute__((amdgpu_num_vgpr(WIDE_VGPRS)))
#else
#define WIDE_ATTR
#endif
__global__ __launch_bounds__(512) WIDE_ATTR void k_fc1_wide(WideParams P) {
    __shared__ __attribute__((aligned(16))) float smem[kWideNst * kWideStage];  // the ONLY LDS object: 116,736 B
    const unsigned wvu = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // this block's share of the pass's row groups
    const int G = (P.M + 15) >> 4;
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    const int base = G / nb, extra = G % nb;
    int g = b * base + min(b, extra);
    const int gend = g + base + (b < extra ? 1 : 0);
    const int n = gend - g;
    if (n <= 0) return;
    const int tiles = (n + kWideRMax - 1) / kWideRMax;
    __builtin_amdgcn_s_setprio(2);  // as the split plan: wins the issue arbitration against the co-resident CTU-load wave
    for (int t = 0; t < tiles; ++t) {
        const int ng = n / tiles + (t < n % tiles ? 1 : 0);   // 6..9 when n >= 6; smaller shares run the 6-group shape masked
        if (ng == 9) wide_tile<5, 4>(smem, P, g, ng, wvu);
        else if (ng == 8) wide_tile<4, 4>(smem, P, g, ng, wvu);
        else if (ng == 7) wide_tile<4, 3>(smem, P, g, ng, wvu);
        else wide_tile<3, 3>(smem, P, g, ng, wvu);
        g += ng;
    }
}

// the wide plan pays when every CU gets at least one full tile and few K loops are run on short tiles
bool fc1_wide_ok(int n, int cus) { return cus > 0 && (n + 15) / 16 >= 6 * cus; }

void launch_fc1_wide(const Workspace& ws, const DeviceWeights& w, int n, float* out, int cus, hipStream_t s) {
    WideParams P;
    P.feat = ws.feat;
    P.wimg = w.fc1_img112;
    P.bias = w.fc1_b;
    P.out = out;
    P.M = n;
    hipLaunchKernelGGL(k_fc1_wide, dim3(cus), dim3(512), 0, s, P);
}

}  // namespace ethcnn
