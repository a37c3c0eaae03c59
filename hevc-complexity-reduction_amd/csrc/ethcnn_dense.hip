// ethcnn_dense.hip -- FC1: h1[N,448] = lrelu(feat[N,2688] . W1 + b1)  (net_CNN.py:156,164,177;
// the three heads' FC1 matrices side by side, 64 | 128 | 256 columns).  77.6 % of the path's
// MACs: the dominant kernel.  gfx950, v_mfma_f32_16x16x4_f32 (exact fp32, 157.3 TFLOP/s peak).
//
// Block = WM waves stacked along M; a wave owns MS groups of 16 CTUs x (16 NS) columns; the
// block's waves share one BK x BN slice of W1 per K chunk (BK = 16 NSUB, BN = 16 NS).
//   * W1 slice (B operand): global_load_lds (LDS-DMA, no staging VGPRs), double buffered, one
//     barrier per chunk.  The host packs W1 per column block in exactly the LDS image order
//     (ethcnn_weights.cpp::pack_fc1_image), so every DMA instruction is a linear 1 KiB copy.
//     The image is bank-permuted: the two k-groups of a 32-lane half read rows e and e+4, so
//     rows with (k>>2) odd have adjacent 16-column groups swapped (BN % 32 == 0) or adjacent
//     rows swapped (BN % 32 == 16); undone on the ds_read.
//   * features (A operand): NOT staged through LDS.  The trunk writes them as
//     feat[group of 16 CTUs][k/4][16 CTUs][4 floats]; lane (ctu, g) loads ONE float4 per
//     16-k sub-chunk -- a fully coalesced 1 KiB per wave instruction -- and feeds element e to
//     MFMA step e.  That fixes the accumulation order inside a sub-chunk to k = 16c + 4g + e
//     (e outer, g inner): the canonical FC order of DESIGN.md, restated by the oracle.
//   * NSPLIT column blocks per M tile, column block = blockIdx.x % NSPLIT: the dispatcher puts
//     block b on XCD b % 8, so each XCD streams only 1/NSPLIT of W1 and keeps it L2-resident
//     (W1 is 4.8 MB, an XCD's L2 4 MiB).  Speed only; results are placement-independent.
//   * one ascending-chunk MFMA chain per accumulator, no split-K; bias + leaky-ReLU epilogue.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ethcnn_kernels.h"

namespace ethcnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int MS, int NS, int WM, int NSUB>
__global__ __launch_bounds__(64 * WM) void k_fc1(const float* __restrict__ feat, const float* __restrict__ Wimg,
                                                 const float* __restrict__ bias, float* __restrict__ out, int M) {
    constexpr int BK = 16 * NSUB, BN = 16 * NS, NSPLIT = kNVec / BN, BM = 16 * MS * WM;
    constexpr int NK = kNFeat / BK;
    constexpr int B_FLOATS = BK * BN;
    constexpr bool COLSWZ = (BN % 32 == 0);
    constexpr int B_INST = B_FLOATS / 256;         // 1 KiB wave-instructions per W1 chunk
    constexpr int B_PER = (B_INST + WM - 1) / WM;  // per wave
    static_assert(kNVec % BN == 0 && kNFeat % BK == 0 && B_FLOATS % 256 == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) float smem[2 * B_FLOATS];  // the ONLY LDS object

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 15, g = lane >> 4;
    const int nb = (int)(blockIdx.x % NSPLIT);
    const int m0 = (int)(blockIdx.x / NSPLIT) * BM + wv * 16 * MS;
    const int n0 = nb * BN;

    f32x4 acc[MS][NS];
#pragma unroll
    for (int i = 0; i < MS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // feat[group][k/4][16][4]: this lane's float4 of sub-chunk s is at k/4 = 4 s + g.  Groups past
    // the batch are clamped (loaded, never stored).
    const int ngroups = (M + 15) >> 4;
    const float* a_src[MS];
#pragma unroll
    for (int i = 0; i < MS; ++i)
        a_src[i] = feat + ((size_t)min((m0 >> 4) + i, ngroups - 1) * (kNFeat / 4) + g) * 64 + col * 4;
    const float* b_src = Wimg + (size_t)nb * NK * B_FLOATS + (size_t)wv * 256 + lane * 4;
    // read side: rows 16 u + 4 g + e have ((row>>2)&1) == (g&1)
    int bcol[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) bcol[j] = (j * 16 + col) ^ (COLSWZ ? ((g & 1) << 4) : 0);
    int brow[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) brow[e] = (COLSWZ ? e : (e ^ (g & 1))) * BN;

#define FC1_B_ISSUE(kc, buf)                                                                         \
    _Pragma("unroll") for (int i = 0; i < B_PER; ++i) {                                              \
        if ((i + 1) * WM <= B_INST || wv + i * WM < B_INST)                                          \
            __builtin_amdgcn_global_load_lds((glb_void*)(b_src + (size_t)(kc) * B_FLOATS + i * WM * 256), \
                                             (lds_void*)(smem + (buf) * B_FLOATS + (wv + i * WM) * 256), 16, 0, 0); \
    }

    float4 a_cur[NSUB][MS], a_nxt[NSUB][MS];
    FC1_B_ISSUE(0, 0);
#pragma unroll
    for (int u = 0; u < NSUB; ++u)
#pragma unroll
        for (int i = 0; i < MS; ++i) a_cur[u][i] = *reinterpret_cast<const float4*>(a_src[i] + u * 256);
    __syncthreads();  // hipcc drains vmcnt here (LDS-DMA pending): chunk 0 has landed
    for (int kc = 0; kc < NK; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < NK) {  // both in flight during the MFMAs below
            FC1_B_ISSUE(kc + 1, buf ^ 1);
#pragma unroll
            for (int u = 0; u < NSUB; ++u)
#pragma unroll
                for (int i = 0; i < MS; ++i)
                    a_nxt[u][i] = *reinterpret_cast<const float4*>(a_src[i] + ((size_t)(kc + 1) * NSUB + u) * 256);
        }
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            const float* bs = smem + buf * B_FLOATS + (16 * u + 4 * g) * BN;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float b[NS];
#pragma unroll
                for (int j = 0; j < NS; ++j) b[j] = bs[brow[e] + bcol[j]];
#pragma unroll
                for (int i = 0; i < MS; ++i) {
                    const float4 av = a_cur[u][i];
                    const float a = (e == 0) ? av.x : (e == 1) ? av.y : (e == 2) ? av.z : av.w;
#pragma unroll
                    for (int j = 0; j < NS; ++j) acc[i][j] = MFMA16(a, b[j], acc[i][j]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NSUB; ++u)
#pragma unroll
            for (int i = 0; i < MS; ++i) a_cur[u][i] = a_nxt[u][i];
        __syncthreads();
    }
#undef FC1_B_ISSUE

    // epilogue: C layout row (CTU) = 4g + r, col = lane & 15
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int n = n0 + j * 16 + col;
        const float bv = bias[n];
#pragma unroll
        for (int i = 0; i < MS; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + i * 16 + 4 * g + r;
                const float h = acc[i][j][r] + bv;
                if (m < M) out[(size_t)m * kNVec + n] = fmaxf(0.2f * h, h);
            }
    }
}

template <int MS, int NS, int WM, int NSUB>
static void launch_fc1_shape(const float* feat, const float* wimg, const float* bias, float* out, int M, hipStream_t s) {
    constexpr int BM = 16 * MS * WM, NSPLIT = kNVec / (16 * NS);
    hipLaunchKernelGGL((k_fc1<MS, NS, WM, NSUB>), dim3(((M + BM - 1) / BM) * NSPLIT), dim3(64 * WM), 0, s, feat, wimg,
                       bias, out, M);
}

static int fc1_variant() {
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("ETHCNN_FC1_VARIANT");  // development knob: tile-shape A/B runs
        v = e ? atoi(e) : -1;
    }
    return v;
}

// Tile shape by batch size.  Both shapes sit on the same plateau for large N (profiles/
// r01_fc1_variants.txt; 64x64 ~2.5 % lower).  What differs is how evenly the blocks divide over
// the 256 CUs when there are only a handful per CU (every block runs the full K loop, so a CU
// with 7 blocks finishes 1/6 later than one with 6).
static int fc1_auto_variant(int n) {
    const int tiles = (n + 63) / 64;
    auto balance = [](int blocks) { return (double)blocks / (256.0 * ((blocks + 255) / 256)); };
    const double e112 = 1.0 * balance(tiles * 4), e64 = 0.975 * balance(tiles * 7);
    return e64 > e112 ? 1 : 0;
}

void launch_fc1(const Workspace& ws, const DeviceWeights& w, int n, float* out, hipStream_t s) {
    int variant = fc1_variant();
    if (variant < 0) variant = fc1_auto_variant(n);
    switch (variant) {
        default:  // 0: 64 CTUs x 112 columns (N split 4), BK 16
            launch_fc1_shape<1, 7, 4, 1>(ws.feat, w.fc1_img112, w.fc1_b, out, n, s);
            break;
        case 1:  // 64 CTUs x 64 columns (N split 7), BK 32
            launch_fc1_shape<1, 4, 4, 2>(ws.feat, w.fc1_img64, w.fc1_b, out, n, s);
            break;
        case 2:  // 128 CTUs x 112 columns, 4 waves x 32 rows
            launch_fc1_shape<2, 7, 4, 1>(ws.feat, w.fc1_img112, w.fc1_b, out, n, s);
            break;
        case 3:  // 128 CTUs x 112 columns, 8 waves x 16 rows
            launch_fc1_shape<1, 7, 8, 1>(ws.feat, w.fc1_img112, w.fc1_b, out, n, s);
            break;
    }
}

}  // namespace ethcnn
