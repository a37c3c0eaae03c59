// ethcnn_dense.hip -- fp32 MFMA dense layer for FC1 ([N,2688]x[2688,448], net_CNN.py:156,164,177)
// and the three FC2 layers (:159,167,180), gfx950.
//
//   out[m][n] = lrelu( sum_k A[m][k] W[k][n]  (+ qn * W[K][n])  + bias[n] )
//
// Block = WM waves stacked along M; every wave owns a (16 MS) x (16 NS) output tile, all waves
// of a block share one BK x BN slice of W per K chunk (BK = 16, BN = 16 NS).
//   * B (weights): global_load_lds (LDS-DMA, no staging VGPRs) into a LINEAR [16][BN] image,
//     double buffered, one barrier per chunk.  The two k-groups of a 32-lane half (g = 0,1)
//     read rows 4g+e, i.e. rows e and e+4, which would share banks; so rows with (k>>2) odd
//     are stored permuted -- adjacent 16-column groups swapped when BN % 32 == 0, adjacent
//     ROWS swapped when BN % 32 == 16 -- applied to the per-lane GLOBAL address of the DMA
//     (the LDS image stays linear) and undone on the ds_read.
//   * A (activations): NOT staged through LDS.  v_mfma_f32_16x16x4_f32 wants lane (row, g)
//     to supply A[row][k] for the step's 4 k values one per g; a lane instead loads ONE
//     float4 A[row][16c + 4g .. 4g+3] per chunk and feeds element e to MFMA step e.  That
//     fixes the accumulation order inside a chunk to k = 16c + 4g + e (e outer, g inner) --
//     the canonical FC1/FC2 order of DESIGN.md, restated by the oracle.
//   * NSPLIT column blocks per M tile, column block = blockIdx.x % NSPLIT: the dispatcher puts
//     block b on XCD b % 8, so each XCD streams only 1/NSPLIT of W and keeps it L2-resident
//     (the FC1 matrix is 4.8 MB, an XCD's L2 4 MiB).  Speed only; results are placement-free.
//   * one ascending-chunk MFMA chain per accumulator, no split-K.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ethcnn_kernels.h"

namespace ethcnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ float lrelu_d(float h) { return fmaxf(0.2f * h, h); }

template <int MS, int NS, int WM, int NSPLIT, bool QP, bool GROUP = false, int ABL = 0, int NSUB = 1, int PRIO = 0>
__global__ __launch_bounds__(64 * WM) void k_dense(const float* __restrict__ A, int lda, int K,
                                                   const float* __restrict__ W, const float* __restrict__ bias,
                                                   float qn, float* __restrict__ out, int ldo, int M) {
    // a K chunk = NSUB sub-chunks of 16 (one barrier per chunk); the accumulation order inside
    // every 16-wide sub-chunk is k = 16c + 4g + e regardless of NSUB.
    constexpr int BK = 16 * NSUB, BN = 16 * NS, LDW = BN * NSPLIT, BM = 16 * MS * WM;
    constexpr int B_FLOATS = BK * BN;
    constexpr bool COLSWZ = (BN % 32 == 0);
    constexpr int B_INST = B_FLOATS / 256;         // 1 KiB wave-instructions per B tile
    constexpr int B_PER = (B_INST + WM - 1) / WM;  // per wave
    __shared__ __attribute__((aligned(16))) float smem[2 * B_FLOATS];  // the ONLY LDS object

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 15, g = lane >> 4;
    // block -> (M tile, column block).  default: column block = b % NSPLIT (an XCD sees one W
    // slice).  GROUP: the NSPLIT column blocks of an M tile are consecutive slots of ONE XCD
    // (b % 8), so the A rows are fetched from HBM once and shared through that XCD's L2.
    int nb, mt;
    if (GROUP) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        nb = slot % NSPLIT;
        mt = (slot / NSPLIT) * 8 + xcd;
    } else {
        nb = (NSPLIT > 1) ? (int)(blockIdx.x % NSPLIT) : 0;
        mt = (int)(blockIdx.x / NSPLIT);
    }
    if (mt * BM >= M) return;
    const int m0 = mt * BM + wv * 16 * MS;
    const int n0 = nb * BN;

    f32x4 acc[MS][NS];
#pragma unroll
    for (int i = 0; i < MS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* a_src[MS];  // rows beyond M clamp to M-1: loaded, never stored
#pragma unroll
    for (int i = 0; i < MS; ++i) a_src[i] = A + (size_t)min(m0 + i * 16 + col, M - 1) * lda + 4 * g;
    const float* b_src[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int e = (wv + i * WM) * 64 + lane;  // float4 index in the linear tile
        int row = (e / (BN / 4)) % BK;            // LDS row position
        int c4 = e % (BN / 4);
        if (COLSWZ) c4 ^= ((row >> 2) & 1) << 2;
        else row ^= (row >> 2) & 1;               // involution: position p holds global row p ^ ((p>>2)&1)
        b_src[i] = W + (size_t)row * LDW + n0 + c4 * 4;
    }
    // read side: rows 16s + 4g + e have ((row>>2)&1) == (g&1)
    int bcol[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) bcol[j] = (j * 16 + col) ^ (COLSWZ ? ((g & 1) << 4) : 0);
    int brow[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) brow[e] = (COLSWZ ? e : (e ^ (g & 1))) * BN;

#define DENSE_B_ISSUE(kc, buf)                                                                        \
    _Pragma("unroll") for (int i = 0; i < B_PER; ++i) {                                               \
        if ((i + 1) * WM <= B_INST || wv + i * WM < B_INST)                                           \
            __builtin_amdgcn_global_load_lds((glb_void*)(b_src[i] + (size_t)(kc) * BK * LDW),         \
                                             (lds_void*)(smem + (buf) * B_FLOATS + (wv + i * WM) * 256), 16, 0, 0); \
    }

    const int nk = K / BK;
    float4 a_cur[NSUB][MS], a_nxt[NSUB][MS];
    DENSE_B_ISSUE(0, 0);
#pragma unroll
    for (int u = 0; u < NSUB; ++u)
#pragma unroll
        for (int i = 0; i < MS; ++i) a_cur[u][i] = *reinterpret_cast<const float4*>(a_src[i] + 16 * u);
    __syncthreads();  // hipcc drains vmcnt here (LDS-DMA pending): chunk 0 has landed
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) {  // both in flight during the MFMAs below
            if (!(ABL & 1)) { DENSE_B_ISSUE(kc + 1, buf ^ 1); }
#pragma unroll
            for (int u = 0; u < NSUB; ++u)
#pragma unroll
                for (int i = 0; i < MS; ++i)
                    a_nxt[u][i] = (ABL & 2) ? a_cur[u][i]
                                            : *reinterpret_cast<const float4*>(a_src[i] + (size_t)(kc + 1) * BK + 16 * u);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            const float* bs = smem + buf * B_FLOATS + (16 * u + 4 * g) * BN;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float b[NS];
#pragma unroll
                for (int j = 0; j < NS; ++j) b[j] = (ABL & 8) ? (float)(j + e) : bs[brow[e] + bcol[j]];
#pragma unroll
                for (int i = 0; i < MS; ++i) {
                    const float4 av = a_cur[u][i];
                    const float a = (e == 0) ? av.x : (e == 1) ? av.y : (e == 2) ? av.z : av.w;
#pragma unroll
                    for (int j = 0; j < NS; ++j) acc[i][j] = MFMA16(a, b[j], acc[i][j]);
                }
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int u = 0; u < NSUB; ++u)
#pragma unroll
            for (int i = 0; i < MS; ++i) a_cur[u][i] = a_nxt[u][i];
        if (!(ABL & 4)) __syncthreads();
    }
#undef DENSE_B_ISSUE

    // epilogue: C layout row = 4g + r, col = lane & 15
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int n = n0 + j * 16 + col;
        const float bv = bias[n];
        float wq = 0.f;
        if (QP) wq = W[(size_t)K * LDW + n];
#pragma unroll
        for (int i = 0; i < MS; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + i * 16 + 4 * g + r;
                float v = acc[i][j][r];
                if (QP) v = fmaf(qn, wq, v);
                v = lrelu_d(v + bv);
                if (m < M) out[(size_t)m * ldo + n] = v;
            }
    }
}

template <int MS, int NS, int WM, int NSPLIT, bool QP, bool GROUP = false, int ABL = 0, int NSUB = 1, int PRIO = 0>
static void launch_dense(const float* A, int lda, int K, const float* W, const float* bias, float qn, float* out,
                         int ldo, int M, hipStream_t s) {
    constexpr int BM = 16 * MS * WM;
    const int mtiles = (M + BM - 1) / BM;
    const int blocks = GROUP ? ((mtiles + 7) / 8) * 8 * NSPLIT : mtiles * NSPLIT;
    hipLaunchKernelGGL((k_dense<MS, NS, WM, NSPLIT, QP, GROUP, ABL, NSUB, PRIO>), dim3(blocks), dim3(64 * WM), 0, s, A,
                       lda, K, W, bias, qn, out, ldo, M);
}

static int fc1_variant() {
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("ETHCNN_FC1_VARIANT");  // development knob: tile-shape A/B runs
        v = e ? atoi(e) : -1;
    }
    return v;
}

// Tile shape by batch size.  Two shapes sit on the same plateau for large N (profiles/
// r01_fc1_variants.txt): 64 x 112 (N split 4) and 64 x 64 (N split 7, BK 32, ~2.5 % lower).  What
// differs is how evenly the blocks divide over the 256 CUs when there are only a handful per CU
// (every block runs the full K loop, so a CU with 7 blocks finishes 1/6 later than one with 6).
static int fc1_auto_variant(int n) {
    const int tiles = (n + 63) / 64;
    auto balance = [](int blocks) { return (double)blocks / (256.0 * ((blocks + 255) / 256)); };
    const double e112 = 1.0 * balance(tiles * 4), e64 = 0.975 * balance(tiles * 7);
    return e64 > e112 ? 43 : 1;
}

void launch_fc1(const Workspace& ws, const DeviceWeights& w, int n, float* out, hipStream_t s) {
    const float* A = ws.feat;
    int variant = fc1_variant();
    if (variant < 0) variant = fc1_auto_variant(n);
    switch (variant) {
        case 0:  // BM 128 (4 waves x 32 rows) x BN 112, N split 4
            launch_dense<2, 7, 4, 4, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        default:  // 1: BM 64 (4 waves x 16 rows) x BN 112, N split 4 -- measured best (profiles/r01_fc1_variants.txt)
            launch_dense<1, 7, 4, 4, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 2:  // BM 64 (2 waves x 32 rows)
            launch_dense<2, 7, 2, 4, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 3:  // BM 256 (4 waves x 64 rows)
            launch_dense<4, 7, 4, 4, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 4:  // BM 256 (8 waves x 32 rows)
            launch_dense<2, 7, 8, 4, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 5:  // BM 96 (2 waves x 48 rows)
            launch_dense<3, 7, 2, 4, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 6:  // BN 224 (N split 2), BM 64 (4 waves x 16 rows)
            launch_dense<1, 14, 4, 2, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 7:  // BM 128 x BN 224, 8 waves x 16 rows
            launch_dense<1, 14, 8, 2, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 8:  // BM 128 x BN 224, 4 waves x 32 rows
            launch_dense<2, 14, 4, 2, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 9:  // BM 128 x BN 448, 8 waves x 16 rows
            launch_dense<1, 28, 8, 1, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 10:  // BM 128 x BN 112, 8 waves x 16 rows
            launch_dense<1, 7, 8, 4, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 11:  // BM 256 x BN 112, 8 waves x 32 rows
            launch_dense<2, 7, 8, 4, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 101: launch_dense<1, 7, 4, 4, false, false, 1>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;   // ablations of variant 1 (timing only)
        case 102: launch_dense<1, 7, 4, 4, false, false, 2>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;
        case 103: launch_dense<1, 7, 4, 4, false, false, 3>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;
        case 107: launch_dense<1, 7, 4, 4, false, false, 7>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;
        case 108: launch_dense<1, 7, 4, 4, false, false, 8>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;
        case 115: launch_dense<1, 7, 4, 4, false, false, 15>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;
        case 104: launch_dense<1, 7, 4, 4, false, false, 4>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;
        case 21: launch_dense<1, 7, 4, 4, false, false, 0, 2>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;  // BK 32
        case 22: launch_dense<1, 7, 4, 4, false, false, 0, 4>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;  // BK 64
        case 23: launch_dense<1, 7, 4, 4, false, false, 0, 1, 1>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;  // setprio
        case 24: launch_dense<2, 7, 4, 4, false, false, 0, 2>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;  // BM128 BK 32
        case 25: launch_dense<1, 7, 4, 4, false, false, 0, 2, 1>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;  // BK 32 + setprio
        case 26: launch_dense<1, 7, 8, 4, false, false, 0, 2>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;  // 8 waves BK 32
        case 31: launch_dense<1, 7, 2, 4, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;  // BM 32
        case 32: launch_dense<1, 7, 3, 4, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;  // BM 48
        case 33: launch_dense<1, 7, 2, 4, false, false, 0, 2>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;  // BM 32 BK 32
        case 34: launch_dense<1, 7, 1, 4, false, false, 0, 2>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;  // BM 16 (1 wave) BK 32
        case 35: launch_dense<1, 7, 6, 4, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;  // BM 96
        case 41: launch_dense<1, 4, 4, 7, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;  // BM 64 x BN 64, N split 7
        case 42: launch_dense<2, 4, 4, 7, false>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;  // BM 128 x BN 64
        case 43: launch_dense<1, 4, 4, 7, false, false, 0, 2>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s); break;  // BM 64 x BN 64 BK 32
        case 12:  // variant 1 with the column blocks of an M tile grouped on one XCD
            launch_dense<1, 7, 4, 4, false, true>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 13:  // variant 0 grouped
            launch_dense<2, 7, 4, 4, false, true>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
        case 14:  // variant 10 grouped
            launch_dense<1, 7, 8, 4, false, true>(A, kNFeat, kNFeat, w.fc1_w, w.fc1_b, 0.f, out, kNVec, n, s);
            break;
    }
}

}  // namespace ethcnn
