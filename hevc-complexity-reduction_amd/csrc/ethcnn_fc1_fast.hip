// ethcnn_fc1_fast.hip -- FC1 plan 1 (opt-in, ethcnn_set_fc1_plan): h1[N,448] = lrelu(feat[N,2688] . W1 + b1)
// (net_CNN.py:156,164,177) on the BF16 matrix pipe of gfx950 (v_mfma_f32_32x32x16_bf16, 16x the rate of the exact-fp32 MFMA).
//
// Arithmetic.  Every fp32 feature a and weight w is carried as three bf16 pieces, a = a0 + a1 + a2 and w = w0 + w1 + w2
// EXACTLY (round to nearest even at each step; trunk epilogue ethcnn_trunk_task.h::store_pair_bf16x3, host
// ethcnn_weights.cpp::split_bf16x3).  A product of two bf16 values is exact in fp32, the MFMA accumulates in fp32, and
//      a w = sum over i, j of a_i w_j,   |a_i w_j| <= 2^(-9 (i + j)) |a w|:
// the six terms with i + j <= 2 are issued, the three dropped ones are below 2^-27 |a w| -- under half an ulp of the fp32
// product itself.  So this is NOT narrower arithmetic: measured against float64 the sums are as accurate as the exact-fp32 fmaf
// chain of plan 0 (scripts/ubench/bf16x3_probe.hip -> profiles/r04_bf16x3_probe.txt: rms error 3.7e-7 vs 4.1e-7 at K = 2688).
// What changes is the ORDER of the fp32 additions (16 products are summed inside one MFMA; undocumented), so the results are
// not bit-identical to plan 0 / the oracle; the plan is held to the north star's 1e-4 and is never the default.
//
// Shape.  K is walked in 168 chunks of 16 (one MFMA k step); a block owns WM row tiles of 32 CTUs (one wave each) x NS column
// tiles of 32; per chunk a wave issues NS x 6 MFMAs on NS accumulator tiles (16 registers each).  Operands arrive by LDS-DMA in
// exactly the order the MFMA wants them (1 KiB per (tile, chunk, piece), lane l = 16 bytes l: one conflict-free ds_read_b128):
//   A  featb[pair of groups][chunk][piece][k half][row][8]   written by the trunk (plan 1 form)
//   B  fc1_fast[chunk][column tile][piece][k half][col][8]   packed at weight upload
// An NST-stage ring, prefetch distance NST - 1, counted vmcnt + one raw s_barrier per chunk as in ethcnn_fc1_tile.h.
// Bytes staged per chunk: 3 KiB x (WM + NS); 256 x 224 tiles (WM 8, NS 7) need 17 B/clk/CU from the L2 at the full MFMA rate.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ethcnn_fc1_tile.h"
#include "ethcnn_kernels.h"

namespace ethcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA32B(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int kFastColTiles = kNVec / 32;  // 14

template <int WM, int NS, int NST>
struct FastShape {
    static constexpr int PIECES = 3 * (WM + NS);           // 1 KiB pieces per stage
    static constexpr int PER = (PIECES + WM - 1) / WM;     // DMA instructions per wave per chunk (the tail repeats the last piece)
    static constexpr int STAGE = PIECES * 1024;
    static constexpr int LDS_BYTES = NST * STAGE;
};

template <int WM, int NS, int NST>
__device__ __forceinline__ void fc1_fast_tile(char* __restrict__ smem, const char* __restrict__ featb, const char* __restrict__ Wf,
                                              const float* __restrict__ bias, float* __restrict__ out, int M, const int mt, const int nb) {
    using S = FastShape<WM, NS, NST>;
    constexpr int NK = kFastChunks, DIST = NST - 1, PER = S::PER, STAGE = S::STAGE;
    static_assert(NK % NST == 0 && NK >= NST, "K chunks must come in whole rounds of the stage ring");
    static_assert(DIST * PER <= 63, "vmcnt is a 6-bit counter");
    static_assert(kFastColTiles % NS == 0, "column tiles per block must divide 14");

    const int lane = threadIdx.x & 63;
    const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int npairs = (M + 31) >> 5;
    const int pair0 = mt * WM;

    f32x16 acc[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // DMA plan of this wave: pieces q = wv + i WM of the stage image [A: WM x 3][B: NS x 3]; every source is a wave-uniform
    // 64-bit base (SGPRs, advanced by SALU) + lane * 16 in one VGPR (ethcnn_fc1_tile.h)
    const char* src[PER];
    unsigned dst[PER], step[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const unsigned q = min(wv + (unsigned)i * WM, (unsigned)(S::PIECES - 1));
        if (q < 3u * WM) {
            const unsigned rt = q / 3u, p = q - 3u * rt;
            const int pr = min(pair0 + (int)rt, npairs - 1);  // pairs beyond the pass repeat the last one (their rows are never stored)
            src[i] = featb + (size_t)pr * kFastPairBytes + p * 1024u;
            step[i] = 3072u;
        } else {
            src[i] = Wf + (size_t)(nb * NS) * 3072 + (q - 3u * WM) * 1024u;
            step[i] = kFastColTiles * 3072u;
        }
        dst[i] = q * 1024u;
    }
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)smem);
#define FAST_DMA(sbase, lds_byte_addr)                                                                 \
    {                                                                                                  \
        unsigned keep_;                                                                                \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(lane16), "s"(sbase), "s"(lds_byte_addr) : "memory");          \
    }
#define FAST_ISSUE(kc, st)                                                                             \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < PER; ++i)                                                \
            FAST_DMA(src[i] + (size_t)(kc) * step[i], lds_base + (unsigned)((st) * STAGE) + dst[i]);   \
    }
    const char* a_lds = smem + (3 * wv) * 1024 + lane * 16;
    const char* b_lds = smem + (3 * WM) * 1024 + lane * 16;
#define FAST_COMPUTE(st)                                                                               \
    {                                                                                                  \
        bf16x8 a[3];                                                                                   \
        _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                  \
            a[p] = *reinterpret_cast<const bf16x8*>(a_lds + (st) * STAGE + p * 1024);                  \
        _Pragma("unroll") for (int j = 0; j < NS; ++j) {                                               \
            bf16x8 b[3];                                                                               \
            _Pragma("unroll") for (int p = 0; p < 3; ++p)                                              \
                b[p] = *reinterpret_cast<const bf16x8*>(b_lds + (st) * STAGE + (3 * j + p) * 1024);    \
            acc[j] = MFMA32B(a[0], b[0], acc[j]);                                                      \
            acc[j] = MFMA32B(a[1], b[0], acc[j]);                                                      \
            acc[j] = MFMA32B(a[0], b[1], acc[j]);                                                      \
            acc[j] = MFMA32B(a[2], b[0], acc[j]);                                                      \
            acc[j] = MFMA32B(a[1], b[1], acc[j]);                                                      \
            acc[j] = MFMA32B(a[0], b[2], acc[j]);                                                      \
        }                                                                                              \
    }
#define FAST_STEP(kc, st)                                                                              \
    {                                                                                                  \
        if ((kc) + DIST < NK) { FAST_ISSUE((kc) + DIST, ((st) + DIST) % NST); }                        \
        FAST_COMPUTE(st);                                                                              \
        vm_wait_groups<PER, DIST - 1>(NK - 2 - (kc));                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
        __builtin_amdgcn_s_barrier();                                                                  \
    }

#pragma unroll
    for (int c0 = 0; c0 < DIST; ++c0) { FAST_ISSUE(c0, c0); }
    vm_wait<(DIST - 1) * PER>();  // chunk 0 landed (the younger ones may still be in flight)
    __builtin_amdgcn_s_barrier();
    for (int kc = 0; kc < NK; kc += NST) {
#pragma unroll
        for (int st = 0; st < NST; ++st) { FAST_STEP(kc + st, st); }
    }
#undef FAST_DMA
#undef FAST_ISSUE
#undef FAST_COMPUTE
#undef FAST_STEP

    // epilogue: bias + leaky-ReLU.  C layout of the 32x32 tile: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    // One buffer_store per value (lanes 0..31 = 128 contiguous bytes of a row); the row part of the offset sits in the VGPR, which
    // is what the hardware range check covers: rows >= M of a ragged last tile are dropped by it (ethcnn_fc1_tile.h).
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(out, 0, M * kNVec * 4, 0x00020000);
    const int m0 = (pair0 + (int)wv) * 32, n0 = nb * NS * 32;
    const int lane_out = ((m0 + 4 * (lane >> 5)) * kNVec + (lane & 31)) * 4;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const float bv = bias[n0 + j * 32 + (lane & 31)];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float h = acc[j][r] + bv;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(0.2f * h, h)), rO,
                                                  lane_out + ((r & 3) + 8 * (r >> 2)) * kNVec * 4, (n0 + j * 32) * 4, 0);
        }
    }
}

// ---- the same tile in PING-PONG form (WM = 8: two waves per SIMD).  In the form above all eight waves of the block move in lock
// step: after every barrier they all issue DMA and wait for their LDS reads while the matrix pipe idles, then all queue MFMAs
// (measured: pipe 74 % busy inside a round).  Here the two waves of a SIMD (w and w + 4: a block's waves go to the SIMDs
// round-robin) work in OPPOSITE phases, one phase behind each other:
//      phase 2k      waves 0..3: LOAD(k)       waves 4..7: COMPUTE(k - 1)
//      phase 2k + 1  waves 0..3: COMPUTE(k)    waves 4..7: LOAD(k)
// LOAD(k) = issue this wave's share of chunk k + 2's DMA, read ALL of chunk k's operands into registers (3 A + 3 NS B fragments:
// 96 VGPRs at NS = 7), wait for them and for every DMA group but the newest; COMPUTE(k) = 6 NS back-to-back MFMAs, products
// outer / column tiles inner (NS independent accumulators between two uses of one).  One s_barrier per phase.  Stage reuse:
// chunk k + 2 lands in the stage chunk k - 1 used, last read in phase 2k - 1, and is first issued in phase 2k; it is first read
// in phase 2k + 4, and every share of it has been waited for by the end of phase 2k + 3.
template <int NS, int NST>
__device__ __forceinline__ void fc1_fast_tile_pp(char* __restrict__ smem, const char* __restrict__ featb, const char* __restrict__ Wf,
                                                 const float* __restrict__ bias, float* __restrict__ out, int M, const int mt, const int nb) {
    constexpr int WM = 8;
    using S = FastShape<WM, NS, NST>;
    constexpr int NK = kFastChunks, PER = S::PER, STAGE = S::STAGE;
    static_assert(NST == 3 && NK % NST == 0, "three stages: chunk k + 2 reuses the stage of chunk k - 1");
    static_assert(2 * PER <= 63, "vmcnt is a 6-bit counter");
    static_assert(kFastColTiles % NS == 0, "column tiles per block must divide 14");

    const int lane = threadIdx.x & 63;
    const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool late = wv >= 4;
    const int npairs = (M + 31) >> 5;
    const int pair0 = mt * WM;

    f32x16 acc[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const char* src[PER];
    unsigned dst[PER], step[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const unsigned q = min(wv + (unsigned)i * WM, (unsigned)(S::PIECES - 1));
        if (q < 3u * WM) {
            const unsigned rt = q / 3u, p = q - 3u * rt;
            const int pr = min(pair0 + (int)rt, npairs - 1);
            src[i] = featb + (size_t)pr * kFastPairBytes + p * 1024u;
            step[i] = 3072u;
        } else {
            src[i] = Wf + (size_t)(nb * NS) * 3072 + (q - 3u * WM) * 1024u;
            step[i] = kFastColTiles * 3072u;
        }
        dst[i] = q * 1024u;
    }
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)smem);
#define PP_DMA(sbase, lds_byte_addr)                                                                   \
    {                                                                                                  \
        unsigned keep_;                                                                                \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(lane16), "s"(sbase), "s"(lds_byte_addr) : "memory");          \
    }
#define PP_ISSUE(kc, st)                                                                               \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < PER; ++i)                                                \
            PP_DMA(src[i] + (size_t)(kc) * step[i], lds_base + (unsigned)((st) * STAGE) + dst[i]);     \
    }
    // (sched_barrier: MFMAs have no memory effects, so nothing else keeps hipcc from moving a whole COMPUTE across its barriers)
#define PP_BARRIER()                                                                                   \
    {                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        asm volatile("" ::: "memory");                                                                 \
        __builtin_amdgcn_s_barrier();                                                                  \
        asm volatile("" ::: "memory");                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    }
    const char* a_lds = smem + (3 * wv) * 1024 + lane * 16;
    const char* b_lds = smem + (3 * WM) * 1024 + lane * 16;

    PP_ISSUE(0, 0);
    PP_ISSUE(1, 1);
    vm_wait<0>();
    PP_BARRIER();
    if (late) PP_BARRIER();  // phase 0 belongs to waves 0..3 alone
    for (int kc = 0; kc < NK; kc += NST) {
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int k = kc + st;
            // ---- LOAD(k)
            if (k + 2 < NK) { PP_ISSUE(k + 2, (st + 2) % NST); }
            bf16x8 a[3], b[NS][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8*>(a_lds + st * STAGE + p * 1024);
#pragma unroll
            for (int j = 0; j < NS; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) b[j][p] = *reinterpret_cast<const bf16x8*>(b_lds + st * STAGE + (3 * j + p) * 1024);
            if (k + 2 < NK) vm_wait<PER>(); else vm_wait<0>();  // everything but the newest DMA group of this wave has landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PP_BARRIER();
            // ---- COMPUTE(k): six products, big terms first; NS independent accumulators inside each
#pragma unroll
            for (int j = 0; j < NS; ++j) acc[j] = MFMA32B(a[0], b[j][0], acc[j]);
#pragma unroll
            for (int j = 0; j < NS; ++j) acc[j] = MFMA32B(a[1], b[j][0], acc[j]);
#pragma unroll
            for (int j = 0; j < NS; ++j) acc[j] = MFMA32B(a[0], b[j][1], acc[j]);
#pragma unroll
            for (int j = 0; j < NS; ++j) acc[j] = MFMA32B(a[2], b[j][0], acc[j]);
#pragma unroll
            for (int j = 0; j < NS; ++j) acc[j] = MFMA32B(a[1], b[j][1], acc[j]);
#pragma unroll
            for (int j = 0; j < NS; ++j) acc[j] = MFMA32B(a[0], b[j][2], acc[j]);
            PP_BARRIER();
        }
    }
    if (!late) PP_BARRIER();  // the last phase belongs to waves 4..7 alone
#undef PP_DMA
#undef PP_ISSUE
#undef PP_BARRIER

    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(out, 0, M * kNVec * 4, 0x00020000);
    const int m0 = (pair0 + (int)wv) * 32, n0 = nb * NS * 32;
    const int lane_out = ((m0 + 4 * (lane >> 5)) * kNVec + (lane & 31)) * 4;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const float bv = bias[n0 + j * 32 + (lane & 31)];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float h = acc[j][r] + bv;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(0.2f * h, h)), rO,
                                                  lane_out + ((r & 3) + 8 * (r >> 2)) * kNVec * 4, (n0 + j * 32) * 4, 0);
        }
    }
}

template <int NS, int NST>
__global__ __launch_bounds__(512) void k_fc1_fast_pp(const char* __restrict__ featb, const char* __restrict__ Wf,
                                                     const float* __restrict__ bias, float* __restrict__ out, int M) {
    __shared__ __attribute__((aligned(16))) char smem[FastShape<8, NS, NST>::LDS_BYTES];  // the ONLY LDS object
    int mt, nb;
    fc1_block_to_tile<kFastColTiles / NS, true>(blockIdx.x, mt, nb);
    if (mt * 8 * 32 >= M) return;
    fc1_fast_tile_pp<NS, NST>(smem, featb, Wf, bias, out, M, mt, nb);
}

template <int WM, int NS, int NST>
__global__ __launch_bounds__(64 * WM) void k_fc1_fast(const char* __restrict__ featb, const char* __restrict__ Wf,
                                                      const float* __restrict__ bias, float* __restrict__ out, int M) {
    __shared__ __attribute__((aligned(16))) char smem[FastShape<WM, NS, NST>::LDS_BYTES];  // the ONLY LDS object
    int mt, nb;
    fc1_block_to_tile<kFastColTiles / NS, true>(blockIdx.x, mt, nb);  // the column blocks of an M tile share one XCD's L2
    if (mt * WM * 32 >= M) return;
    fc1_fast_tile<WM, NS, NST>(smem, featb, Wf, bias, out, M, mt, nb);
}

template <int WM, int NS, int NST>
static void launch_shape(const char* featb, const char* wf, const float* bias, float* out, int M, hipStream_t s) {
    constexpr int NSPLIT = kFastColTiles / NS;
    const int mtiles = ((M + 31) / 32 + WM - 1) / WM;
    hipLaunchKernelGGL((k_fc1_fast<WM, NS, NST>), dim3(((mtiles + 7) / 8) * 8 * NSPLIT), dim3(64 * WM), 0, s, featb, wf, bias, out, M);
}

void launch_fc1_fast(const Workspace& ws, const DeviceWeights& w, int n, float* out, hipStream_t s) {
    const char* fb = reinterpret_cast<const char*>(ws.featb);
    const char* wf = reinterpret_cast<const char*>(w.fc1_fast);
#ifdef ETHCNN_EXPERIMENTS
    static const int shape = [] { const char* e = getenv("ETHCNN_FC1_FAST_SHAPE"); return e ? atoi(e) : 0; }();
#else
    constexpr int shape = 0;
#endif
    switch (shape) {
        case 4: {  // 256 x 224, ping-pong
            const int mtiles = ((n + 31) / 32 + 7) / 8;
            hipLaunchKernelGGL((k_fc1_fast_pp<7, 3>), dim3(((mtiles + 7) / 8) * 8 * 2), dim3(512), 0, s, fb, wf, w.fc1_b, out, n);
            break;
        }
        default: launch_shape<8, 7, 3>(fb, wf, w.fc1_b, out, n, s); break;   // 256 x 224, 135 KB of LDS: one block per CU
        case 1: launch_shape<4, 7, 2>(fb, wf, w.fc1_b, out, n, s); break;    // 128 x 224, 66 KB: two blocks per CU
        case 2: launch_shape<4, 7, 3>(fb, wf, w.fc1_b, out, n, s); break;    // 128 x 224, 99 KB: one block per CU
        case 3: launch_shape<8, 7, 2>(fb, wf, w.fc1_b, out, n, s); break;    // 256 x 224, 90 KB
    }
}

}  // namespace ethcnn
