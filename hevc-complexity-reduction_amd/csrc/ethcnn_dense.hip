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

namespace ethcnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ---- k_fc1_p3.  Everything arrives by LDS-DMA (W1 chunk shared by the block, each wave's own
// 1 KiB feature sub-chunks: read back with one conflict-free ds_read_b128), 3 LDS stages,
// prefetch distance 2.
// All VMEM operations of a wave are DMA instructions issued in a fixed number per iteration, so
// the data of chunk kc+1 is known to have landed when `vmcnt` has drained down to this
// iteration's own issue count: a COUNTED s_waitcnt, never vmcnt(0), and a raw s_barrier (a
// __syncthreads() would drain the queue, ROCm 7.2).  WAR: stage (kc+3)%3 == kc%3 is refilled in
// iteration kc+1, after every wave has passed the barrier that ends iteration kc.
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most `keep` (0..MAXK) groups of ISSUE DMA instructions are still in flight
template <int ISSUE, int MAXK>
__device__ __forceinline__ void vm_wait_groups(int keep) {
    if constexpr (MAXK == 0) {
        vm_wait<0>();
    } else {
        if (keep >= MAXK) vm_wait<MAXK * ISSUE>();
        else vm_wait_groups<ISSUE, MAXK - 1>(keep);
    }
}

// NST = LDS stages (prefetch distance NST - 1).  3 everywhere: deeper rings (4..6 stages, distance up to 5) were measured on
// the short-range shapes, which run one block per CU or less, and change nothing (profiles/r02_fc1_variants.txt section 5):
// those launches are bound by the per-chunk barrier + ds_read -> MFMA latency of a lone wave per SIMD, not by DMA latency.
template <int MS, int NS, int WM, int NSUB, int NST>
struct Fc1Shape {
    static constexpr int B_FLOATS = 16 * NSUB * 16 * NS;
    static constexpr int STAGE = B_FLOATS + WM * NSUB * MS * 256;
    static constexpr int LDS_FLOATS = NST * STAGE;
};

// one output tile (block `bid` of a grid of this shape over M rows); smem: >= Fc1Shape<...>::LDS_FLOATS floats
template <int MS, int NS, int WM, int NSUB, bool GROUP, int NST>
__device__ __forceinline__ void fc1_tile(float* __restrict__ smem, const float* __restrict__ feat, const float* __restrict__ Wimg,
                                         const float* __restrict__ bias, float* __restrict__ out, int M, const unsigned bid) {
    constexpr int BK = 16 * NSUB, BN = 16 * NS, NSPLIT = kNVec / BN, BM = 16 * MS * WM;
    constexpr int NK = kNFeat / BK;
    constexpr int B_FLOATS = BK * BN;
    constexpr bool COLSWZ = (BN % 32 == 0);
    constexpr int B_INST = B_FLOATS / 256;
    constexpr int B_PER = (B_INST + WM - 1) / WM;   // every wave issues exactly B_PER (tail duplicates the last piece)
    constexpr int A_PER = NSUB * MS;                // 1 KiB feature pieces per wave per chunk
    constexpr int ISSUE = B_PER + A_PER;            // VMEM ops per wave per iteration
    constexpr int A_FLOATS = WM * A_PER * 256;
    constexpr int STAGE = B_FLOATS + A_FLOATS;
    constexpr int DIST = NST - 1;
    static_assert(NK % NST == 0 && NK >= NST, "K chunks must come in whole rounds of the stage ring");
    static_assert(DIST * ISSUE <= 63, "vmcnt is a 6-bit counter");
    static_assert(STAGE == Fc1Shape<MS, NS, WM, NSUB, NST>::STAGE, "LDS budget of the shape");

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 15, g = lane >> 4;
    // block -> (M tile, column block).  Spread: column block = b % NSPLIT, an XCD (b % 8) streams one
    // slice of W1 but every M tile's features are fetched by NSPLIT different XCDs.  GROUP: the
    // NSPLIT column blocks of an M tile are consecutive slots of ONE XCD, so its features come
    // from HBM once and are shared through that XCD's L2 (W1 is then swept whole per XCD).
    int nb, mt;
    if (GROUP) {
        const int xcd = bid & 7, slot = bid >> 3;
        nb = slot % NSPLIT;
        mt = (slot / NSPLIT) * 8 + xcd;
        if (mt * BM >= M) return;  // whole block leaves before any barrier
    } else {
        nb = (int)(bid % NSPLIT);
        mt = (int)(bid / NSPLIT);
    }
    const int m0 = mt * BM + wv * 16 * MS;
    const int n0 = nb * BN;

    f32x4 acc[MS][NS];
#pragma unroll
    for (int i = 0; i < MS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ngroups = (M + 15) >> 4;
    // DMA addressing: every source is (wave-uniform 64-bit base in SGPRs) + (lane * 16 B in ONE VGPR), the
    // `global_load_lds_dwordx4 vOffset, s[base:base+1]` form.  The per-chunk advance is scalar arithmetic (free beside
    // MFMAs); the 64-bit per-lane addresses of the `vAddr, off` form cost two v_lshl_add_u64 per DMA and two VGPR
    // address reads -- VALU / VGPR-port time that comes straight out of the matrix pipe on gfx950
    // (profiles/r01_ubench_gfx950_issue_costs.txt).  LDS destinations are wave-uniform too (M0).
    const unsigned wvu = __builtin_amdgcn_readfirstlane(wv);
    const unsigned lane16 = (unsigned)lane * 16u;
    const int mg0 = __builtin_amdgcn_readfirstlane(m0 >> 4);
    const float* a_base[MS];  // this wave's group images [k/4][16][4]: 1 KiB per 16-k sub-chunk
#pragma unroll
    for (int i = 0; i < MS; ++i) a_base[i] = feat + (size_t)min(mg0 + i, ngroups - 1) * (kNFeat / 4) * 64;
    const float* b_base[B_PER];
    unsigned b_dst[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const unsigned piece = min(wvu + i * WM, (unsigned)(B_INST - 1));
        b_base[i] = Wimg + (size_t)nb * NK * B_FLOATS + (size_t)piece * 256;
        b_dst[i] = piece * 1024u;
    }
    int bcol[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) bcol[j] = (j * 16 + col) ^ (COLSWZ ? ((g & 1) << 4) : 0);
    int brow[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) brow[e] = (COLSWZ ? e : (e ^ (g & 1))) * BN;
    float* a_lds = smem + B_FLOATS + wv * A_PER * 256;  // + stage * STAGE

    // LDS-DMA through inline asm: hipcc does not model it, so it neither drains it at an LDS read
    // (as it does for the builtin: a vmcnt(0) before the first ds_read) nor counts it -- the
    // counted waits below are the only ordering.  M0 = wave-uniform LDS byte address, written in
    // the same statement that uses it (guide 5.7).
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)smem);
    const unsigned a_dst0 = lds_base + 4u * (B_FLOATS + wvu * A_PER * 256);
#define P3_DMA(sbase, lds_byte_addr)                                                                   \
    {                                                                                                  \
        unsigned keep_;                                                                                \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(lane16), "s"(sbase), "s"(lds_byte_addr) : "memory");          \
    }
#define P3_ISSUE(kc, st)                                                                               \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < B_PER; ++i)                                              \
            P3_DMA(b_base[i] + (size_t)(kc) * B_FLOATS, lds_base + 4u * (st) * STAGE + b_dst[i]);      \
        _Pragma("unroll") for (int u = 0; u < NSUB; ++u)                                               \
            _Pragma("unroll") for (int i = 0; i < MS; ++i)                                             \
                P3_DMA(a_base[i] + ((size_t)(kc) * NSUB + u) * 256,                                    \
                       a_dst0 + 4u * ((st) * STAGE + (u * MS + i) * 256));                             \
    }
#define P3_COMPUTE(st)                                                                                 \
    {                                                                                                  \
        _Pragma("unroll") for (int u = 0; u < NSUB; ++u) {                                             \
            float4 av[MS];                                                                             \
            _Pragma("unroll") for (int i = 0; i < MS; ++i)                                             \
                av[i] = *reinterpret_cast<const float4*>(a_lds + (st) * STAGE + (u * MS + i) * 256 + lane * 4); \
            const float* bs = smem + (st) * STAGE + (16 * u + 4 * g) * BN;                             \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                            \
                float b[NS];                                                                           \
                _Pragma("unroll") for (int j = 0; j < NS; ++j) b[j] = bs[brow[e] + bcol[j]];           \
                _Pragma("unroll") for (int i = 0; i < MS; ++i) {                                       \
                    const float a = (e == 0) ? av[i].x : (e == 1) ? av[i].y : (e == 2) ? av[i].z : av[i].w; \
                    _Pragma("unroll") for (int j = 0; j < NS; ++j) acc[i][j] = MFMA16(a, b[j], acc[i][j]); \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
    }
    // iteration kc: refill the stage consumed in iteration kc-1 with chunk kc+DIST, compute chunk kc, then make sure chunk
    // kc+1 has landed: of the chunks issued so far only the newest min(DIST-1, NK-2-kc) may still be in flight
#define P3_STEP(kc, st)                                                                                \
    {                                                                                                  \
        if ((kc) + DIST < NK) { P3_ISSUE((kc) + DIST, ((st) + DIST) % NST); }                          \
        P3_COMPUTE(st);                                                                                \
        vm_wait_groups<ISSUE, DIST - 1>(NK - 2 - (kc));                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
        __builtin_amdgcn_s_barrier();                                                                  \
    }

    // wins the issue arbitration against the co-resident wave of the next pass's tile stage (priority 0), which then fills
    // the gaps this kernel leaves instead of taking slots from it: FC1 inside the timed region 1.712 -> 1.686 ms, the step
    // +0.2 % (the tile stage, now slower, becomes what the next trunk waits for); priority 3 measures the same
    __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int c0 = 0; c0 < DIST; ++c0) { P3_ISSUE(c0, c0); }
    vm_wait<(DIST - 1) * ISSUE>();  // chunk 0 landed (the younger ones may still be in flight)
    __builtin_amdgcn_s_barrier();
    for (int kc = 0; kc < NK; kc += NST) {
#pragma unroll
        for (int st = 0; st < NST; ++st) { P3_STEP(kc + st, st); }
    }
#undef P3_DMA
#undef P3_ISSUE
#undef P3_COMPUTE
#undef P3_STEP

    // epilogue: bias + leaky-ReLU, one buffer_store per value: SGPR resource + uniform column offset (soffset) + one
    // VGPR offset holding the ROW part.  The hardware range check covers voffset only (the SGPR offset is not part of
    // it), and the resource ends at row M: rows of a ragged last tile are dropped by the check, no exec masking.
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(out, 0, M * kNVec * 4, 0x00020000);
    const int lane_out = ((m0 + 4 * g) * kNVec + col) * 4;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const float bv = bias[n0 + j * 16 + col];
#pragma unroll
        for (int i = 0; i < MS; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float h = acc[i][j][r] + bv;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(0.2f * h, h)), rO,
                                                      lane_out + (i * 16 + r) * kNVec * 4, (n0 + j * 16) * 4, 0);
            }
    }
}

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

template <int NS, int WM, int NSUB>
__global__ __launch_bounds__(64 * WM) void k_fc1_bulk(const float* __restrict__ feat, const float* __restrict__ Wimg,
                                                      const float* __restrict__ bias, float* __restrict__ out, int m_main,
                                                      int m_total, unsigned rem_blocks) {
    __shared__ __attribute__((aligned(16))) float smem[Fc1Shape<2, NS, WM, NSUB, 3>::LDS_FLOATS];  // the ONLY LDS object
    FC1_STAMP(0);
    if (blockIdx.x < rem_blocks)
        fc1_tile<1, NS, WM, NSUB, true, 3>(smem, feat + (size_t)(m_main / 16) * kNFeat * 16, Wimg, bias, out + (size_t)m_main * kNVec,
                                           m_total - m_main, blockIdx.x);
    else
        fc1_tile<2, NS, WM, NSUB, true, 3>(smem, feat, Wimg, bias, out, m_main, blockIdx.x - rem_blocks);
    FC1_STAMP(1);
}

template <int MS, int NS, int WM, int NSUB, bool GROUP = false, int NST = 3>
static void launch_fc1_p3(const float* feat, const float* wimg, const float* bias, float* out, int M, hipStream_t s) {
    constexpr int BM = 16 * MS * WM, NSPLIT = kNVec / (16 * NS);
    const int mtiles = (M + BM - 1) / BM;
    const int blocks = GROUP ? ((mtiles + 7) / 8) * 8 * NSPLIT : mtiles * NSPLIT;
    hipLaunchKernelGGL((k_fc1_p3<MS, NS, WM, NSUB, GROUP, NST>), dim3(blocks), dim3(64 * WM), 0, s, feat, wimg, bias, out, M);
}

static int fc1_variant() {
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("ETHCNN_FC1_VARIANT");  // development knob: tile-shape A/B runs
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
