// ethcnn_fc1_fast.hip -- FC1 of plans 2 and 3 (opt-in, ethcnn_set_fc1_plan): h1[N,448] = lrelu(feat[N,2688] . W1 + b1)
// (net_CNN.py:156,164,177) on the 16-BIT matrix pipe of gfx950 (v_mfma_f32_32x32x16_f16: 16x the rate of the exact-fp32
// MFMA), with fp32 operands carried as SPLIT 16-bit pieces and fp32 accumulation.
//
// Arithmetic: fp16 x 2.  Features and weights are scaled by powers of two (exact; chosen at weight load so that no piece can
//           overflow: fast_feature_bound) and carried as two fp16 pieces, a s = h0 + h1 to 2^-24 relative (two 11-bit significands,
//           round to nearest even).  Three products: h0 g0, h1 g0, h0 g1 (the dropped h1 g1 is below 2^-22 |a w|); the result is
//           scaled back in the epilogue (exact).
// (Round 4 also shipped "plan 1", exact three-way bf16 splits with six products: slower than this form in every metric -- 47 vs 64
// M CTU/s on C3 -- and no more accurate, rms error vs float64 3.7e-7 against 2.6e-7; removed in round 5, the probe stays:
// scripts/ubench/bf16x3_probe.hip -> profiles/r04_bf16x3_probe.txt.)
// Not narrower arithmetic in effect: measured against float64 the sum is as accurate as the exact-fp32 fmaf chain of plan 0 (K =
// 2688: rms error fp32 chain 4.1e-7, fp16 x 2 2.6e-7, unchanged while the operand scale is moved over 12 octaves) -- the error of a
// long fp32 sum is dominated by the roundings of the accumulation, not by 2^-24 representation errors of the terms.  What changes is
// the ORDER of the fp32 additions (16 products are summed inside one MFMA; undocumented), so results are not bit-identical to plan
// 0 / the oracle: the plan is held to the north star's 1e-4 and is never the default.
//
// Shape.  K is walked in 168 chunks of 16 (one MFMA k step).  A block = 8 waves = 256 CTUs (a wave owns one row tile of 32) x
// NS = 7 column tiles of 32; per chunk a wave issues NS x NPROD MFMAs on NS accumulator tiles (16 registers each).  Operands
// arrive by LDS-DMA in exactly the order the MFMA wants them (1 KiB per (tile, chunk, piece), lane l = bytes 16 l .. 16 l + 15:
// one conflict-free ds_read_b128 each):
//   A  featb[pair of groups][chunk][piece][k half][row][8]   written by the trunk in the plan's form
//   B  fc1_fast[chunk][column tile][piece][k half][col][8]   packed at weight upload (pack_fc1_fast_image)
// through a 3-stage ring.  PING-PONG: a lock-step form (all eight waves issue DMA, read LDS and queue MFMAs in the same phases:
// the pipe idles while everybody waits for LDS) was measured first: matrix pipe 74 % busy inside a round.  Here the two waves of
// a SIMD (w and w + 4: a block's waves go to the SIMDs round-robin) work in OPPOSITE phases, one phase behind each other:
//      phase 2k      waves 0..3: LOAD(k)       waves 4..7: COMPUTE(k - 1)
//      phase 2k + 1  waves 0..3: COMPUTE(k)    waves 4..7: LOAD(k)
// LOAD(k) = issue this wave's share of chunk k + 2's DMA, read ALL of chunk k's operands into registers (NP (1 + NS) fragments),
// wait for them and for every DMA group but the newest; COMPUTE(k) = NPROD x NS back-to-back MFMAs, products outer / column
// tiles inner (NS independent accumulators between two uses of one).  One s_barrier per phase: pipe 91 % busy inside a round.
// Stage reuse: chunk k + 2 lands in the stage chunk k - 1 used, last read in phase 2k - 1, and is first issued in phase 2k; it is
// first read in phase 2k + 4, and every share of it has been waited for by the end of phase 2k + 3.
//
// What bounds it.  Not the issue rate: at this MFMA density the chip lowers its shader clock (~2.0 GHz at the 1400 W cap against
// 2.38 GHz under plan 0: scripts/power_probe.py -> profiles/r04_power_probe.txt), and keeps it low for the neighbouring kernels
// of the step.  Fewer MFMAs per CTU is what moves the step, not a denser stream.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ethcnn_fc1_tile.h"
#include "ethcnn_kernels.h"

namespace ethcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kFastColTiles = kNVec / 32;  // 14

template <int PLAN> struct FastPlan;
template <> struct FastPlan<2> {
    static constexpr int NP = 2, NPROD = 3;
    using frag = f16x8;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static constexpr int PA[3] = {0, 1, 0}, PB[3] = {0, 0, 1};
};

// NINE: the stage also holds a NINTH row tile (below): one more A record per chunk
template <int PLAN, int NS, bool NINE = false>
struct FastShape {
    static constexpr int WM = 8, NST = 3, NP = FastPlan<PLAN>::NP;
    static constexpr int ROWS = WM + (NINE ? 1 : 0);
    static constexpr int PIECES = NP * (ROWS + NS);        // 1 KiB pieces per stage
    static constexpr int PER = (PIECES + WM - 1) / WM;     // DMA instructions per wave per chunk (the tail repeats the last piece)
    static constexpr int STAGE = PIECES * 1024;
    static constexpr int LDS_BYTES = NST * STAGE;          // 90 KB (96 KB with the ninth row tile)
};

// M tiles of a launch.  k_fc1_fast keeps ONE block per CU, so a grid runs in rounds of `cus` blocks, and C3's 3188 row tiles of 32
// CTUs made 399 M tiles x 2 column halves = 798 blocks = 3.12 rounds: a fourth round for 4 % of the work (20 % of the stage).
// NINE: the number of M tiles is cut to whole rounds (tiles = a multiple of cus / 2) and the row tiles left over -- fewer than
// one per M tile -- are handed out one each to the first `extra` M tiles as a NINTH row tile, which the block's waves 0..6 share:
// wave w multiplies it with column tile w (whose B fragments it holds anyway): 3 more MFMAs per chunk on one more accumulator.
// Those blocks take 8/7 of a block time; the launch ends after 3 rounds + 1/7 instead of 4.
struct FastTiles {
    int tiles, extra;  // M tile t owns row tiles [8 t + min(t, extra), ... + 8 + (t < extra))
};
static inline FastTiles fast_tiles(int npairs, int cus, bool allow_nine) {
    const int per_round = cus / 2;
    const int whole = per_round > 0 ? (npairs / 8 / per_round) * per_round : 0;
    if (allow_nine && whole > 0 && npairs - 8 * whole <= whole) return {whole, npairs - 8 * whole};
    return {(npairs + 7) / 8, 0};
}

template <int PLAN, int NS, bool NINE>
__device__ __forceinline__ void fc1_fast_tile(char* __restrict__ smem, const char* __restrict__ featb, const char* __restrict__ Wf,
                                              const float* __restrict__ bias, float* __restrict__ out, int M, float unscale, const int mt,
                                              const int nb, const int extra) {
    using P = FastPlan<PLAN>;
    using S = FastShape<PLAN, NS, NINE>;
    using frag = typename P::frag;
    constexpr int WM = S::WM, NST = S::NST, NP = P::NP, NK = kFastChunks, PER = S::PER, STAGE = S::STAGE;
    static_assert(NST == 3 && NK % NST == 0, "three stages: chunk k + 2 reuses the stage of chunk k - 1");
    static_assert(2 * PER <= 63, "vmcnt is a 6-bit counter");
    static_assert(kFastColTiles % NS == 0, "column tiles per block must divide 14");

    const int lane = threadIdx.x & 63;
    const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool late = wv >= 4;
    const int npairs = (M + 31) >> 5;
    const int pair0 = mt * WM + (NINE ? min(mt, extra) : 0);
    const bool has9 = NINE && mt < extra;                    // block-uniform: this M tile owns a ninth row tile
    const bool mine9 = has9 && wv < (unsigned)NS;            // wave w < 7 multiplies it with column tile w
    static_assert(!NINE || NS == 7, "the ninth row tile is shared by seven waves, one column tile each");

    f32x16 acc[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // DMA plan of this wave: pieces q = wv + i WM of the stage image [A: WM x NP][B: NS x NP]; every source is a wave-uniform
    // 64-bit base (SGPRs, advanced by SALU) + lane * 16 in one VGPR (ethcnn_fc1_tile.h)
    const char* src[PER];
    unsigned dst[PER], step[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const unsigned q = min(wv + (unsigned)i * WM, (unsigned)(S::PIECES - 1));
        if (q < (unsigned)(NP * S::ROWS)) {
            const unsigned rt = q / NP, p = q - NP * rt;  // (row tile 8 of a block without one repeats a neighbour: read by nobody)
            const int pr = min(pair0 + (int)rt, npairs - 1);  // pairs beyond the pass repeat the last one (their rows are never stored)
            src[i] = featb + (size_t)pr * fast_pair_bytes(PLAN) + p * 1024u;
            step[i] = NP * 1024u;
        } else {
            src[i] = Wf + (size_t)(nb * NS) * (NP * 1024) + (q - NP * S::ROWS) * 1024u;
            step[i] = kFastColTiles * NP * 1024u;
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
    const char* a_lds = smem + (NP * wv) * 1024 + lane * 16;
    const char* b_lds = smem + (NP * S::ROWS) * 1024 + lane * 16;
    const char* a9_lds = smem + (NP * WM) * 1024 + lane * 16;   // the ninth row tile's A fragments
    const char* b9_lds = b_lds + (NP * wv) * 1024;              // column tile wv once more (its own registers: wave-uniform index)
    f32x16 acc9;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc9[r] = 0.f;

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
            frag a[NP], b[NS][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) a[p] = *reinterpret_cast<const frag*>(a_lds + st * STAGE + p * 1024);
#pragma unroll
            for (int j = 0; j < NS; ++j)
#pragma unroll
                for (int p = 0; p < NP; ++p) b[j][p] = *reinterpret_cast<const frag*>(b_lds + st * STAGE + (NP * j + p) * 1024);
            frag a9[NP], b9[NP];
            if (NINE && mine9) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    a9[p] = *reinterpret_cast<const frag*>(a9_lds + st * STAGE + p * 1024);
                    b9[p] = *reinterpret_cast<const frag*>(b9_lds + st * STAGE + p * 1024);
                }
            }
            if (k + 2 < NK) vm_wait<PER>(); else vm_wait<0>();  // everything but the newest DMA group of this wave has landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PP_BARRIER();
            // ---- COMPUTE(k): products outer (big terms first), column tiles inner
#pragma unroll
            for (int q = 0; q < P::NPROD; ++q)
#pragma unroll
                for (int j = 0; j < NS; ++j) acc[j] = P::mfma(a[P::PA[q]], b[j][P::PB[q]], acc[j]);
            if (NINE && mine9) {
#pragma unroll
                for (int q = 0; q < P::NPROD; ++q) acc9 = P::mfma(a9[P::PA[q]], b9[P::PB[q]], acc9);
            }
            PP_BARRIER();
        }
    }
    if (!late) PP_BARRIER();  // the last phase belongs to waves 4..7 alone
#undef PP_DMA
#undef PP_ISSUE
#undef PP_BARRIER

    // epilogue: (plan 2: scale back,) bias + leaky-ReLU.  C layout of the 32x32 tile: column = lane & 31, row = (r & 3) + 8 (r >> 2)
    // + 4 (lane >> 5).  One buffer_store per value (lanes 0..31 = 128 contiguous bytes of a row); the row part of the offset sits in
    // the VGPR, which is what the hardware range check covers: rows >= M of a ragged last tile are dropped by it (ethcnn_fc1_tile.h).
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(out, 0, M * kNVec * 4, 0x00020000);
    const int m0 = (pair0 + (int)wv) * 32, n0 = nb * NS * 32;
    const int lane_out = ((m0 + 4 * (lane >> 5)) * kNVec + (lane & 31)) * 4;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const float bv = bias[n0 + j * 32 + (lane & 31)];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float h = (PLAN == 2 ? acc[j][r] * unscale : acc[j][r]) + bv;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(0.2f * h, h)), rO,
                                                  lane_out + ((r & 3) + 8 * (r >> 2)) * kNVec * 4, (n0 + j * 32) * 4, 0);
        }
    }
    if (NINE && mine9) {  // the ninth row tile: rows of pair pair0 + 8, this wave's column tile
        const int m9 = (pair0 + WM) * 32;
        const int lane9 = ((m9 + 4 * (lane >> 5)) * kNVec + (lane & 31)) * 4;
        const float bv = bias[n0 + (int)wv * 32 + (lane & 31)];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float h = (PLAN == 2 ? acc9[r] * unscale : acc9[r]) + bv;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(0.2f * h, h)), rO,
                                                  lane9 + ((r & 3) + 8 * (r >> 2)) * kNVec * 4, (n0 + (int)wv * 32) * 4, 0);
        }
    }
}

template <int PLAN, int NS, bool NINE>
__global__ __launch_bounds__(512) void k_fc1_fast(const char* __restrict__ featb, const char* __restrict__ Wf,
                                                  const float* __restrict__ bias, float* __restrict__ out, int M, float unscale,
                                                  int tiles, int extra) {
    __shared__ __attribute__((aligned(16))) char smem[FastShape<PLAN, NS, NINE>::LDS_BYTES];  // the ONLY LDS object
    int mt, nb;
    fc1_block_to_tile<kFastColTiles / NS, true>(blockIdx.x, mt, nb);  // the column blocks of an M tile share one XCD's L2
    if (mt >= tiles) return;
    fc1_fast_tile<PLAN, NS, NINE>(smem, featb, Wf, bias, out, M, unscale, mt, nb, extra);
}

void launch_fc1_fast(const Workspace& ws, const DeviceWeights& w, int n, float* out, int plan, hipStream_t s, int cus) {
    const char* fb = reinterpret_cast<const char*>(ws.featb);
    const char* wf = reinterpret_cast<const char*>(w.fc1_fast);
    (void)plan;  // (2: the only 16-bit form of FC1; plan 3 uses it as well)
    const FastTiles ft = fast_tiles((n + 31) / 32, cus, true);
    const dim3 grid(((ft.tiles + 7) / 8) * 8 * 2);
    if (ft.extra > 0)
        hipLaunchKernelGGL((k_fc1_fast<2, 7, true>), grid, dim3(512), 0, s, fb, wf, w.fc1_b, out, n, 1.0f / (w.fast_scale_a * w.fast_scale_w), ft.tiles, ft.extra);
    else
        hipLaunchKernelGGL((k_fc1_fast<2, 7, false>), grid, dim3(512), 0, s, fb, wf, w.fc1_b, out, n, 1.0f / (w.fast_scale_a * w.fast_scale_w), ft.tiles, 0);
}

}  // namespace ethcnn
