// ethcnn_fused.hip -- FC1 + the three heads + the batch gates of a big pass as ONE launch (net_CNN.py:156-187).
//
// Why: a pass used to be trunk | FC1 | heads | gate, four dependent launches.  FC1's grid (3328 blocks on 768 slots at C3)
// drains over ~250 us of half-empty CUs, the heads launch behind it has a ~60 us latency-bound tail of its own, and every
// kernel boundary costs ~10-15 us of idle GPU on the 8-XCD part (profiles/r02_fc1_timeline.txt, r02_heads_timeline.txt).
// Here the heads blocks are APPENDED to FC1's grid: the dispatcher hands them the slots FC1's draining rounds free, so the
// heads' MFMAs fill FC1's ragged end, and two launch boundaries (FC1 -> heads, heads -> gate) disappear.
//
//   blocks [0, rem_blocks)            FC1, remainder rows as 64 x 112 tiles (as in k_fc1_bulk)
//   blocks [rem_blocks, fc1_blocks)   FC1, bulk rows as 128 x 112 tiles
//   blocks [fc1_blocks, +3 tiles64)   one head of a 64-CTU tile (head 16, 32, 64 of tile 0, then tile 1, ...), remainder
//                                     rows' tiles first: in the order FC1 finishes them
//
// Hand-off inside the launch (no launch boundary = no implicit cache flush between the stages):
//   * an FC1 block stores its h1 tile with agent-scope stores (sc1: written through its XCD's L2), waits for them
//     (s_waitcnt vmcnt(0)), and only then adds 1 to the completion counter of its M tile (4 column blocks per tile);
//   * a heads block spins (thread 0, s_sleep) until its tile's counter reads 4, then reads h1 with agent-scope loads (sc1).
//     It can never wait for a block that has not been dispatched: heads blocks have the highest block ids of the grid and
//     workgroups are dispatched in id order, so every FC1 block is resident (or done) before the first heads block starts
//     -- and a resident FC1 block never waits for anything.  The spin is bounded all the same (trap after ~2 s: a loud
//     launch failure instead of a hung GPU).
//   * the gates (tf.cond, net_CNN.py:175,187) are applied per sub-batch by the heads block that completes it (arrival
//     counters, ethcnn_heads_pass.h): probabilities and predicates are agent-scope stores, completed before the block arrives.
// SHARED GPUs: the "only waits for dispatched blocks" argument holds for ONE such launch on the GPU.  Two of them from different
// processes can fill each other's XCDs with waiting blocks while their FC1 blocks queue behind (ethcnn_small.hip met exactly
// that and grew a claim-or-execute path).  This launch plan has none: it is opt-in, measured slower than three launches, and
// must not be selected when several processes share the GPU.
// Counters and gate predicates live in one "sync area" that the pass's CTU-load stage zeroes (ethcnn_tile.hip).
// Same accumulation chains as the separate kernels: results are bit-identical (the parity suite runs over both paths).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ethcnn_fc1_tile.h"
#include "ethcnn_heads_pass.h"
#include "ethcnn_kernels.h"

namespace ethcnn {

struct FusedParams {
    const float* feat;
    const float* wimg;
    const float* bias;
    float* h1;
    int m_main, m_total;          // bulk rows [0, m_main), remainder rows [m_main, m_total)
    unsigned rem_blocks, fc1_blocks, heads_blocks;
    int rem_tiles;                // 64-row remainder tiles (their completion counters come first)
    int* sync;                    // [2 * nchunks gate predicates][nchunks arrival counters][tile completion counters]
    int nchunks;
    HeadsParams hp;
    float qn;
    GateIndex gi;
    float thr1, thr2;
    float *h2, *logits, *raw, *probs;
};

__device__ __forceinline__ unsigned long long realtime_100mhz() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__global__ __launch_bounds__(256) void k_fc1_heads(FusedParams P) {
    __shared__ __attribute__((aligned(16))) float smem[Fc1Shape<2, 7, 4, 1, 3>::LDS_FLOATS];  // 46 KB: FC1's ring / the heads' stages
    __shared__ GateArrive s_ga;
    const unsigned bid = blockIdx.x;
    int* const done = P.sync + 3 * P.nchunks;  // [2 n predicates][n sub-batch arrival counters][tile completion counters]
    if (bid < P.fc1_blocks) {
        int mt, nb, cidx;
        if (bid < P.rem_blocks) {
            fc1_block_to_tile<4, true>(bid, mt, nb);
            const int M = P.m_total - P.m_main;
            if (mt * 64 >= M) return;  // padding block of the XCD grouping: leaves before any barrier
            fc1_tile_at<1, 7, 4, 1, 3, true>(smem, P.feat + (size_t)(P.m_main / 16) * kNFeat * 16, P.wimg, P.bias,
                                             P.h1 + (size_t)P.m_main * kNVec, M, mt, nb);
            cidx = mt;
        } else {
            fc1_block_to_tile<4, true>(bid - P.rem_blocks, mt, nb);
            if (mt * 128 >= P.m_main) return;
            fc1_tile_at<2, 7, 4, 1, 3, true>(smem, P.feat, P.wimg, P.bias, P.h1, P.m_main, mt, nb);
            cidx = P.rem_tiles + mt;
        }
        // publish: every h1 store of this block (agent scope, write-through) has completed before the counter moves
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(done + cidx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }

    // ---- one head of one 64-CTU tile
    const unsigned hb = bid - P.fc1_blocks;
    const int t = (int)(hb / 3u), head_ = (int)(hb % 3u);
    int tile64, cidx;
    if (t < P.rem_tiles) { tile64 = P.m_main / 64 + t; cidx = t; }
    else { tile64 = t - P.rem_tiles; cidx = P.rem_tiles + (tile64 >> 1); }
    if (threadIdx.x == 0) {
        const unsigned long long t0 = realtime_100mhz();
        while (__hip_atomic_load(done + cidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 4) {
            __builtin_amdgcn_s_sleep(16);
            if (realtime_100mhz() - t0 > 200000000ull) __builtin_trap();  // 2 s: cannot happen (see header); never hang the GPU
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const unsigned wvu = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15;
    const int N = P.m_total;
    const int ctu_raw = (tile64 * 4 + (int)wvu) * 16 + col;
    const bool valid = ctu_raw < N;
    const int ctu = min(ctu_raw, N - 1);
    float* h2row = P.h2 ? P.h2 + (size_t)ctu * kNFc2 : nullptr;
    int* fl = P.sync;
    if (head_ != 0) fl += 2 * gate_chunk(P.gi, ctu);
    if (head_ == 0)
        head_pass<2, true, true>(smem, P.h1, P.hp, P.qn, lane, wvu, valid, ctu, h2row, P.logits, P.raw, P.probs, fl, fl + 1, P.thr1, P.thr2);
    else if (head_ == 1)
        head_pass<1, true, true>(smem, P.h1, P.hp, P.qn, lane, wvu, valid, ctu, h2row, P.logits, P.raw, P.probs, fl, fl + 1, P.thr1, P.thr2);
    else
        head_pass<0, true, true>(smem, P.h1, P.hp, P.qn, lane, wvu, valid, ctu, h2row, P.logits, P.raw, P.probs, fl, fl + 1, P.thr1, P.thr2);

    // ---- gates: applied per sub-batch by the block that completes it (ethcnn_heads_pass.h)
    heads_gates_arrive(P.sync, P.sync + 2 * P.nchunks, P.gi, N, tile64 * 64, P.thr2, P.probs, &s_ga);
}

// the bulk part must run for several rounds of resident blocks (same rule as k_fc1_bulk, ethcnn_dense.hip): below that the
// separate launches win
static int fused_big_tiles(int n) {
    const int big_tiles = ((n / 128) * 4 / 256) * 256 / 4;
    return big_tiles >= 3 * 192 ? big_tiles : 0;
}

bool fc1_heads_fusable(int n) { return fused_big_tiles(n) > 0; }

void launch_fc1_heads(const Workspace& ws, const DeviceWeights& w, int n, float qn, int nctu, long ctu0, float thr1, float thr2,
                      float* d_probs, int nchunks, hipStream_t s) {
    FusedParams P;
    const int big_tiles = fused_big_tiles(n);
    P.feat = ws.feat;
    P.wimg = w.fc1_img112;
    P.bias = w.fc1_b;
    P.h1 = ws.h1;
    P.m_main = big_tiles * 128;
    P.m_total = n;
    P.rem_tiles = (n - P.m_main + 63) / 64;
    P.rem_blocks = (unsigned)((P.rem_tiles + 7) / 8) * 8 * 4;  // GROUP mapping: tiles in eights, 4 column blocks
    P.fc1_blocks = P.rem_blocks + (unsigned)((big_tiles + 7) / 8) * 8 * 4;
    P.heads_blocks = (unsigned)((n + 63) / 64) * 3u;
#ifdef ETHCNN_EXPERIMENTS  // A/B builds only (scripts/build_variant.sh NAME -DETHCNN_EXPERIMENTS): these produce WRONG results
    static const int exp_mode = [] { const char* e = dev_env("ETHCNN_FUSED_EXP"); return e ? atoi(e) : 0; }();
    if (exp_mode == 1) P.heads_blocks = 0;  // FC1 part alone (agent-scope stores + completion counters), no heads blocks
#endif
    P.sync = ws.flags;
    P.nchunks = nchunks;
    for (int h = 0; h < 3; ++h) {
        P.hp.w2[h] = w.fc2_w[h];
        P.hp.w2lane[h] = w.fc2_lane[h];
        P.hp.b2[h] = w.fc2_b[h];
        P.hp.w3[h] = w.fc3_w[h];
        P.hp.b3[h] = w.fc3_b[h];
    }
    P.qn = qn;
    P.gi = make_gate_index(nctu, ctu0);
    P.thr1 = thr1;
    P.thr2 = thr2;
    P.h2 = ws.h2;
    P.logits = ws.logits;
    P.raw = ws.raw;
    P.probs = d_probs;
    hipLaunchKernelGGL(k_fc1_heads, dim3(P.fc1_blocks + P.heads_blocks), dim3(256), 0, s, P);
}

}  // namespace ethcnn
