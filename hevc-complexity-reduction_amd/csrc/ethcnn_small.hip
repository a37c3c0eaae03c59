// ethcnn_small.hip -- a SMALL pass (one picture: the in-process encoder hook, every Low-Delay-P frame, the reference's own
// 768x512 case) as ONE launch: CTU load + trunk -> FC1 -> heads -> gates (video_to_cu_depth.py:61-73 + net_CNN.py:103-187; LDP
// front-end: resi_cnn, net_CNN_LSTM_one_step.py:151-199, stops after FC1).
//
// Why: a picture of a few hundred CTUs was five dependent launches (tile, trunk, FC1, heads, gate) of 10-20 us each with the
// GPU < 10 % occupied -- every one pays its own dispatch, kernarg fetch, cold instruction cache, weight staging and drain, and
// the stages cannot overlap (profiles/r02_latency.txt: 69 us for a 1080p picture whose longest dependent chain, FC1's 672 MFMA
// steps, is 12.9 us of dependent-issue latency, profiles/r03_chain_probe.txt).  Here the whole pass is a DATAFLOW inside one grid:
//
//   blocks [0, nT)          trunk: one wave = one unit position of 16 CTUs, pixels gathered STRAIGHT from the luma frame
//                           (Trunk<.., DIRECT>: the CTU-load stage folded into its consumer; no record buffers, no tile launch)
//   blocks [nT, +nF)        FC1 64 x (16 NS) tiles (register-fed up to 1536 CTUs and for the LDP front-end: ethcnn_fc1_regs.h);
//                           a block starts when the 4 x 21 trunk tasks of its 64 CTUs have landed
//   blocks [.., +3 groups)  one head of a group of 16 CTUs (its waves split the head's FC2 tiles: head_pass_regs); starts when
//                           the NSPLIT FC1 column blocks of its 64-CTU tile have landed; applies the gates per sub-batch
//
// PULL form (round 4; the picture lies in page-locked HOST memory: the encoder hook's own buffer, the LDP daemon's frame): blocks
// [0, groups) come FIRST and read the picture over PCIe themselves -- one group of 16 CTUs each, 64 KiB requested at once, turned
// into the trunk's pixel records (tile_group, ethcnn_tile_group.h) -- and the trunk blocks (then ordered group by group) start
// as their group's records land.  No copy-engine launch in front of the kernel, no stream dependency, and trunk / FC1 / heads of
// the first CTU rows run while the last rows are still on the bus (3840x2160: 151 us of PCIe, scripts/ubench/row_arrival_probe.hip).
//
// Every consumer block has a higher block id than its producers and workgroups are dispatched in id order, so ALONE on the GPU a
// waiting block can only wait for blocks that are resident or finished, and producers never wait (all blocks of a 1080p pass
// are co-resident anyway).  When several processes share the GPU that is not enough -- two such launches can fill each other's
// XCDs with waiting blocks -- hence CLAIM OR EXECUTE (do_fc1_item / the heads role below): a consumer that has waited too long
// executes the unclaimed items it depends on itself.  All hand-offs are agent-scope (sc1) stores / loads, completed
// (s_waitcnt vmcnt(0)) before the producer's counter moves -- the scheme of round 3's fused big-pass launch (removed in round 6; DESIGN.md
// "hand-offs inside a launch") -- with the signalling shaped for LATENCY (see SmallSync below): finisher-notifies-private-flag
// instead of polled counters.  The stages' launch overheads, weight staging and drains overlap instead of adding up.
// The sync area is ZERO between launches by construction: every word is reset by its unique last user.
// Same device functions, same accumulation chains as the multi-launch path: bit-identical results (tests run both).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "ethcnn_fc1_regs.h"
#include "ethcnn_fc1_tile.h"
#include "ethcnn_heads_pass.h"
#include "ethcnn_kernels.h"
#include "ethcnn_tile_group.h"
#include "ethcnn_trunk_task.h"

namespace ethcnn {

struct SmallParams {
    DirectSrc src;                // PULL form: `luma` is page-locked host memory, read once by the pull blocks ...
    uint4 *xs, *xm, *xl;          // ... into the pixel records of the pass (the workspace's: ethcnn_tile.hip layouts)
    TileWait tw;                  // streamed input (rows != null): a pull item first waits for the caller to report its CTU rows
    const float* trunk_w;
    const float* trunk_b;
    float* feat;
    int n, ngroups, ntiles, nchunks;
    int npull;                    // PULL form: pull blocks of the launch; block b pulls groups b, b + npull, ... (see launch_small_pass)
    int bS, bM, bL;               // trunk blocks per branch (S / M: 4 waves = 4 tasks; L: one task per block)
    const float* wimg;            // FC1 weights as the LDS image of the launch's tile shape ...
    const float* wlane;           // ... and in MFMA-operand order per 16-column tile (fc1_tile_regs)
    const float* fc1_b;
    float* h1;                    // FC1 output: the pass's h1 (AI) or the caller's 448-vectors (LDP front-end)
    unsigned fc1_blocks, heads_blocks;
    int* sync;                    // SmallSync layout below; zero on entry, zero again on exit
    HeadsParams hp;
    float qn;
    GateIndex gi;
    float thr1, thr2;
    float *h2, *logits, *raw, *probs;
    int exp;  // 0 in production; timing experiments of -DETHCNN_EXPERIMENTS builds: 1 = consumers do not wait (WRONG results)
    int epoch;      // this launch's claim tag (never 0; claim words hold the tag of the last launch that claimed them)
    unsigned* done;      // completion word in page-locked HOST memory (null: none) ...
    unsigned done_seq;   // ... the launch's last finishing block stores this value there, after every output of the launch
    float* host_probs;   // page-locked HOST memory, 16-byte aligned (null: none): that block first copies the launch's probabilities
                         // there, sixteen bytes per lane -- the host then needs no copy launch behind the kernel (host -> host calls)
    int steal_test; // 0 in production; k > 0: role blocks with id % k == 1 leave WITHOUT claiming and consumers have no patience,
                    // so the items must be executed by the consumers that depend on them (tests; results stay correct)
};

// ---- signalling.  Measured on MI355X (profiles/r03_small_pass_timeline.txt): agent-scope atomics and polls on ONE address are served
// one at a time, ~150 ns each -- 224 blocks drawing tickets from one word cost 35 us, 28 blocks polling one counter delay the
// producer's own add by microseconds.  So:
//   * counters are only ever ADDED to (never polled): the add returns the old value, and the adder that completes a counter
//     is its FINISHER -- it resets the counter to zero (nobody else will touch it again in this launch) and notifies the
//     consumers;
//   * every consumer block polls a PRIVATE flag word in its own 128-byte line, written once by a finisher, and resets it
//     itself: one poller, one writer per address;
//   * counters and flags are padded to one per 128-byte line (one memory channel queue each);
//   * the trunk signals per BLOCK (the four tasks of an S / M block belong to one group): 6 adds per group, not 21.
// Every word is reset by its unique finisher / consumer, so the area is zero again when the launch ends: no clearing pass,
// no launch-wide ticket.
constexpr int kPad = 32;  // ints per 128-byte line
struct SmallSync {
    int* pred;         // [2 nchunks]          gate predicates (dense)
    int* arrive;       // [nchunks]            heads blocks arrived per gate sub-batch (dense)
    int* feat_done;    // [ngroups] x kPad     trunk tasks finished per group of 16 CTUs (21 = complete)
    int* tile_groups;  // [ntiles] x kPad      complete groups per 64-CTU tile
    int* fc1_flag;     // [fc1_blocks] x kPad  "your tile's features have landed", one per FC1 block
    int* fc1_done;     // [ntiles] x kPad      FC1 column blocks finished per tile
    int* heads_flag;   // [3 ngroups] x kPad   "your tile's h1 has landed", one per heads block (16 CTUs x head)
    int* done_cnt;     // [1] x kPad           gate sub-batches completed (nchunks = the launch's outputs are final)
    int* trunk_flag;   // [6 ngroups] x kPad   PULL form: "your group's pixel records have landed", one per trunk work item
    int words;         // of the part above: every word of it is zero between launches (reset by its last user)
    // claim words (epoch tags, NEVER reset: a stale tag simply differs from the current one) live at FIXED offsets behind the
    // largest possible self-cleaning part, because the layout above moves with the geometry and a stale tag must never be
    // read as a counter or a flag of a later launch
    int* claim_t;      // [trunk blocks] x kPad  claim word per trunk work item
    int* claim_f;      // [fc1_blocks] x kPad    claim word per FC1 work item
    int* claim_l;      // [ngroups] x kPad       PULL form: claim word per pull work item (= group)
};
constexpr int kMaxGroups = (kSmallPassMaxCtus + 15) / 16, kMaxTiles = (kSmallPassMaxCtus + 63) / 64, kMaxFc1Blocks = kMaxTiles * 28;
constexpr int kClaimBase = 2 * kSmallPassMaxCtus + (kSmallPassMaxCtus + kPad - 1) / kPad * kPad + kPad +
                           (kMaxGroups + kMaxTiles + kMaxFc1Blocks + kMaxTiles + 3 * kMaxGroups + 1 + 6 * kMaxGroups) * kPad;  // >= words for any n, nchunks <= n
constexpr int kSyncTotalWords = kClaimBase + (6 * kMaxGroups + kMaxFc1Blocks + kMaxGroups) * kPad;
__host__ __device__ inline SmallSync small_sync(int* base, int nchunks, int ngroups, int ntiles, int fc1_blocks) {  // (heads blocks = 3 ngroups)
    SmallSync s;
    s.pred = base;
    s.arrive = s.pred + 2 * nchunks;
    s.feat_done = s.arrive + (nchunks + kPad - 1) / kPad * kPad + kPad;  // (first padded line starts on a fresh 128 B)
    s.tile_groups = s.feat_done + ngroups * kPad;
    s.fc1_flag = s.tile_groups + ntiles * kPad;
    s.fc1_done = s.fc1_flag + fc1_blocks * kPad;
    s.heads_flag = s.fc1_done + ntiles * kPad;
    s.done_cnt = s.heads_flag + 3 * ngroups * kPad;
    s.trunk_flag = s.done_cnt + kPad;
    s.words = (int)(s.trunk_flag + 6 * ngroups * kPad - base);
    s.claim_t = base + kClaimBase;
    s.claim_f = s.claim_t + 6 * kMaxGroups * kPad;  // trunk blocks: 4 S + 1 M + 1 L per group
    s.claim_l = s.claim_f + kMaxFc1Blocks * kPad;
    return s;
}

__device__ __forceinline__ int add_ret(int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void put(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int get(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one thread polls its block's private flag, then resets it.  budget (100 MHz ticks) > 0: give up after that long and
// return false (the caller then executes the unclaimed items it depends on); budget 0: wait until it comes -- every item
// it stands for is claimed by a resident block by then.  The trap after 30 s of wall clock is a last resort against a hung
// GPU (wall clock: a process descheduled under GPU sharing must not trip it).
__device__ __forceinline__ bool wait_flag(int* p, int exp, unsigned long long budget) {
    if (exp == 1) return true;
    unsigned long long t0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    while (get(p) == 0) {
        __builtin_amdgcn_s_sleep(1);
        unsigned long long t;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        if (budget != 0 && t - t0 > budget) return false;
        if (t - t0 > 3000000000ull) __builtin_trap();
    }
    put(p, 0);
    return true;
}
#ifdef SMALL_NO_STEAL  // A/B builds only: the first form of the launch (wait for ever) -- to show what the shared-GPU test catches
constexpr unsigned long long kPatience = 0;
#else
constexpr unsigned long long kPatience = 10000;  // 100 us
#endif

#ifdef SMALL_STAMPS
// development probe (scripts/ubench/small_probe.hip): device-wide 100 MHz stamps per block: entry, woken, computed, exit
__device__ unsigned long long g_small_stamps[1 << 13][4];
__device__ __forceinline__ void small_stamp(int slot) {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    if (threadIdx.x == 0) g_small_stamps[blockIdx.x & 0x1fff][slot] = t;
}
#define SMALL_STAMP(i) small_stamp(i)
#else
#define SMALL_STAMP(i)
#endif

template <int A, int B>
struct MaxOf { static constexpr int value = A > B ? A : B; };

struct SmallShared {  // block-uniform scratch of the roles
    int owned;   // this block holds the claim of the item it is working on
    int woken;   // wait_flag's answer
    GateArrive ga;
};

// every thread of the block; patience: 0 = wait for the flag, whatever it takes
__device__ __forceinline__ bool block_wait(int* flag, const SmallParams& P, SmallShared* sh, unsigned long long patience) {
    if (threadIdx.x == 0) sh->woken = wait_flag(flag, P.exp, patience) ? 1 : 0;
    __syncthreads();
    const bool w = sh->woken != 0;
    __syncthreads();
    if (w) ETHCNN_HANDOFF_ACQUIRE();  // (every wave: what the flag announces is read behind this point)
    return w;
}

// ---- a pull work item (PULL form): the pixel records of group `grp`, read from page-locked host memory.  Claims the item; the
// owner wakes the group's six trunk items.  Called by the block the item was meant for -- or by a trunk item tired of waiting.
__device__ __forceinline__ void do_pull_item(const SmallParams& P, const SmallSync& Y, int grp, float* smem, SmallShared* sh) {
    if (threadIdx.x == 0) sh->owned = (__hip_atomic_exchange(Y.claim_l + grp * kPad, P.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != P.epoch);
    __syncthreads();
    const bool mine = sh->owned != 0;
    __syncthreads();
    if (!mine) return;
    if (P.tw.rows != nullptr) tile_wait_rows(P.tw, P.src.ctu0, grp, P.src.n_total, P.src.nctu, P.src.cw);
    tile_group<true, true, true>(reinterpret_cast<uint32_t*>(smem), P.src.luma, P.src.width, P.src.height, P.src.pitch, P.src.frame_stride, P.src.cw,
                                 P.src.nctu, P.src.ctu0, P.src.n_total, grp, P.xs, P.xm, P.xl);
    ETHCNN_HANDOFF_RELEASE();  // the records (agent-scope stores) have completed before the flags move
    __syncthreads();
    if (threadIdx.x < 6) {  // the group's trunk items: 4 S blocks, the M block, the L block
        const int k = threadIdx.x, item = k < 4 ? 4 * grp + k : (k == 4 ? P.bS + grp : P.bS + P.bM + grp);
        put(Y.trunk_flag + item * kPad, 1);
    }
    __syncthreads();
}

// ---- a trunk work item (+ CTU gather): items [0, 4 G) = S blocks (four tasks each), [4 G, 5 G) = M blocks, [5 G, 6 G) = L
// blocks (one task, gathered by the four waves together), G = groups of 16 CTUs; every item belongs to ONE group.  Claims
// the item (inside Trunk::run, under its first loads); the owner signals the group, the group's finisher the tile, the
// tile's finisher wakes the FC1 blocks.  Called by the block the item was meant for -- or by a consumer tired of waiting.
template <int NSPLIT, bool RESI, bool PULL>
__device__ __forceinline__ void do_trunk_item(const SmallParams& P, const SmallSync& Y, int item, float* smem, SmallShared* sh) {
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int* const claim = Y.claim_t + item * kPad;
    int grp, ntask;
    if (PULL) {
        // the item is claimed HERE (its flag has one poller: the owner), then waits for its group's records -- pulling them
        // itself if the pull block has not come by (claim or execute) -- and runs the trunk on the records as the
        // multi-launch path does
        if (threadIdx.x == 0) sh->owned = (__hip_atomic_exchange(claim, P.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != P.epoch);
        __syncthreads();
        const bool mine = sh->owned != 0;
        __syncthreads();
        if (!mine) return;
        grp = item < P.bS ? item >> 2 : (item < P.bS + P.bM ? item - P.bS : item - P.bS - P.bM);
        // the wait sits INSIDE the trunk task, behind its weight staging (20 KB of fragments; LDP: + the division table): when the
        // records land the task only has to load them
        auto ready = [&]() -> bool {
            bool lent = false;
            // (patience: the last rows of a 2160p picture are 150 us away by design; a trunk item that gave up after 100 us pulled its
            // group itself, out of turn, beside the 16 pull blocks -- correct, but not the order the transfer was laid out for)
            if (!block_wait(Y.trunk_flag + item * kPad, P, sh, P.steal_test ? 1 : 5 * kPatience)) {
                do_pull_item(P, Y, grp, smem, sh);  // (its 16-row slabs go through the block's LDS)
                (void)block_wait(Y.trunk_flag + item * kPad, P, sh, 0);  // the group is claimed by a resident block now
                lent = true;
            }
            SMALL_STAMP(1);
            return lent;
        };
        if (item < P.bS) {
            ntask = 4;
            Trunk<0, RESI, false, true>::run(P.xs, P.ngroups * 16, item * 4 + wv, P.bS * 4, P.trunk_w, P.trunk_b, P.feat, P.n, smem, nullptr, nullptr, 0, nullptr, 1.0f, ready);
        } else if (item < P.bS + P.bM) {
            ntask = 4;
            Trunk<1, RESI, false, true>::run(P.xm, P.ngroups * 4, (item - P.bS) * 4 + wv, P.bM * 4, P.trunk_w, P.trunk_b, P.feat, P.n, smem, nullptr, nullptr, 0, nullptr, 1.0f, ready);
        } else {
            ntask = 1;  // (one task: wave 0's; the others help with the weight staging and leave)
            Trunk<2, RESI, false, true>::run(P.xl, wv == 0 ? P.ngroups : 0, grp, P.bL, P.trunk_w, P.trunk_b, P.feat, P.n, smem, nullptr, nullptr, 0, nullptr, 1.0f, ready);
        }
        if (threadIdx.x == 0) sh->owned = 1;  // (claimed above; a pull item tried inside the wait has used the word for its own claim)
    } else if (item < P.bS) {
        grp = item >> 2; ntask = 4;
        Trunk<0, RESI, true, true>::run(nullptr, P.ngroups * 16, item * 4 + wv, P.bS * 4, P.trunk_w, P.trunk_b, P.feat, P.n, smem, &P.src, claim, P.epoch, &sh->owned);
    } else if (item < P.bS + P.bM) {
        grp = item - P.bS; ntask = 4;
        Trunk<1, RESI, true, true>::run(nullptr, P.ngroups * 4, (item - P.bS) * 4 + wv, P.bM * 4, P.trunk_w, P.trunk_b, P.feat, P.n, smem, &P.src, claim, P.epoch, &sh->owned);
    } else {
        grp = item - P.bS - P.bM; ntask = 1;
        Trunk<2, RESI, true, true>::run(nullptr, P.ngroups, grp, P.bL, P.trunk_w, P.trunk_b, P.feat, P.n, smem, &P.src, claim, P.epoch, &sh->owned);
    }
    // the block's features (agent-scope stores) have completed before its group's counter moves
    ETHCNN_HANDOFF_RELEASE();
    __syncthreads();
    if (threadIdx.x == 0 && sh->owned && add_ret(Y.feat_done + grp * kPad, ntask) + ntask == 21) {  // 16 S + 4 M + 1 L: group complete
        put(Y.feat_done + grp * kPad, 0);
        const int tile = grp >> 2, in_tile = min(4, P.ngroups - 4 * tile);
        if (add_ret(Y.tile_groups + tile * kPad, 1) + 1 == in_tile) {             // tile complete: wake its FC1 items
            put(Y.tile_groups + tile * kPad, 0);
            for (int nb = 0; nb < NSPLIT; ++nb) put(Y.fc1_flag + (tile * NSPLIT + nb) * kPad, 1);
        }
    }
    __syncthreads();  // (smem and sh are free for the caller's next item)
}

// ---- an FC1 work item: column block nb of 64-CTU tile mt.  Claims it, waits for the tile's features (executing the tile's
// unclaimed trunk items itself when that takes too long), computes, signals the tile; the tile's finisher wakes its heads.
#ifndef SMALL_D2
#define SMALL_D2 10  // ring depth (sub-chunks) of the 32-column shape
#endif
#ifndef SMALL_D4
#define SMALL_D4 6   // ... of the 64-column shape
#endif
#ifndef SMALL_FC1_REGS
#define SMALL_FC1_REGS 1  // 0: the LDS-staged tile everywhere (A/B builds)
#endif
// Which tile a launch uses (measured, profiles/r03_fc1_regs.txt): register-fed for the 16-column shape (<= 1536 CTUs: 1080p
// 51.7 -> 46.5 us, 768x512 49.7 -> 43.6) and for the LDP front-end at every size (2160p 85.7 -> 80.0); a 2160p All-Intra
// picture (1600 blocks on 512 slots, the heads blocks among them) keeps the LDS-staged 32-column tile (101 vs 106 us).
template <int NS, bool RESI>
__device__ __host__ constexpr bool fc1_regs() { return SMALL_FC1_REGS != 0 && (NS == 1 || RESI); }

template <int NS, int NSUB, bool RESI, bool PULL>
__device__ __forceinline__ void do_fc1_item(const SmallParams& P, const SmallSync& Y, int fb, float* smem, SmallShared* sh) {
    constexpr int NSPLIT = kNVec / (16 * NS);
    if (threadIdx.x == 0) sh->owned = (__hip_atomic_exchange(Y.claim_f + fb * kPad, P.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != P.epoch);
    __syncthreads();
    const bool mine = sh->owned != 0;
    __syncthreads();
    if (!mine) return;
    const int nb = fb % NSPLIT, mt = fb / NSPLIT;
    if (!block_wait(Y.fc1_flag + fb * kPad, P, sh, P.steal_test ? 1 : (PULL ? 5 : 1) * kPatience)) {  // (PULL: the last rows of a 2160p picture are 150 us away)
        // claim or execute: the trunk items of this tile's groups that nobody has claimed yet (an item that already carries
        // this launch's tag needs no second look: the claim inside Trunk::run is only answered after its first loads)
        const int g0 = 4 * mt, g1 = min(4 * mt + 4, P.ngroups);
#pragma unroll 1
        for (int g = g0; g < g1; ++g)
#pragma unroll 1
            for (int k = 0; k < 6; ++k) {  // 4 S blocks, the M block, the L block of the group
                const int item = k < 4 ? 4 * g + k : (k == 4 ? P.bS + g : P.bS + P.bM + g);
                // ONE thread looks and the block follows its answer: sixty-four loads of a word that is changing under them
                // would split the block on the way into a function full of barriers
                if (threadIdx.x == 0) sh->woken = (get(Y.claim_t + item * kPad) != P.epoch);
                __syncthreads();
                const bool unclaimed = __builtin_amdgcn_readfirstlane(sh->woken) != 0;
                __syncthreads();
                if (unclaimed) do_trunk_item<NSPLIT, RESI, PULL>(P, Y, item, smem, sh);
            }
        (void)block_wait(Y.fc1_flag + fb * kPad, P, sh, 0);  // every item of the tile is claimed by a resident block now
    }
    SMALL_STAMP(1);
    if (fc1_regs<NS, RESI>()) fc1_tile_regs<NS, (NS == 1 ? 16 : (NS == 2 ? SMALL_D2 : SMALL_D4)), true>(P.feat, P.wlane, P.fc1_b, P.h1, P.n, mt, nb);
    else fc1_tile_at<1, NS, 4, NSUB, 3, true, true>(smem, P.feat, P.wimg, P.fc1_b, P.h1, P.n, mt, nb);
    SMALL_STAMP(2);
    if (!RESI) {  // (LDP front-end: the vectors are the launch's output, nothing waits for them inside it)
        ETHCNN_HANDOFF_RELEASE();
        __syncthreads();
        if (threadIdx.x == 0 && add_ret(Y.fc1_done + mt * kPad, 1) + 1 == NSPLIT) {  // all column blocks of the tile: wake its heads
            put(Y.fc1_done + mt * kPad, 0);
            const int g1 = min(4 * mt + 4, P.ngroups);
            for (int hb = 12 * mt; hb < 3 * g1; ++hb) put(Y.heads_flag + hb * kPad, 1);  // (group, head) blocks of the tile
        }
    }
    __syncthreads();
}

template <int NS, int NSUB, bool RESI, bool PULL>
__global__ __launch_bounds__(256, 2) void k_small_pass(SmallParams P) {  // <= 256 registers: two blocks per CU
    constexpr int NSPLIT = kNVec / (16 * NS);
    constexpr int LDS_FLOATS = MaxOf<MaxOf<(fc1_regs<NS, RESI>() ? 0 : Fc1Shape<1, NS, 4, NSUB, 3>::LDS_FLOATS), (RESI ? kTrunkResiLds : kTrunkWFrags * 64 + 8 * 64 * 4)>::value, (RESI ? (PULL ? 16 * kSlabCtuPitch : 0) : (NS == 1 ? 12 * 256 : kHeadsLatStages * kHeadsStage + 12 * 256))>::value;
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
    __shared__ SmallShared sh;
    const int nL = PULL ? P.npull : 0;  // pull blocks come first: they wait for nobody
    const int bid = (int)blockIdx.x - nL;
    const int nT = P.bS + P.bM + P.bL;
    const SmallSync Y = small_sync(P.sync, P.nchunks, P.ngroups, P.ntiles, (int)P.fc1_blocks);
    SMALL_STAMP(0);
    const bool is_producer = bid < nT + (int)P.fc1_blocks;
    // tests: "this producer block never got a slot" -- only blocks somebody in this launch waits for (nobody waits for the FC1
    // blocks of the LDP front-end: a late one of those simply runs late)
    if (P.steal_test > 0 && (RESI ? bid < nT : is_producer) && (int)blockIdx.x % P.steal_test == 1) return;

    if (PULL && bid < 0) {
#pragma unroll 1
        for (int grp = (int)blockIdx.x; grp < P.ngroups; grp += nL) do_pull_item(P, Y, grp, smem, &sh);
        SMALL_STAMP(3);
        return;
    }
    if (bid < nT) {
        // PULL form: the six trunk items of a group are neighbours in block order (the groups' records arrive one after the
        // other: a group's M and L blocks must not queue behind the S blocks of groups that are still on the bus)
        const int k6 = bid % 6, g6 = bid / 6;
        const int item = !PULL ? bid : (k6 < 4 ? 4 * g6 + k6 : (k6 == 4 ? P.bS + g6 : P.bS + P.bM + g6));
        do_trunk_item<NSPLIT, RESI, PULL>(P, Y, item, smem, &sh);
        SMALL_STAMP(3);
        return;
    }
    if (is_producer) {
        do_fc1_item<NS, NSUB, RESI, PULL>(P, Y, bid - nT, smem, &sh);
        SMALL_STAMP(3);
        return;
    }
    if (RESI) return;  // (no heads blocks are launched for the LDP front-end)
    // ---- one head of one group of 16 CTUs (the block's waves split the head's FC2 tiles: head_pass_regs)
    const int hb = bid - nT - (int)P.fc1_blocks;
    const int grp = hb / 3, head_ = hb % 3;
    if (!block_wait(Y.heads_flag + hb * kPad, P, &sh, P.steal_test ? 1 : (PULL ? 5 : 1) * kPatience)) {
        // claim or execute: the FC1 items of this group's tile (each of which does the same for its trunk items)
        const int mt = grp >> 2;
#pragma unroll 1
        for (int nb = 0; nb < NSPLIT; ++nb) do_fc1_item<NS, NSUB, RESI, PULL>(P, Y, mt * NSPLIT + nb, smem, &sh);
        (void)block_wait(Y.heads_flag + hb * kPad, P, &sh, 0);
    }
    SMALL_STAMP(1);
    const int lane = threadIdx.x & 63;
    const unsigned wvu = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int N = P.n;
    const int ctu_raw = grp * 16 + (lane & 15);
    const bool valid = ctu_raw < N;
    const int ctu = min(ctu_raw, N - 1);
    float* h2row = P.h2 ? P.h2 + (size_t)ctu * kNFc2 : nullptr;
    int* fl = Y.pred;
    if (head_ != 0) fl += 2 * gate_chunk(P.gi, ctu);
    if (NS == 1) {  // up to 1536 CTUs: register-fed heads (1080p 45.7 -> 43.6 us); bigger pictures: the LDS-ring form (2160p 103 vs 110 us)
        if (head_ == 0) head_pass_regs<2>(smem, P.h1, P.hp, P.qn, lane, wvu, valid, ctu, h2row, P.logits, P.raw, P.probs, fl, fl + 1, P.thr1, P.thr2);
        else if (head_ == 1) head_pass_regs<1>(smem, P.h1, P.hp, P.qn, lane, wvu, valid, ctu, h2row, P.logits, P.raw, P.probs, fl, fl + 1, P.thr1, P.thr2);
        else head_pass_regs<0>(smem, P.h1, P.hp, P.qn, lane, wvu, valid, ctu, h2row, P.logits, P.raw, P.probs, fl, fl + 1, P.thr1, P.thr2);
    } else {
        if (head_ == 0) head_pass_split<2>(smem, P.h1, P.hp, P.qn, lane, wvu, valid, ctu, h2row, P.logits, P.raw, P.probs, fl, fl + 1, P.thr1, P.thr2);
        else if (head_ == 1) head_pass_split<1>(smem, P.h1, P.hp, P.qn, lane, wvu, valid, ctu, h2row, P.logits, P.raw, P.probs, fl, fl + 1, P.thr1, P.thr2);
        else head_pass_split<0>(smem, P.h1, P.hp, P.qn, lane, wvu, valid, ctu, h2row, P.logits, P.raw, P.probs, fl, fl + 1, P.thr1, P.thr2);
    }
    // gates per sub-batch, applied by the block that completes it, which also hands its words back as zeros
    SMALL_STAMP(2);
    heads_gates_arrive<true, 16>(Y.pred, Y.arrive, P.gi, N, grp * 16, P.thr2, P.probs, &sh.ga);
    // completion word: a host thread spinning on a word in page-locked memory sees the end of the launch ~5 us before
    // hipStreamSynchronize returns (scripts/ubench/launch_rtt.hip).  Every sub-batch is completed by exactly one block; the
    // block that completes the last one has seen, through the arrival counters, every probability of the launch stored
    // (s_waitcnt vmcnt(0) before each arrival) -- its own zero-fills included once it has waited for them here.
    if (P.done && sh.ga.n > 0) {
        ETHCNN_HANDOFF_RELEASE();
        __syncthreads();
        if (threadIdx.x == 0) {
            const bool last = add_ret(Y.done_cnt, sh.ga.n) + sh.ga.n == P.nchunks;
            if (last) put(Y.done_cnt, 0);
            sh.woken = last ? 1 : 0;
        }
        __syncthreads();
        if (sh.woken) ETHCNN_HANDOFF_ACQUIRE();
        if (sh.woken) {  // (block-uniform) this block completes the launch
            if (P.host_probs != nullptr) {
                // every probability of the launch is in memory (agent-scope stores, completed before their blocks' arrivals): hand them
                // to the host with full-width stores -- 96 heads blocks writing 4-byte words over PCIe were slower than a copy launch
                // (profiles/r03_completion_word.txt); one block writing 16 bytes per lane is not
                const int n16 = (N * kNOut) / 4, tail = (N * kNOut) % 4;
                const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(P.probs, 0, -1, 0x00020000);
                u32x4* dst = reinterpret_cast<u32x4*>(P.host_probs);
#pragma unroll 4
                for (int i = (int)threadIdx.x; i < n16; i += 256) dst[i] = __builtin_amdgcn_raw_buffer_load_b128(rP, i * 16, 0, kAuxSc1);
                if ((int)threadIdx.x < tail) {
                    const int i = n16 * 4 + (int)threadIdx.x;
                    P.host_probs[i] = __hip_atomic_load(P.probs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the block's stores to the host have left before the word does
                __syncthreads();
            }
            if (threadIdx.x == 0) __hip_atomic_store(P.done, P.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    SMALL_STAMP(3);
}

// the whole area (fixed size: self-cleaning part for the largest pass + the claim words)
int small_pass_sync_words(int n, int nchunks) {
    (void)n;
    (void)nchunks;
    return kSyncTotalWords;
}

// rows must be 16-byte aligned for the direct gather (the tile stage's own fast-path condition)
bool small_pass_ok(const uint8_t* d_luma, const FrameGeom& g, int n) {
    return n > 0 && n <= kSmallPassMaxCtus && (g.width % 16 == 0) && (g.pitch % 16 == 0) && (g.frame_stride % 16 == 0) &&
           (reinterpret_cast<uintptr_t>(d_luma) % 16 == 0);
}

template <bool RESI, bool PULL>
static void launch_small_t(const SmallParams& P, int shape, unsigned blocks, hipStream_t s) {
    // FC1 shape by row count, as fc1_short_variant (ethcnn_dense.hip): narrow tiles = short per-chunk chains for few rows
    if (shape == 0) hipLaunchKernelGGL((k_small_pass<1, 4, RESI, PULL>), dim3(blocks), dim3(256), 0, s, P);       // 64 x 16, BK 64
    else if (shape == 1) hipLaunchKernelGGL((k_small_pass<2, 2, RESI, PULL>), dim3(blocks), dim3(256), 0, s, P);  // 64 x 32, BK 32
    else hipLaunchKernelGGL((k_small_pass<4, 2, RESI, PULL>), dim3(blocks), dim3(256), 0, s, P);                  // 64 x 64, BK 32
}

void launch_small_pass(const uint8_t* d_luma, const FrameGeom& g, long ctu0, int n, bool resi, const Workspace& ws,
                       const DeviceWeights& w, float* fc1_out, float qn, float thr1, float thr2, float* d_probs, int nchunks,
                       int* d_sync, int epoch, unsigned* done, unsigned done_seq, hipStream_t s, bool pull, const unsigned* wait_rows,
                       unsigned wait_seq, unsigned* gave_up, float* host_probs) {
    // (the LDP front-end keeps the tile-stage launch for a COMPLETE page-locked picture: measured equal to slower in PULL form -- 1080p
    // 116 against 105 us per resident-state call, 2160p 313 against 317 -- its launch ends with FC1 and has no long tail to hide the
    // bus under.  Streamed input is another matter: the rows arrive at the caller's pace, and trunk + FC1 of the first rows run
    // while the caller is still copying the last ones.)
    if (resi && wait_rows == nullptr) pull = false;
    if (!pull) wait_rows = nullptr;
    SmallParams P;
    P.tw = TileWait{wait_rows, wait_seq, gave_up};
    P.xs = ws.xs;
    P.xm = ws.xm;
    P.xl = ws.xl;
    P.done = resi ? nullptr : done;  // (the LDP front-end is not the end of its call)
    P.host_probs = (P.done != nullptr && reinterpret_cast<uintptr_t>(host_probs) % 16 == 0) ? host_probs : nullptr;
    P.done_seq = done_seq;
    P.src.luma = d_luma;
    P.src.width = g.width;
    P.src.height = g.height;
    P.src.pitch = g.pitch;
    P.src.frame_stride = g.frame_stride;
    P.src.cw = g.cw;
    P.src.nctu = g.nctu;
    P.src.ctu0 = ctu0;
    P.src.n_total = n;
    P.trunk_w = w.trunk_w;
    P.trunk_b = w.trunk_b;
    P.feat = ws.feat;
    P.n = n;
    P.ngroups = (n + 15) / 16;
    P.ntiles = (n + 63) / 64;
    P.nchunks = nchunks;
    P.bS = (P.ngroups * 16 + 3) / 4;
    P.bM = (P.ngroups * 4 + 3) / 4;
    P.bL = P.ngroups;  // one L task per block
    static const int force = [] { const char* e = dev_env("ETHCNN_SMALL_SHAPE"); return e ? atoi(e) : -1; }();  // development knob
    // 64 x 16 tiles (register-fed FC1 and heads) up to 1536 CTUs, 64 x 32 above (scripts/latency_mid.py, profiles/r03_latency_mid.txt:
    // 920 CTUs 61.8 vs 81.5 us, 1536 CTUs 82.9 vs 89.9, 1800 CTUs 101.2 vs 99.6)
    // PULL form: 64 x 16 tiles at every size -- the CTUs arrive at the bus's pace (3840x2160: 150 us), the GPU is never full, and what
    // counts is how soon the LAST tile is done after its rows have landed (11 us register-fed against 38 us for the 64 x 32 tile
    // among 448 of its kind: scripts/ubench/small_probe.hip, profiles/r04_pull_timeline.txt; launch 229 -> 209 us)
    int shape = (n <= 1536 || pull) ? 0 : (n <= 2304 ? 1 : 2);
    if (force >= 0 && force <= 2) shape = force;
    const int nsplit = shape == 0 ? 28 : (shape == 1 ? 14 : 7);
    P.wimg = shape == 0 ? w.fc1_img16 : (shape == 1 ? w.fc1_img32 : w.fc1_img64);
    P.wlane = w.fc1_lane16;
    P.fc1_b = w.fc1_b;
    P.h1 = fc1_out;
    P.fc1_blocks = (unsigned)(P.ntiles * nsplit);
    P.heads_blocks = resi ? 0u : (unsigned)P.ngroups * 3u;
    P.sync = d_sync;
    for (int h = 0; h < 3; ++h) {
        P.hp.w2[h] = w.fc2_w[h];
        P.hp.w2lane[h] = w.fc2_lane[h];
        P.hp.b2[h] = w.fc2_b[h];
        P.hp.w3[h] = w.fc3_w[h];
        P.hp.b3[h] = w.fc3_b[h];
    }
    P.qn = qn;
    P.gi = make_gate_index(g.nctu, ctu0);
    P.thr1 = thr1;
    P.thr2 = thr2;
    P.h2 = ws.h2;
    P.logits = ws.logits;
    P.raw = ws.raw;
    P.probs = d_probs;
    P.exp = 0;
    P.epoch = epoch;
    static const int steal_test = [] { const char* e = dev_env("ETHCNN_SMALL_STEAL_TEST"); return e ? atoi(e) : 0; }();  // tests
    P.steal_test = steal_test > 1 ? steal_test : 0;
    unsigned blocks = (unsigned)(P.bS + P.bM + P.bL) + P.fc1_blocks + P.heads_blocks;
#ifdef ETHCNN_EXPERIMENTS  // A/B builds only (scripts/build_variant.sh NAME -DETHCNN_EXPERIMENTS): these produce WRONG results
    static const int exp_mode = [] { const char* e = dev_env("ETHCNN_SMALL_EXP"); return e ? atoi(e) : 0; }();
    P.exp = exp_mode;                                                            // 1: consumers do not wait
    if (exp_mode == 2) blocks = (unsigned)(P.bS + P.bM + P.bL);                  // trunk part alone
    if (exp_mode == 3) blocks = (unsigned)(P.bS + P.bM + P.bL) + P.fc1_blocks;   // trunk + FC1
#endif
    // Pull blocks: FEW above 1080p, each walking its groups in order.  The bus serves the requests of all resident blocks evenly: with
    // one block per group every group of a 2160p picture completes in the last third of the transfer and nothing overlaps (launch
    // 287 us); 16 blocks x 64 KiB in flight cover the bus's bandwidth-delay product several times -- the transfer runs at the copy
    // engine's rate, 8.3 MB in 155 us -- and the groups land in raster order, 1 MiB apart (launch 209 us).  Up to 32 groups (1080p)
    // one block per group: the whole picture is 2 MiB, and a block's second group would wait for its first (84 against 87-92 us).
    // (scripts/ubench/small_probe.hip W H 0 1, profiles/r04_pull_timeline.txt)
    static const int npull_env = [] { const char* e = dev_env("ETHCNN_PULL_BLOCKS"); return e ? atoi(e) : 0; }();  // development knob
    // (streamed input: the same -- a caller that fills the buffer faster than the bus empties it, four converting threads at 2160p,
    // would otherwise have every group's block on the bus at once: 413 -> 328 us per picture was all the helpers bought with one block
    // per group; a pull block sleeps until the caller has reported the rows of the group it has reached)
    P.npull = !pull ? 0 : std::min(P.ngroups, npull_env > 0 ? npull_env : (P.ngroups <= 32 ? 32 : 16));
    blocks += (unsigned)P.npull;
    if (resi) pull ? launch_small_t<true, true>(P, shape, blocks, s) : launch_small_t<true, false>(P, shape, blocks, s);
    else pull ? launch_small_t<false, true>(P, shape, blocks, s) : launch_small_t<false, false>(P, shape, blocks, s);
}

}  // namespace ethcnn
