// ethcnn_lstm.hip -- "next" row 1 of SURVEY.md 8f: ETH-LSTM one step + the LDP heads, i.e. what
// sess.run fetches in predict_cu_depth() (HM-16.5_Test_LDP/bin/resi_to_cu_depth_LDP.py:114-129)
// after resi_cnn: lstm() x3 and the gates of net() (net_CNN_LSTM_one_step.py:201-323).
//   tf.contrib.rnn.LSTMCell(n, forget_bias=1.0, cell_clip=5.0), n = 64 / 128 / 256:
//     z = [x, h_prev] . kernel + bias ;  i, j, f, o = split(z, 4)
//     c = sigmoid(f + 1) * c_prev + sigmoid(i) * tanh(j) ;  c = clip(c, -5, 5) ;  h = sigmoid(o) * tanh(c)
//   h2 = lrelu([h, efs] W2 + b2) ;  y = sigmoid([h2, efs] W3 + b3) ;  efs = [qp/51*0.18, onehot4(i_frame % 4)]
//
// gfx950, v_mfma_f32_16x16x4_f32, one wave = 16 CTUs, "transposed" (rows = gate / output units,
// columns = CTUs) exactly like ethcnn_heads.hip: the four gate pre-activations of hidden unit u
// land in the SAME lane and register slot (tiles u/16 of the i, j, f, o row ranges), so the cell
// update is lane-local, and h_new in C layout is directly the B operand of fc2^T, whose output is
// the B operand of fc3^T.  Canonical order: chains over the leading multiple-of-16 inputs in
// k = 16c + 4g + e order, then the 5 efs columns, then the bias (oracle: oracle_lstm_step).
// LDP calls this once per frame on a few hundred CTUs (lock-step with the encoder): the design
// goal is latency.  Weight operands come straight from L2 (each is used once per block), the
// dependent MFMA chain per wave is kept short: one wave per (hidden tile, gate) in the cell, one per fc2 tile in the heads.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ethcnn_kernels.h"

namespace ethcnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#ifdef LSTM_STAMPS
// development probe (scripts/ubench/lstm_probe.hip): device-wide 100 MHz stamps per block: entry, operands staged, chain done, exit
__device__ unsigned long long g_lstm_stamps[2][1 << 11][4];
__device__ __forceinline__ void lstm_stamp(int kernel, int slot) {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    if (threadIdx.x == 0) g_lstm_stamps[kernel][(blockIdx.y * gridDim.x + blockIdx.x) & 0x7ff][slot] = t;
}
#define LSTM_STAMP(k, i) lstm_stamp(k, i)
#else
#define LSTM_STAMP(k, i)
#endif

__device__ __forceinline__ float lrelu_l(float h) { return fmaxf(0.2f * h, h); }
__device__ __forceinline__ float expf_l(float x) {  // the canonical exp of DESIGN.md
    x = fminf(x, 80.0f);
    x = fmaxf(x, -86.0f);
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    return __int_as_float(__float_as_int(p) + (((int)n) << 23));
}
__device__ __forceinline__ float sigmoid_l(float z) { return 1.0f / (1.0f + expf_l(-z)); }
__device__ __forceinline__ float tanh_l(float x) {
    const float e = expf_l(2.0f * x);
    return (e - 1.0f) / (e + 1.0f);
}

struct LstmParams {
    const float* blob;  // the TF-V2 .data payload of model_LDP_200000_qpXX.dat, as stored
    float efs[5];       // [qp/51*0.18, onehot4(i_frame % 4)]
};

// float offsets into the blob per level (64, 32, 16): fc2_b, fc2_w, fc3_b, fc3_w, bias, kernel
__device__ __constant__ int kLstmOff[3][6] = {{723640, 723688, 727000, 727001, 727054, 727310},
                                              {578784, 578880, 591648, 591652, 592056, 592568},
                                              {0, 192, 50304, 50320, 53472, 54496}};

// Two launches per frame, both latency-oriented (few hundred CTUs, lock-step with the encoder):
//   k_lstm_cell : one block per (16 CG CTUs, hidden tile of 16 units), one wave per gate: its accumulator is a
//                 dependent chain of 2 N / 4 MFMA steps; all of its kernel rows (packed copy, one dwordx4 per lane per
//                 16 k) are requested into registers and the block's [x, h_prev] quads into LDS before the first
//                 MFMA; the four gates of a unit meet through 1 KB of LDS and wave q updates unit r = q of every
//                 lane; writes (c, h) to state_out.
//   k_lstm_heads: one block per (16 CTUs, level): wave j owns fc2 tile j with h_new read back from
//                 state_out (L2) as the B operand; h2 crosses waves through LDS in [tile][g][ctu][r]
//                 order (the writer's C-layout quad and the reader's B-operand quad of lane (ctu, g)
//                 are the same float4 slot); wave 0 finishes fc3 + sigmoid + gate predicates.
template <int LV>
struct LstmDims {
    static constexpr int N = (LV == 0) ? 64 : (LV == 1 ? 128 : 256);
    static constexpr int N2 = (LV == 0) ? 48 : (LV == 1 ? 96 : 192);
    static constexpr int N3 = (LV == 0) ? 1 : (LV == 1 ? 4 : 16);
    static constexpr int O1 = (LV == 0) ? 0 : (LV == 1 ? 64 : 192);
    static constexpr int O3 = (LV == 0) ? 0 : (LV == 1 ? 1 : 5);
    static constexpr int NT = N / 16, NT2 = N2 / 16;
};

// One gate (wave q) of hidden tile t for CG column groups of 16 CTUs: the kernel rows fetched once feed CG MFMAs.  What
// set this kernel's time was not bandwidth but one exposed memory latency per prefetch window (a 4- or 8-chunk-deep
// register prefetch: 20 us per 1080p frame with the weight loads, the x / h loads or the epilogue removed in turn);
// with every operand requested before the first MFMA it is paid once per block: 12 us.
// COH (the one-launch frame kernel): the new state is read by heads blocks of the SAME launch -> agent-scope stores.
template <int LV, int CG, bool COH = false>
__device__ __forceinline__ void lstm_cell(const LstmParams& lp, const float* __restrict__ vec, const float* __restrict__ state_in,
                                          float* __restrict__ state_out, int N_ctus, int group0, int lane, int q, int t,
                                          f32x4* xch, f32x4* xh) {
    using D = LstmDims<LV>;
    constexpr int N = D::N, O1 = D::O1, NC = 2 * N / 16;
    const int col = lane & 15, g = lane >> 4;
    const float* bk = lp.blob + kLstmOff[LV][4];
    // wave q owns gate q (i, j, f, o) of the tile: ONE accumulator per column group, the same chain as ever (k = 16 c + 4 g + e);
    // its A operands come from the packed copy of the kernel (ethcnn_spec.h): one dwordx4 per lane per chunk, 1 KB runs
    const float* kpk = lp.blob + kLstmBlobFloats + kLstmPackOff[LV] + (size_t)((t * 4 + q) * NC) * 256 + lane * 4;
    const float* xsrc[CG];
    const float* hsrc[CG];
    bool valid[CG];
    size_t row[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) {
        const int ctu_raw = (group0 + c) * 16 + col;
        valid[c] = ctu_raw < N_ctus;
        row[c] = (size_t)min(ctu_raw, N_ctus - 1);
        xsrc[c] = vec + row[c] * kNVec + O1 + 4 * g;
        hsrc[c] = state_in ? state_in + row[c] * 2 * kNVec + kNVec + O1 + 4 * g : xsrc[c];  // null state: any valid address, zeroed below
    }

    // every operand of the tile is requested before the first MFMA: the kernel rows of this gate into registers (NC dwordx4),
    // the block's [x, h_prev] quads into LDS (each wave fetches a quarter of the chunks for all four) -- one memory latency
    // per block instead of one per prefetch window
    float4 a[NC];
#pragma unroll
    for (int kc = 0; kc < NC; ++kc) a[kc] = *reinterpret_cast<const float4*>(kpk + (size_t)kc * 256);

#pragma unroll
    for (int i = 0; i < NC / 4; ++i) {
        const int kc = 4 * i + q;
        const bool in_x = kc < N / 16;
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            const float* src = in_x ? xsrc[c] + 16 * kc : hsrc[c] + 16 * (kc - N / 16);
            float4 v = *reinterpret_cast<const float4*>(src);
            if (!in_x && !state_in) v = make_float4(0.f, 0.f, 0.f, 0.f);
            xh[(c * NC + kc) * 64 + lane] = (f32x4){v.x, v.y, v.z, v.w};
        }
    }
    // ... and what the cell update needs behind the chain (the unit's four biases, its previous cell state): requested here (behind the
    // [x, h_prev] loads, which the staging barrier below waits for), not
    // behind the gate exchange's barrier, where they were one more exposed round trip
    const int u = 16 * t + 4 * g + q;
    const float b_i = bk[u], b_j = bk[N + u], b_f = bk[2 * N + u], b_o = bk[3 * N + u];
    float cprev[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) cprev[c] = state_in ? state_in[row[c] * 2 * kNVec + O1 + u] : 0.0f;
    __syncthreads();
    LSTM_STAMP(0, 1);
    f32x4 acc[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < NC; ++kc) {
        f32x4 hq[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) hq[c] = xh[(c * NC + kc) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                const float av = (e == 0) ? a[kc].x : (e == 1) ? a[kc].y : (e == 2) ? a[kc].z : a[kc].w;
                acc[c] = MFMA16(av, hq[c][e], acc[c]);
            }
    }
    // the four gates of lane (ctu, g) meet in LDS; wave q then updates unit r = q of every lane: u = 16 t + 4 g + q
#pragma unroll
    for (int c = 0; c < CG; ++c) xch[(c * 4 + q) * 64 + lane] = acc[c];
    __syncthreads();
    LSTM_STAMP(0, 2);
#pragma unroll
    for (int c = 0; c < CG; ++c) {
        const float* xf = reinterpret_cast<const float*>(xch + c * 256) + lane * 4 + q;
        const float gi = xf[0] + b_i, gj = xf[256] + b_j, gf = xf[512] + b_f, go = xf[768] + b_o;
        const float cp = cprev[c];
        float cc = sigmoid_l(gf + 1.0f) * cp + sigmoid_l(gi) * tanh_l(gj);
        cc = fminf(fmaxf(cc, -5.0f), 5.0f);
        const float hn = sigmoid_l(go) * tanh_l(cc);
        if (valid[c]) {
            float* so = state_out + row[c] * 2 * kNVec;
            if (COH) {
                __hip_atomic_store(so + O1 + u, cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(so + kNVec + O1 + u, hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                so[O1 + u] = cc;
                so[kNVec + O1 + u] = hn;
            }
        }
    }
}

// grid = (groups of 16 CG CTUs, 28 hidden tiles: 16 of level 16 first, then 8 of level 32, 4 of level 64); four waves per
// block, one per gate: the dependent MFMA chain of a wave is a quarter of the tile's (128 steps at level 16) and four times
// as many waves share the loads.  CG = 1 for small frames (parallelism), 2 from 720p up (weight reuse; 73 KB of LDS).
template <int CG>
__global__ __launch_bounds__(256) void k_lstm_cell(const float* __restrict__ vec, const float* __restrict__ state_in,
                                                   float* __restrict__ state_out, LstmParams lp, int N) {
    __shared__ f32x4 xch[CG * 4 * 64];
    __shared__ f32x4 xh[CG * 32 * 64];  // [x, h_prev] quads of the block: [column group][chunk <= 32][lane] (32 KB per group)
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int group0 = blockIdx.x * CG;
    const int ti = blockIdx.y;
    LSTM_STAMP(0, 0);
    if (ti < 16) lstm_cell<2, CG>(lp, vec, state_in, state_out, N, group0, lane, q, ti, xch, xh);
    else if (ti < 24) lstm_cell<1, CG>(lp, vec, state_in, state_out, N, group0, lane, q, ti - 16, xch, xh);
    else lstm_cell<0, CG>(lp, vec, state_in, state_out, N, group0, lane, q, ti - 24, xch, xh);
    LSTM_STAMP(0, 3);
}

template <int LV>
__device__ __forceinline__ void lstm_heads(const LstmParams& lp, const float* __restrict__ hrow, bool valid, int ctu,
                                           int lane, int wave, f32x4* h2T, float* __restrict__ raw,
                                           float* __restrict__ probs, int* flag32, int* flag16, float thr1, float thr2) {
    using D = LstmDims<LV>;
    constexpr int N = D::N, N2 = D::N2, N3 = D::N3, O1 = D::O1, O3 = D::O3, NT = D::NT, NT2 = D::NT2;
    const int col = lane & 15, g = lane >> 4;
    const float* b2 = lp.blob + kLstmOff[LV][0];
    const float* W2 = lp.blob + kLstmOff[LV][1];
    const float* b3 = lp.blob + kLstmOff[LV][2];
    const float* W3 = lp.blob + kLstmOff[LV][3];
    // fc2^T: wave j owns output tile j; step (t, r) consumes k = 16 t + 4 g + r.  Latency form: ALL operands of the tile (64 W2
    // values and 16 h quads per lane at level 16) are requested before the first MFMA -- fetched a step ahead, the loop paid
    // one L2 round trip per step with the GPU otherwise idle (round 3: 1080p LSTM heads 17 -> see profiles/r03_latency_ldp.txt)
    float w3v[NT2][4], w3e[4][5], b3v[4];  // wave 0: fc3 operands (requested behind its fc2 chain, used behind the exchange)
    if (wave < NT2) {
        const int j = wave;
        f32x4 a2 = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* wcol = W2 + 16 * j + col + (size_t)(4 * g) * N2;
        const float* hsrc = hrow + O1 + 4 * g;
        float wv[NT][4];
        float4 hq[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            hq[t] = *reinterpret_cast<const float4*>(hsrc + 16 * t);
#pragma unroll
            for (int r = 0; r < 4; ++r) wv[t][r] = wcol[(size_t)(16 * t + r) * N2];
        }
        float we[4][5], b2v[4];  // the efs rows + bias of this lane's four outputs
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int e = 0; e < 5; ++e) we[r][e] = W2[(N + e) * N2 + 16 * j + 4 * g + r];
            b2v[r] = b2[16 * j + 4 * g + r];
        }
        // every request above is issued before the first MFMA below waits for its operands: the ORDER is pinned -- left alone,
        // hipcc's scheduler sinks each load to just before its use (fewest live registers), i.e. one exposed round trip per
        // handful of MFMAs (lstm_probe: 7.8 us between "operands landed" and the h2 exchange for a 64-link chain)
        __builtin_amdgcn_sched_barrier(0);
#ifdef LSTM_STAMPS_FINE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LSTM_STAMP(1, 1);
#endif
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float hv[4] = {hq[t].x, hq[t].y, hq[t].z, hq[t].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) a2 = MFMA16(wv[t][r], hv[r], a2);
        }
        // wave 0 runs fc3^T behind the exchange: its operands (and the efs rows / bias of the fc3 epilogue) are requested HERE,
        // where the registers of the fc2 operands have just become free, so that their round trip passes under the fc2
        // epilogue and the wait for the other waves instead of standing behind the barrier
        if (wave == 0) {
            const int c3 = min(col, N3 - 1);  // (columns >= N3 feed zeros: loaded from a valid address, then dropped -- no branch per load)
#pragma unroll
            for (int jj = 0; jj < NT2; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r) w3v[jj][r] = W3[(16 * jj + 4 * g + r) * N3 + c3];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = min(4 * g + r, N3 - 1);
#pragma unroll
                for (int e = 0; e < 5; ++e) w3e[r][e] = W3[(N2 + e) * N3 + o];
                b3v[r] = b3[o];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = a2[r];
#pragma unroll
            for (int e = 0; e < 5; ++e) v = fmaf(lp.efs[e], we[r][e], v);
            a2[r] = lrelu_l(v + b2v[r]);
        }
        h2T[(j * 4 + g) * 16 + col] = a2;
    }
    __syncthreads();
#ifdef LSTM_STAMPS_FINE
    LSTM_STAMP(1, 2);
#else
    LSTM_STAMP(1, 1);
#endif
    if (wave == 0) {  // fc3^T
        f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
            const f32x4 hv = h2T[(j * 4 + g) * 16 + col];
#pragma unroll
            for (int r = 0; r < 4; ++r) z = MFMA16(col < N3 ? w3v[j][r] : 0.0f, hv[r], z);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 4 * g + r;
            if (o < N3 && valid) {
                float zz = z[r];
#pragma unroll
                for (int e = 0; e < 5; ++e) zz = fmaf(lp.efs[e], w3e[r][e], zz);
                const float p = sigmoid_l(zz + b3v[r]);
                const size_t idx = (size_t)ctu * kNOut + O3 + o;
                if (raw) raw[idx] = p;
                // agent-scope store (written through the XCD's L2): the block that applies the gates may run on another XCD
                // and overwrite this word; with plain stores every block would need an L2 write-back before its ticket
                __hip_atomic_store(&probs[idx], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (LV == 0 && p > thr1 && __hip_atomic_load(flag32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                    __hip_atomic_store(flag32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (LV == 1 && p > thr2 && __hip_atomic_load(flag16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                    __hip_atomic_store(flag16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// The end of a frame's heads work, called by every heads block (all its threads) once its probabilities and predicates are
// stored: ticket tree, gates by the block that draws the last ticket, completion word.  bid / nblk: this block's number among
// the nblk heads blocks of the frame.
__device__ __forceinline__ void lstm_finish_frame(int* gate, int chunks, int N, float thr2, float* __restrict__ probs, unsigned bid,
                                                  unsigned nblk, unsigned* done, unsigned done_seq, int* s_last) {
    int* const pred = gate;
    ETHCNN_HANDOFF_RELEASE();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned l1 = bid >> 3, n_l1 = (nblk + 7) >> 3, in_l1 = min(8u, nblk - 8u * l1);
        int* const root = gate + 2 * chunks + 32;  // (a fresh line behind the predicates)
        int* const leaf = root + 32 * (1 + (int)l1);
        int last = 0;
        if ((unsigned)__hip_atomic_fetch_add(leaf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == in_l1) {
            __hip_atomic_store(leaf, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)__hip_atomic_fetch_add(root, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == n_l1) {
                __hip_atomic_store(root, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        *s_last = last;
    }
    __syncthreads();
    LSTM_STAMP(1, 3);
    if (!*s_last) return;
    ETHCNN_HANDOFF_ACQUIRE();  // the last block reads every block's predicates and may overwrite their probabilities
    for (int ch = 0; ch < chunks; ++ch) {
        const bool open32 = __hip_atomic_load(pred + 2 * ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        // y16 is gated on the GATED y32: a closed L1 gate leaves zeros, and any(0 > thr2) decides (the 0 > thr2 corner)
        const bool open16 = open32 ? (__hip_atomic_load(pred + 2 * ch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) : (0.0f > thr2);
        if (open32 && open16) continue;
        const int c0 = ch * kSubBatch, cnt = min(N - c0, kSubBatch) * kNOut;
        for (int idx = threadIdx.x; idx < cnt; idx += (int)blockDim.x) {
            const int j = idx % kNOut;
            if (j != 0 && (j < 5 ? !open32 : !open16))
                __hip_atomic_store(&probs[(size_t)c0 * kNOut + idx], 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();  // every thread has read the predicates
    for (int i = threadIdx.x; i < 2 * chunks; i += (int)blockDim.x) __hip_atomic_store(pred + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // completion word (page-locked host memory): every block's probabilities had completed before its ticket, this block's
    // zero-fills complete here; the host thread spinning on the word sees the frame ~5 us before hipStreamSynchronize returns
    if (done) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// grid = (groups of 16 CTUs, level); 12 waves per block (levels 64 / 32 use 3 / 6 of them).
// The tf.cond gates (net():305,317 -- the AI path's k5_gate) are applied by the LAST block to finish: every block publishes
// its probabilities and predicates (agent-scope stores), takes a ticket, and the block that draws the launch's last ticket
// zero-fills the closed sub-batches (usually none) and hands the predicate words back as zeros for the next frame.  One
// launch instead of three (memset of the predicates, heads, gate) in a latency-bound chain.
__global__ __launch_bounds__(768) void k_lstm_heads(const float* __restrict__ state_out, LstmParams lp, int N, float thr1,
                                                    float thr2, float* __restrict__ raw, float* __restrict__ probs,
                                                    int* __restrict__ gate, unsigned* done, unsigned done_seq) {
    __shared__ f32x4 h2T[12 * 64];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ctu_raw = blockIdx.x * 16 + (lane & 15);
    const bool valid = ctu_raw < N;
    const int ctu = min(ctu_raw, N - 1);
    const float* hrow = state_out + (size_t)ctu * 2 * kNVec + kNVec;
    const int chunks = (N + kSubBatch - 1) / kSubBatch;
    int* const pred = gate;                   // two predicate words per mini-batch, then the ticket tree (lstm_gate_words)
    int* fl = pred + 2 * (ctu / kSubBatch);  // one frame: mini-batches of 1024 in raster order
    const int lv = 2 - (int)blockIdx.y;
    LSTM_STAMP(1, 0);
    if (lv == 0) lstm_heads<0>(lp, hrow, valid, ctu, lane, wave, h2T, raw, probs, fl, fl + 1, thr1, thr2);
    else if (lv == 1) lstm_heads<1>(lp, hrow, valid, ctu, lane, wave, h2T, raw, probs, fl, fl + 1, thr1, thr2);
    else lstm_heads<2>(lp, hrow, valid, ctu, lane, wave, h2T, raw, probs, fl, fl + 1, thr1, thr2);

    // This block's probabilities and predicates are agent-scope atomic stores: once they have completed (vmcnt 0) they are
    // visible to every XCD, so the ticket needs no release fence ("hand-offs inside a launch"; with plain stores
    // every block would need an L2 write-back: measured 36 -> 58 us for a 1080p frame).  The gate block only reads predicates
    // (atomic loads) and overwrites probabilities, so it needs no acquire either.
    // The ticket is a two-level TREE: all blocks of a frame finish within a microsecond of each other, and agent-scope
    // atomics on one word are served one at a time (~150 ns each: 96 blocks on a single ticket word kept the 1080p launch
    // open for ~14 us after its last MFMA).  Eight blocks share a first-level word (own 128-byte line), its finisher moves the
    // root; every word is reset by its finisher, so the tree is zero again when the launch ends.
#ifndef LSTM_STAMPS_FINE
    LSTM_STAMP(1, 2);
#endif
    lstm_finish_frame(gate, chunks, N, thr2, probs, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, done, done_seq, &s_last);
}

unsigned lstm_heads_blocks(int n) { return (unsigned)((n + 15) / 16) * 3u; }
// ints of the gate area of one frame of n CTUs: predicates + the ticket tree (root and one leaf per 8 blocks, a line each)
__host__ __device__ inline int lstm_gate_words_hd(int n) { return 2 * ((n + kSubBatch - 1) / kSubBatch) + 32 + 32 * (1 + (((n + 15) / 16) * 3 + 7) / 8); }
int lstm_gate_words(int n) { return lstm_gate_words_hd(n); }

// ================================================================ one launch per frame ======
// k_lstm_frame: the cell blocks and the heads blocks of a frame in ONE launch (a dataflow inside the grid, the scheme of the
// single-launch small pass, ethcnn_small.hip).  Why: as two launches the heads enter 1.5 us after the LAST cell block has left,
// load their weights then (3 us for the 256-unit level: 12 waves pull 200 KB through one CU's vector cache) and only then
// start their chains (scripts/ubench/lstm_probe.hip: cells out at 10-12 us, heads of level 16 done at 26).  Here
//   blocks [0, 28 gx)        cell items, as k_lstm_cell (item = tile index x gx + group pair; the 256-unit level first);
//   blocks [28 gx, + 3 G)    one level's heads of a group of 16 CTUs, FOUR waves that split the fc2 tiles (3 / 2 / 1 per wave)
//                            and are REGISTER-FED: a wave's fc2 operands are its own, one dwordx4 per lane per (tile, chunk)
//                            from the MFMA-operand-ordered copy of W2 behind the blob (ethcnn_spec.h) through an 8-chunk
//                            register ring, requested -- with the efs rows, biases and W3 staged into LDS -- BEFORE the block
//                            waits for the cells of its group and level; behind the wake-up: h quads (agent-scope), chains,
//                            exchange, fc3 by wave 0, ticket.
// Hand-offs as in DESIGN.md 3b: the new state is stored with agent scope, a cell block waits for its stores (vmcnt 0) before
// it adds to the completion counter of its (group pair, level); the adder that completes it resets it and raises the private
// flag (own 128-byte line) of each heads block of that pair and level.  FORWARD PROGRESS on a shared GPU: claim or execute --
// every cell item has a claim word tagged with the launch's epoch; a heads block that has waited 100 us executes the unclaimed
// cell items of its pair and level itself (same device function), a late cell block whose item is gone leaves.
// Same chains per accumulator as the two-launch form: bit-identical results (tests/test_gpu_lstm.py runs both).
#ifndef LSTM_FRAME_SHORT_FIRST
#define LSTM_FRAME_SHORT_FIRST 0  // 1: the short levels first in block order (A/B builds; measured the same end to end)
#endif
#ifndef LSTM_FRAME_RING
#define LSTM_FRAME_RING 6  // chunks of fc2 operands in registers (8: 0.7 us slower at 1080p, more spills)
#endif
constexpr int kLPad = 32;        // ints per 128-byte line
constexpr int kAuxSc1L = 16;     // buffer-instruction cache policy bit of an agent-scope access (sc1), as in ethcnn_fc1_tile.h
struct LstmFrameParams {
    const float* vec;
    const float* state_in;
    float* state_out;
    LstmParams lp;
    int N, groups, gx;        // CTUs, groups of 16, cell blocks per tile row (= ceil(groups / CG))
    float thr1, thr2;
    float *raw, *probs;
    int* sync;                // [lstm_gate_words: predicates + ticket tree][cell_done: 3 gx lines][heads_flag: 3 G lines][claim: 28 gx lines]
    int epoch;                // claim tag of this launch (never 0)
    int steal_test;           // tests: cell blocks with id % k == 1 leave without claiming, heads have no patience
    unsigned* done;
    unsigned done_seq;
};
__device__ __forceinline__ int lf_add(int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lf_put(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int lf_get(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// thread 0 polls the block's private flag (and resets it); budget > 0: give up after that many 100 MHz ticks
__device__ __forceinline__ bool lf_wait(int* p, unsigned long long budget) {
    unsigned long long t0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    while (lf_get(p) == 0) {
        __builtin_amdgcn_s_sleep(1);
        unsigned long long t;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        if (budget != 0 && t - t0 > budget) return false;
        if (t - t0 > 3000000000ull) __builtin_trap();  // 30 s of wall clock: a hung GPU, never a slow one
    }
    lf_put(p, 0);
    return true;
}
struct LstmFrameSync {
    int *cell_done, *heads_flag, *claim;
};
__host__ __device__ inline int lstm_frame_base(int n) { return (lstm_gate_words_hd(n) + kLPad - 1) / kLPad * kLPad; }
__device__ __forceinline__ LstmFrameSync lstm_frame_sync(int* sync, int n, int groups, int gx) {
    LstmFrameSync s;
    s.cell_done = sync + lstm_frame_base(n);
    s.heads_flag = s.cell_done + 3 * gx * kLPad;
    s.claim = s.heads_flag + 3 * groups * kLPad;
    return s;
}

// one cell item (tile index ti in [0, 28), group pair gpx): claim, compute, signal.  All threads of the block.
template <int LV, int CG>
__device__ __forceinline__ void lstm_frame_cell_item(const LstmFrameParams& P, const LstmFrameSync& Y, int t, int gpx, f32x4* xch, f32x4* xh,
                                                     int* s_flag) {
    constexpr int TI0 = (LV == 2) ? 0 : (LV == 1 ? 16 : 24), NT = LstmDims<LV>::NT;
    const int item = (TI0 + t) * P.gx + gpx;
    if (threadIdx.x == 0) *s_flag = (__hip_atomic_exchange(Y.claim + item * kLPad, P.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != P.epoch);
    __syncthreads();
    const bool mine = *s_flag != 0;
    __syncthreads();
    if (!mine) return;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    lstm_cell<LV, CG, true>(P.lp, P.vec, P.state_in, P.state_out, P.N, gpx * CG, lane, q, t, xch, xh);
    ETHCNN_HANDOFF_RELEASE();  // the tile's new state (agent-scope stores) has completed ...
    __syncthreads();
    if (threadIdx.x == 0 && lf_add(Y.cell_done + (gpx * 3 + LV) * kLPad, 1) + 1 == NT) {  // ... before the counter moves
        lf_put(Y.cell_done + (gpx * 3 + LV) * kLPad, 0);
        for (int c = 0; c < CG; ++c) {
            const int g = gpx * CG + c;
            if (g < P.groups) lf_put(Y.heads_flag + ((2 - LV) * P.groups + g) * kLPad, 1);
        }
    }
    __syncthreads();
}

// the same as a real CALL (the claim-or-execute path of the heads blocks): inlined there, the cell's 217 registers made the
// register allocator spill the heads' prefetched operands at their loads (a vmcnt(0) per load and scratch reloads between the
// chain's MFMAs); out of line only this rare path pays.  Everything by value: no reference into the kernarg segment.
template <int LV, int CG>
__device__ __attribute__((noinline)) void lstm_frame_cell_item_call(LstmFrameParams P, LstmFrameSync Y, int t, int gpx, f32x4* xch, f32x4* xh,
                                                                    int* s_flag) {
    lstm_frame_cell_item<LV, CG>(P, Y, t, gpx, xch, xh, s_flag);
}

// the heads of level LV for group g: 4 waves, TPW fc2 tiles each.  smem: the block's LDS (>= 30 KB).
template <int LV, int CG>
__device__ __forceinline__ void lstm_frame_heads(const LstmFrameParams& P, const LstmFrameSync& Y, int g, float* smem, f32x4* xch, f32x4* xh,
                                                 int* s_flag) {
    using D = LstmDims<LV>;
    constexpr int N = D::N, N2 = D::N2, N3 = D::N3, O1 = D::O1, O3 = D::O3, NT = D::NT, NT2 = D::NT2;
    constexpr int TPW = (NT2 + 3) / 4, RING = NT < LSTM_FRAME_RING ? NT : LSTM_FRAME_RING;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = lane & 15, gq = lane >> 4;
    const float* blob = P.lp.blob;
    const float* W2 = blob + kLstmOff[LV][1];
    const float* W3 = blob + kLstmOff[LV][3];
    // LDS: [h2 exchange NT2 x 256][efs rows of W2: 5 x N2][b2: N2][W3 incl. its efs rows: (N2 + 5) x N3][b3: N3]
    f32x4* const h2T = reinterpret_cast<f32x4*>(smem);
    float* const sE = smem + NT2 * 256;
    float* const sB2 = sE + 5 * N2;
    float* const sW3 = sB2 + N2;
    float* const sB3 = sW3 + (N2 + 5) * N3;
    const int j0 = wave * TPW;
    const __amdgpu_buffer_rsrc_t rWl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(blob) + kLstmBlobFloats + kLstmPackFc2Off[LV], 0, N * N2 * 4, 0x00020000);
    const int voff = lane * 16;
    int tile_off[TPW];
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) tile_off[jj] = __builtin_amdgcn_readfirstlane(min(j0 + jj, NT2 - 1) * NT * 1024);
    f32x4 wr[RING][TPW];
    int* const flag = Y.heads_flag + ((2 - LV) * P.groups + g) * kLPad;

#ifndef LSTM_FRAME_DELAY
#define LSTM_FRAME_DELAY 0  // 100 MHz ticks the heads blocks idle before they fetch (A/B builds; 600 / 1000 measured slower)
#endif
    // ---- everything that does not depend on the new state: the ring's first chunks into registers, the small operands into LDS
#define LF_FETCH()                                                                                                      \
    {                                                                                                                   \
        _Pragma("unroll") for (int t = 0; t < RING; ++t)                                                                \
            _Pragma("unroll") for (int jj = 0; jj < TPW; ++jj)                                                          \
                wr[t][jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rWl, voff, tile_off[jj] + t * 1024, 0)); \
        for (int i = threadIdx.x; i < 5 * N2; i += 256) sE[i] = W2[N * N2 + i];                                         \
        for (int i = threadIdx.x; i < N2; i += 256) sB2[i] = blob[kLstmOff[LV][0] + i];                                 \
        for (int i = threadIdx.x; i < (N2 + 5) * N3; i += 256) sW3[i] = W3[i];                                          \
        if (threadIdx.x < N3) sB3[threadIdx.x] = blob[kLstmOff[LV][2] + threadIdx.x];                                   \
    }
#define LF_WAIT(budget, result)                                                                                         \
    {                                                                                                                   \
        if (threadIdx.x == 0) *s_flag = lf_wait(flag, (budget)) ? 1 : 0;                                                \
        __syncthreads();                                                                                                \
        result = *s_flag != 0;                                                                                          \
        __syncthreads();                                                                                                \
        if (result) ETHCNN_HANDOFF_ACQUIRE();                                                                           \
    }
    // The block is resident ~10 us before the cells of its group are done and fetches during that wait (holding the fetch back
    // for the first 6 / 10 us, while the cell blocks' own operand loads are in flight, measured slower).
    bool woke = false;
    if (LSTM_FRAME_DELAY > 0 && P.steal_test == 0) LF_WAIT(LSTM_FRAME_DELAY, woke);
    LF_FETCH();
    if (!woke) LF_WAIT(P.steal_test ? 1 : 10000, woke);
    if (!woke) {
        // claim or execute: the unclaimed cell items of this group pair and level (they use the block's LDS; everything above is
        // fetched again afterwards, so nothing of it has to survive this rare path)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int gpx = g / CG;
#pragma unroll 1
        for (int t = 0; t < NT; ++t) lstm_frame_cell_item_call<LV, CG>(P, Y, t, gpx, xch, xh, s_flag);
        LF_WAIT(0, woke);  // every item of the pair and level is claimed by a resident block now
        LF_FETCH();
    }
#undef LF_FETCH
#undef LF_WAIT
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // (LDS operands staged by all threads)
    LSTM_STAMP(1, 1);

    // ---- behind the wake-up: the group's new h (agent-scope loads), the fc2 chains of this wave's tiles
    const int ctu_raw = g * 16 + col;
    const bool valid = ctu_raw < P.N;
    const int ctu = min(ctu_raw, P.N - 1);
    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(P.state_out, 0, -1, 0x00020000);
    const unsigned h_off = 4u * (unsigned)(ctu * 2 * kNVec + kNVec + O1 + 4 * gq);
    f32x4 hq[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) hq[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rH, h_off, t * 64, kAuxSc1L));
    f32x4 a2[TPW];
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) a2[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int slot = t % RING;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int jj = 0; jj < TPW; ++jj) a2[jj] = MFMA16(wr[slot][jj][r], hq[t][r], a2[jj]);
        if (t + RING < NT) {
#pragma unroll
            for (int jj = 0; jj < TPW; ++jj)
                wr[slot][jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rWl, voff, tile_off[jj] + (t + RING) * 1024, 0));
        }
        __builtin_amdgcn_sched_barrier(0);  // (the order above is the schedule)
    }
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) {
        const int j = j0 + jj;
        if (j < NT2) {  // wave-uniform
            f32x4 a = a2[jj];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int idx = 16 * j + 4 * gq + r;
                float v = a[r];
#pragma unroll
                for (int e = 0; e < 5; ++e) v = fmaf(P.lp.efs[e], sE[e * N2 + idx], v);
                a[r] = lrelu_l(v + sB2[idx]);
            }
            h2T[(j * 4 + gq) * 16 + col] = a;
        }
    }
    __syncthreads();
    int* const pred = P.sync;
    int* fl = pred + 2 * (ctu / kSubBatch);
    if (wave == 0) {  // fc3^T
        f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int c3 = min(col, N3 - 1);
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
            const f32x4 hv = h2T[(j * 4 + gq) * 16 + col];
#pragma unroll
            for (int r = 0; r < 4; ++r) z = MFMA16(col < N3 ? sW3[(16 * j + 4 * gq + r) * N3 + c3] : 0.0f, hv[r], z);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 4 * gq + r;
            if (o < N3 && valid) {
                float zz = z[r];
#pragma unroll
                for (int e = 0; e < 5; ++e) zz = fmaf(P.lp.efs[e], sW3[(N2 + e) * N3 + o], zz);
                const float p = sigmoid_l(zz + sB3[o]);
                const size_t idx = (size_t)ctu * kNOut + O3 + o;
                if (P.raw) P.raw[idx] = p;
                __hip_atomic_store(&P.probs[idx], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (LV == 0 && p > P.thr1 && __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                    __hip_atomic_store(fl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (LV == 1 && p > P.thr2 && __hip_atomic_load(fl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                    __hip_atomic_store(fl + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int CG>
__global__ __launch_bounds__(256, 2) void k_lstm_frame(LstmFrameParams P) {  // two blocks per CU: every block of a 1080p frame (544) is resident
    __shared__ f32x4 xch[CG * 4 * 64];
    __shared__ f32x4 xh[(CG * 32 * 64 < 2048 ? 2048 : CG * 32 * 64)];  // cell: [x, h_prev] quads; heads: exchange + small operands (<= 30 KB)
    __shared__ int s_flag, s_last;
    const LstmFrameSync Y = lstm_frame_sync(P.sync, P.N, P.groups, P.gx);
    const int bid = (int)blockIdx.x, ncell = 28 * P.gx;
    if (bid < ncell) {
        if (P.steal_test > 0 && bid % P.steal_test == 1) return;  // tests: "this block never got a slot"
        const int slot = bid / P.gx, gpx = bid - slot * P.gx;
        // dispatch order: the 16 tiles of 256 units first (the critical path: their heads finish last).  Short levels first lets
        // their cells and heads finish early but delays the long level by as much: same end (lstm_probe).
        const int ti = LSTM_FRAME_SHORT_FIRST ? (slot < 4 ? 24 + slot : (slot < 12 ? 16 + (slot - 4) : slot - 12)) : slot;
        LSTM_STAMP(0, 0);
        if (ti < 16) lstm_frame_cell_item<2, CG>(P, Y, ti, gpx, xch, xh, &s_flag);
        else if (ti < 24) lstm_frame_cell_item<1, CG>(P, Y, ti - 16, gpx, xch, xh, &s_flag);
        else lstm_frame_cell_item<0, CG>(P, Y, ti - 24, gpx, xch, xh, &s_flag);
        LSTM_STAMP(0, 3);
        return;
    }
    const int hb = bid - ncell, lvblk = hb / P.groups, g = hb - lvblk * P.groups;
    LSTM_STAMP(1, 0);
    float* const smem = reinterpret_cast<float*>(xh);
    if (lvblk == 0) lstm_frame_heads<2, CG>(P, Y, g, smem, xch, xh, &s_flag);
    else if (lvblk == 1) lstm_frame_heads<1, CG>(P, Y, g, smem, xch, xh, &s_flag);
    else lstm_frame_heads<0, CG>(P, Y, g, smem, xch, xh, &s_flag);
    const int chunks = (P.N + kSubBatch - 1) / kSubBatch;
    LSTM_STAMP(1, 2);
    lstm_finish_frame(P.sync, chunks, P.N, P.thr2, P.probs, (unsigned)hb, (unsigned)(3 * P.groups), P.done, P.done_seq, &s_last);
}

// ints of the one-launch kernel's sync area: predicates + ticket tree (lstm_gate_words), then counters, flags and claim words
int lstm_frame_words(int n) {
    const int groups = (n + 15) / 16, cg = groups >= 12 ? 2 : 1, gx = (groups + cg - 1) / cg;
    return lstm_frame_base(n) + (3 * gx + 3 * groups + 28 * gx) * kLPad;
}

void launch_lstm(const float* d_vec, const float* d_state_in, float* d_state_out, const float* d_lstm_blob, int n, int qp,
                 int i_frame, float thr1, float thr2, float* d_raw, float* d_probs, int* d_gate, unsigned* done, unsigned done_seq,
                 int one_launch, int epoch, hipStream_t s) {
    LstmParams lp;
    lp.blob = d_lstm_blob;
    lp.efs[0] = ((float)qp / 51.0f) * 0.18f;  // net():283  qp / 51.0 * 0.18
    const int phase = ((i_frame % 4) + 4) % 4;
    for (int e = 0; e < 4; ++e) lp.efs[1 + e] = (e == phase) ? 1.0f : 0.0f;
    const unsigned groups = (unsigned)((n + 15) / 16);
    if (one_launch) {  // cells + heads as one dataflow launch (k_lstm_frame); d_gate holds lstm_frame_words(n) ints
        LstmFrameParams P;
        P.vec = d_vec;
        P.state_in = d_state_in;
        P.state_out = d_state_out;
        P.lp = lp;
        P.N = n;
        P.groups = (int)groups;
        const int cg = groups >= 12 ? 2 : 1;
        P.gx = ((int)groups + cg - 1) / cg;
        P.thr1 = thr1;
        P.thr2 = thr2;
        P.raw = d_raw;
        P.probs = d_probs;
        P.sync = d_gate;
        P.epoch = epoch;
        static const int steal_test = [] { const char* e = dev_env("ETHCNN_LSTM_STEAL_TEST"); return e ? atoi(e) : 0; }();  // tests
        P.steal_test = steal_test > 1 ? steal_test : 0;
        P.done = done;
        P.done_seq = done_seq;
        const unsigned blocks = 28u * (unsigned)P.gx + 3u * groups;
        if (cg == 2) hipLaunchKernelGGL(k_lstm_frame<2>, dim3(blocks), dim3(256), 0, s, P);
        else hipLaunchKernelGGL(k_lstm_frame<1>, dim3(blocks), dim3(256), 0, s, P);
        return;
    }
    if (groups >= 12) hipLaunchKernelGGL(k_lstm_cell<2>, dim3((groups + 1) / 2, 28), dim3(256), 0, s, d_vec, d_state_in, d_state_out, lp, n);
    else hipLaunchKernelGGL(k_lstm_cell<1>, dim3(groups, 28), dim3(256), 0, s, d_vec, d_state_in, d_state_out, lp, n);
    hipLaunchKernelGGL(k_lstm_heads, dim3(groups, 3), dim3(768), 0, s, d_state_out, lp, n, thr1, thr2, d_raw, d_probs, d_gate, done, done_seq);
}

}  // namespace ethcnn
