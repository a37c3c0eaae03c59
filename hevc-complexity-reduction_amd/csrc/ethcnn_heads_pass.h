// ethcnn_heads_pass.h -- device code of one head (FC2 + FC3 + sigmoid + gate predicates) for a wave's 16 CTUs; see
// ethcnn_heads.hip for the design notes.  Shared by k_heads (ethcnn_heads.hip), the single-launch small pass (ethcnn_small.hip)
// and the LSTM heads (ethcnn_lstm.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "ethcnn_fc1_tile.h"  // typedefs, MFMA16, kAuxSc1
#include "ethcnn_kernels.h"

namespace ethcnn {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lrelu_h(float h) { return fmaxf(0.2f * h, h); }

__device__ __forceinline__ float expf_canonical_h(float x) {
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

#ifdef HEADS_STAMPS
// development probe (scripts/ubench/heads_probe.hip): stamps of wave 0 of every block -- s_memtime (shader clock, per
// XCC) at the phase boundaries, s_memrealtime (100 MHz, device-wide) at entry and exit
__device__ unsigned long long g_heads_stamps[1 << 16][8];
__device__ __forceinline__ void heads_stamp(int slot) {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    if (threadIdx.x == 0) g_heads_stamps[(blockIdx.y * gridDim.x + blockIdx.x) & 0xffff][slot] = t;
    if (slot == 0 || slot == 4) {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        if (threadIdx.x == 0) g_heads_stamps[(blockIdx.y * gridDim.x + blockIdx.x) & 0xffff][slot == 0 ? 6 : 7] = t;
    }
}
#define HEADS_STAMP(i) heads_stamp(i)
#else
#define HEADS_STAMP(i)
#endif

struct HeadsParams {
    const float* w2[3];
    const float* w2lane[3];  // DeviceWeights::fc2_lane (single-launch pass only)
    const float* b2[3];
    const float* w3[3];
    const float* b3[3];
};

// compile-time description of head H: 0/1/2 -> (n1, n2, n3) = (64,48,1) / (128,96,4) / (256,192,16)
template <int H>
struct Hd {
    static constexpr int N1 = (H == 0) ? 64 : (H == 1 ? 128 : 256);
    static constexpr int N2 = (H == 0) ? 48 : (H == 1 ? 96 : 192);
    static constexpr int N3 = (H == 0) ? 1 : (H == 1 ? 4 : 16);
    static constexpr int O1 = (H == 0) ? 0 : (H == 1 ? 64 : 192);
    static constexpr int O2 = (H == 0) ? 0 : (H == 1 ? 48 : 144);
    static constexpr int O3 = (H == 0) ? 0 : (H == 1 ? 1 : 5);
    static constexpr int NT = N2 / 16;             // FC2 output tiles (3 / 6 / 12)
    static constexpr int NK = N1 / 16;             // 16-k chunks (4 / 8 / 16)
    static constexpr int B_FLOATS = 16 * N2;       // one W2 chunk
    static constexpr int B_INST = B_FLOATS / 256;  // 3 / 6 / 12 LDS-DMA instructions
    static constexpr int B_PER = (B_INST + 3) / 4; // per wave (the tail duplicates the last piece)
    static constexpr int ISSUE = B_PER + 1;        // VMEM ops per wave per iteration (+ its h1 piece)
    static constexpr bool COLSWZ = (N2 % 32 == 0);
    // head 16's W2 chunk fills a whole 12 KB stage, so its h1 quads go straight into registers (buffer_load, a chunk
    // ahead); heads 32 / 64 (6 / 3 KB chunks) keep theirs in the stage behind the chunk, by LDS-DMA
    static constexpr bool H1REG = (H == 2);
    static constexpr int H1_AT = 16 * N2;          // float offset of the 4 waves' h1 pieces inside a stage (not H1REG)
};
constexpr int kHeadsStage = 16 * 192;  // floats per LDS stage: the widest W2 chunk (12 KB) = chunk + h1 pieces of the others
constexpr int kHeadsStages = 2;  // prefetch distance 1: 24 KB of LDS, < 80 VGPRs per block -> 6 blocks per CU

// One head for this wave's 16 CTUs.  2 LDS stages, prefetch distance 1, W2 by LDS-DMA (inline asm: hipcc neither
// drains nor counts it), explicit vmcnt + raw barrier -- the FC1 pipeline of ethcnn_dense.hip at the heads' sizes;
// occupancy, not depth, covers the DMA latency (3 stages / 3 blocks per CU measured 7 % slower on 102,000 CTUs;
// 6 blocks per CU instead of 4: stage alone 143.6 -> 131.2 us, in the pipeline 0.157 -> 0.148 ms).
// (Round 3's two merged launch plans -- FC1 + heads + gates as one launch, and the gates applied by the heads launch -- read h1 /
// stored the probabilities with agent-scope accesses here; measured equal to 1 % slower than the separate launches, removed in round 6.)
template <int H>
__device__ __forceinline__ void head_pass(float* smem, const float* __restrict__ H1, const HeadsParams& hp, float qn,
                                          int lane, unsigned wvu, bool valid, int ctu, float* __restrict__ h2row,
                                          float* __restrict__ logits, float* __restrict__ raw, float* __restrict__ probs,
                                          int* flag32, int* flag16, float thr1, float thr2) {
    using D = Hd<H>;
    const int col = lane & 15, g = lane >> 4;
    const float* W2 = hp.w2[H];
    const float* W3 = hp.w3[H];
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)smem);

    // DMA sources.  W2 chunk: permuted LDS image (rows / column groups with odd (k>>2) swapped, as in
    // ethcnn_dense.hip) applied to the per-lane source offset.  h1 piece: lane (ctu, g) fetches
    // its own float4 h1[ctu][O1 + 16 kc + 4 g ..], landing linearly at lane * 16 B.
    // every DMA source = wave-uniform base in SGPRs + a 32-bit per-lane byte offset in one VGPR (the saddr form of
    // global_load_lds_dwordx4; scalar per-chunk advance -- see ethcnn_dense.hip).  h1 offsets stay < 2^32 bytes
    // (<= 131072 CTUs per pass x 1792 B).
    unsigned b_off[D::B_PER];
#pragma unroll
    for (int i = 0; i < D::B_PER; ++i) {
        const int q = min((int)wvu + i * 4, D::B_INST - 1);
        const int e = q * 64 + lane;
        int row = (e / (D::N2 / 4)) % 16;
        int c4 = e % (D::N2 / 4);
        if (D::COLSWZ) c4 ^= ((row >> 2) & 1) << 2;
        else row ^= (row >> 2) & 1;
        b_off[i] = 4u * (unsigned)(row * D::N2 + c4 * 4);
    }
    const unsigned a_off = 4u * (unsigned)(ctu * kNVec + D::O1 + 4 * g);
    // A-operand reads: W2[k = 4 g + e][n = 16 j + col] of the chunk sits at  a_base[sel] + e N2 + 16 j  with two per-lane bases
    // (the permutation above moves odd-g lanes by +-16 columns, sel = j & 1, or by +-1 row, sel = e & 1): everything else
    // is an immediate offset of the ds_read -- no address VALU in the K loop
    int a_base[2];
    if (D::COLSWZ) { a_base[0] = 4 * g * D::N2 + col + 16 * (g & 1); a_base[1] = 4 * g * D::N2 + col - 16 * (g & 1); }
    else { a_base[0] = 4 * g * D::N2 + col + D::N2 * (g & 1); a_base[1] = 4 * g * D::N2 + col - D::N2 * (g & 1); }

#define HP_DMA(voff, sbase, lds_byte_off)                                                                \
    {                                                                                                    \
        unsigned keep_;                                                                                  \
        const unsigned dst_ = __builtin_amdgcn_readfirstlane(lds_base + (lds_byte_off));                 \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(voff), "s"(sbase), "s"(dst_) : "memory");                      \
    }
#define HP_ISSUE(kc, st)                                                                                 \
    {                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < D::B_PER; ++i)                                             \
            HP_DMA(b_off[i], W2 + (size_t)(kc) * 16 * D::N2,                                             \
                   4u * ((st) * kHeadsStage + min(wvu + i * 4, (unsigned)(D::B_INST - 1)) * 256));       \
        if (!D::H1REG) HP_DMA(a_off, H1 + (kc) * 16, 4u * ((st) * kHeadsStage + D::H1_AT + wvu * 256))   \
    }
#define HP_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

    f32x4 acc[D::NT];
#pragma unroll
    for (int j = 0; j < D::NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const __amdgpu_buffer_rsrc_t rH1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(H1), 0, -1, 0x00020000);
    constexpr int kH1Aux = 0;
    f32x4 avr = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (D::H1REG) avr = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rH1, a_off, 0, kH1Aux));
    __builtin_amdgcn_s_barrier();  // the previous head's last stage has been consumed by every wave
    HEADS_STAMP(1);
    HP_ISSUE(0, 0);
    HP_WAIT(0);
    __builtin_amdgcn_s_barrier();
    HEADS_STAMP(2);
    int st = 0;
#pragma unroll 1
    for (int kc = 0; kc < D::NK; ++kc) {
        const int st2 = st ^ 1;
        if (kc + 1 < D::NK) { HP_ISSUE(kc + 1, st2); }
        f32x4 av, avn = avr;
        if (D::H1REG) {
            av = avr;
            if (kc + 1 < D::NK) avn = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rH1, a_off, (kc + 1) * 64, kH1Aux));
            asm volatile("" ::: "memory");  // the prefetch stays ahead of this chunk's MFMAs
        } else {
            av = *reinterpret_cast<const f32x4*>(smem + st * kHeadsStage + D::H1_AT + wvu * 256 + lane * 4);
        }
        const float* bsE = smem + st * kHeadsStage + a_base[0];
        const float* bsO = smem + st * kHeadsStage + a_base[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float hv = av[e];
#pragma unroll
            for (int j = 0; j < D::NT; ++j)
                acc[j] = MFMA16((((D::COLSWZ ? j : e) & 1) ? bsO : bsE)[e * D::N2 + 16 * j], hv, acc[j]);  // rows = W2 columns
        }
        HP_WAIT(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        st = st2;
        avr = avn;
    }
#undef HP_DMA
#undef HP_ISSUE
#undef HP_WAIT
    HEADS_STAMP(3);

    // FC2 epilogue in place: lane (ctu = col, g) holds h2[ctu][16 j + 4 g + r].  Small operand fetches below go through
    // buffer instructions (SGPR resource + one VGPR offset): cheaper to issue beside MFMAs than 64-bit VGPR addresses
    const __amdgpu_buffer_rsrc_t rW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W2), 0, (D::N1 + 1) * D::N2 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hp.b2[H]), 0, D::N2 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W3), 0, (D::N2 + 1) * D::N3 * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < D::NT; ++j) {
        const int n = 16 * j + 4 * g;
        const f32x4 wq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW2, 16 * g, (D::N1 * D::N2 + 16 * j) * 4, 0));
        const f32x4 bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB2, 16 * g, 64 * j, 0));
        acc[j][0] = lrelu_h(fmaf(qn, wq.x, acc[j][0]) + bv.x);
        acc[j][1] = lrelu_h(fmaf(qn, wq.y, acc[j][1]) + bv.y);
        acc[j][2] = lrelu_h(fmaf(qn, wq.z, acc[j][2]) + bv.z);
        acc[j][3] = lrelu_h(fmaf(qn, wq.w, acc[j][3]) + bv.w);
        if (valid && h2row) *reinterpret_cast<f32x4*>(h2row + D::O2 + n) = acc[j];
    }
    // FC3^T: rows = outputs (N3 of 16 used), columns = CTUs; step (j, r) consumes k = 16 j + 4 g + r
    f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
    {   // W3 operands fetched one tile ahead of their use (8 VGPRs instead of 4 NT); columns >= N3 read as 0
        float wc[4], wn[4];
        const int w3off = (4 * g * D::N3 + col) * 4;  // lane part of W3[(16 j + 4 g + r) * N3 + col]
#pragma unroll
        for (int r = 0; r < 4; ++r)
            wc[r] = (col < D::N3) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW3, w3off, r * D::N3 * 4, 0)) : 0.0f;
#pragma unroll
        for (int j = 0; j < D::NT; ++j) {
            if (j + 1 < D::NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    wn[r] = (col < D::N3) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW3, w3off, (16 * (j + 1) + r) * D::N3 * 4, 0)) : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) z = MFMA16(wc[r], acc[j][r], z);
#pragma unroll
            for (int r = 0; r < 4; ++r) wc[r] = wn[r];
        }
    }
    // lane (ctu = col, g) holds outputs 4 g + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 4 * g + r;
        if (o < D::N3 && valid) {
            const float zz = fmaf(qn, W3[D::N2 * D::N3 + o], z[r]) + hp.b3[H][o];
            const float p = 1.0f / (1.0f + expf_canonical_h(-zz));
            const size_t idx = (size_t)ctu * kNOut + D::O3 + o;
            if (logits) logits[idx] = zz;  // introspection copies (ethcnn_set_debug_capture), null in production
            if (raw) raw[idx] = p;
            probs[idx] = p;
            if (H == 0 && p > thr1 && __hip_atomic_load(flag32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __hip_atomic_store(flag32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // any(y64 > THR_L1_LOWER)
            if (H == 1 && p > thr2 && __hip_atomic_load(flag16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __hip_atomic_store(flag16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // any(y32_tmp > THR_L2_LOWER)
        }
    }
}


// ---- latency form for ONE group of 16 CTUs (the single-launch small pass, ethcnn_small.hip): the block's waves SPLIT the FC2
// output tiles of the head (head 16: 12 tiles -> 3 per wave; 32: 2, 2, 2, -; 64: 1, 1, 1, -) instead of each taking 16
// CTUs with all tiles: a wave's K loop is a quarter as long (head 16: 192 MFMAs instead of 768 -- the 64-CTU form keeps one
// SIMD busy for 10 us, and a picture's heads blocks have the GPU to themselves).
// REGISTER-FED like the pass's FC1 tile: a wave's W2 operands are its OWN (its tiles' columns), so they need neither LDS nor
// barriers -- one dwordx4 load per lane per (tile, 16-k chunk) from the MFMA-operand-ordered copy of W2 (HeadsParams::w2lane).
// Heads 64 and 32 request everything -- all h1 quads (agent-scope), all W2 pieces, epilogue operands -- in ONE round trip; head
// 16 (48 pieces per wave) runs its pieces through a ring of D chunks.  The first form (W2 chunks through a 3-stage LDS ring
// shared by the block, barrier per chunk) paid half a DMA round trip per chunk: 16 x ~0.5 us for head 16, the launch's
// critical path.  h2 crosses waves through LDS in [tile][lane] order (the writer's C-layout quad of lane (ctu, g) is the
// reader's B-operand quad, as in k_lstm_heads); wave 0 runs FC3 + sigmoid + gate predicates, its W3 operands requested while the
// ring drains.  Same chains per accumulator: same results.  smem: NT * 256 floats (the h2 exchange).
template <int H>
__device__ __forceinline__ void head_pass_regs(float* smem, const float* __restrict__ H1, const HeadsParams& hp, float qn, int lane,
                                               unsigned wvu, bool valid, int ctu, float* __restrict__ h2row,
                                               float* __restrict__ logits, float* __restrict__ raw, float* __restrict__ probs,
                                               int* flag32, int* flag16, float thr1, float thr2) {
    using D = Hd<H>;
    constexpr int TPW = (D::NT + 3) / 4;                     // tiles per wave
    constexpr int RING = (D::NK <= 8) ? D::NK : 6;           // chunks of W2 pieces in registers (heads 64 / 32: all of them)
    const int col = lane & 15, g = lane >> 4;
    const float* W3 = hp.w3[H];
    f32x4* const h2T = reinterpret_cast<f32x4*>(smem);
    const int j0 = (int)wvu * TPW;  // this wave's tiles j0 .. j0 + TPW - 1 (those < NT; idle slots recompute the last tile, never stored)
    const unsigned a_off = 4u * (unsigned)(ctu * kNVec + D::O1 + 4 * g);
    const __amdgpu_buffer_rsrc_t rH1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(H1), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rWl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hp.w2lane[H]), 0, D::N1 * D::N2 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hp.w2[H]), 0, (D::N1 + 1) * D::N2 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hp.b2[H]), 0, D::N2 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W3), 0, (D::N2 + 1) * D::N3 * 4, 0x00020000);
    const int voff = lane * 16;
    int tile_off[TPW];  // byte offset of tile (j0 + jj)'s pieces: [tile][chunk][64 lanes][4]
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) tile_off[jj] = __builtin_amdgcn_readfirstlane(min(j0 + jj, D::NT - 1) * D::NK * 1024);

    // ---- one round trip: h1 quads, the first RING chunks of W2 pieces, the FC2 epilogue operands (qp row of W2, bias)
    f32x4 hq[D::NK];
#pragma unroll
    for (int kc = 0; kc < D::NK; ++kc) hq[kc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rH1, a_off, kc * 64, kAuxSc1));
    f32x4 wr[RING][TPW];
#pragma unroll
    for (int kc = 0; kc < RING; ++kc)
#pragma unroll
        for (int jj = 0; jj < TPW; ++jj) wr[kc][jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rWl, voff, tile_off[jj] + kc * 1024, 0));
    f32x4 wq[TPW], bv[TPW];
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) {
        const int j = min(j0 + jj, D::NT - 1);
        wq[jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW2, 16 * g, (D::N1 * D::N2 + 16 * j) * 4, 0));
        bv[jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB2, 16 * g, 64 * j, 0));
    }
    float w3q[D::NT][4], w3p[4], b3v[4];
    f32x4 acc[TPW];
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) acc[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kc = 0; kc < D::NK; ++kc) {
        const int slot = kc % RING;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int jj = 0; jj < TPW; ++jj) acc[jj] = MFMA16(wr[slot][jj][e], hq[kc][e], acc[jj]);
        if (kc + RING < D::NK) {
#pragma unroll
            for (int jj = 0; jj < TPW; ++jj)
                wr[slot][jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rWl, voff, tile_off[jj] + (kc + RING) * 1024, 0));
        }
        // wave 0 runs FC3 at the end: ALL of its W3 operands are requested as soon as the ring stops refilling (registers free
        // up from there on), so that they are there when the K loop ends; columns >= N3 read as 0
        if (kc == D::NK - RING && wvu == 0) {
            const int w3off = (4 * g * D::N3 + col) * 4;  // lane part of W3[(16 j + 4 g + r) * N3 + col]
#pragma unroll
            for (int j = 0; j < D::NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    w3q[j][r] = (col < D::N3) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW3, w3off, (16 * j + r) * D::N3 * 4, 0)) : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // FC3 epilogue: the qp row of W3 and the bias of output o = 4 g + r
                const int o = min(4 * g + r, D::N3 - 1);
                w3p[r] = W3[D::N2 * D::N3 + o];
                b3v[r] = hp.b3[H][o];
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // (the order above is the schedule: see fc1_tile_regs)
    }
    // FC2 epilogue of this wave's tiles, then hand them to wave 0
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) {
        const int j = j0 + jj;
        if (j < D::NT) {  // wave-uniform
            f32x4 a = acc[jj];
            a[0] = lrelu_h(fmaf(qn, wq[jj].x, a[0]) + bv[jj].x);
            a[1] = lrelu_h(fmaf(qn, wq[jj].y, a[1]) + bv[jj].y);
            a[2] = lrelu_h(fmaf(qn, wq[jj].z, a[2]) + bv[jj].z);
            a[3] = lrelu_h(fmaf(qn, wq[jj].w, a[3]) + bv[jj].w);
            if (valid && h2row) *reinterpret_cast<f32x4*>(h2row + D::O2 + 16 * j + 4 * g) = a;
            h2T[j * 64 + lane] = a;
        }
    }
    __syncthreads();
    if (wvu != 0) return;
    f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < D::NT; ++j) {
        const f32x4 hv = h2T[j * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) z = MFMA16(w3q[j][r], hv[r], z);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 4 * g + r;
        if (o < D::N3 && valid) {
            const float zz = fmaf(qn, w3p[r], z[r]) + b3v[r];
            const float p = 1.0f / (1.0f + expf_canonical_h(-zz));
            const size_t idx = (size_t)ctu * kNOut + D::O3 + o;
            if (logits) logits[idx] = zz;
            if (raw) raw[idx] = p;
            __hip_atomic_store(&probs[idx], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (H == 0 && p > thr1 && __hip_atomic_load(flag32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __hip_atomic_store(flag32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (H == 1 && p > thr2 && __hip_atomic_load(flag16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __hip_atomic_store(flag16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


// ---- the same split with the W2 chunks through a 3-stage LDS ring shared by the block (the first form; kept for pictures of more
// than 1536 CTUs, where up to 1600 blocks share 512 slots and it measures 7 us faster at 2160p than the register-fed form above): the block's waves SPLIT the FC2
// output tiles of the head (head 16: 12 tiles -> 3 per wave; 32: 2, 2, 2, -; 64: 1, 1, 1, -) instead of each taking 16
// CTUs with all tiles: a wave's K loop is a quarter as long (head 16: 192 MFMAs instead of 768 -- the 64-CTU form keeps one
// SIMD busy for 10 us, and a picture's heads blocks have the GPU to themselves).  Every wave requests all h1 quads of the
// group up front (agent-scope loads), the W2 chunks run through the 3-stage LDS ring shared by the block, h2 crosses waves
// through LDS in [tile][lane] order (the writer's C-layout quad of lane (ctu, g) is the reader's B-operand quad, as in
// k_lstm_heads) and wave 0 runs FC3 + sigmoid + gate predicates.  Same chains per accumulator: same results.
// smem: kHeadsLatStages * kHeadsStage floats of W2 stages + NT * 256 floats of h2.
constexpr int kHeadsLatStages = 3;
template <int H>
__device__ __forceinline__ void head_pass_split(float* smem, const float* __restrict__ H1, const HeadsParams& hp, float qn, int lane,
                                                unsigned wvu, bool valid, int ctu, float* __restrict__ h2row,
                                                float* __restrict__ logits, float* __restrict__ raw, float* __restrict__ probs,
                                                int* flag32, int* flag16, float thr1, float thr2) {
    using D = Hd<H>;
    constexpr int TPW = (D::NT + 3) / 4;                     // tiles per wave
    const int col = lane & 15, g = lane >> 4;
    const float* W2 = hp.w2[H];
    const float* W3 = hp.w3[H];
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)smem);
    f32x4* const h2T = reinterpret_cast<f32x4*>(smem + kHeadsLatStages * kHeadsStage);
    unsigned b_off[D::B_PER];
#pragma unroll
    for (int i = 0; i < D::B_PER; ++i) {  // the W2 chunk's permuted LDS image, as in head_pass
        const int q = min((int)wvu + i * 4, D::B_INST - 1);
        const int e = q * 64 + lane;
        int row = (e / (D::N2 / 4)) % 16;
        int c4 = e % (D::N2 / 4);
        if (D::COLSWZ) c4 ^= ((row >> 2) & 1) << 2;
        else row ^= (row >> 2) & 1;
        b_off[i] = 4u * (unsigned)(row * D::N2 + c4 * 4);
    }
    const unsigned a_off = 4u * (unsigned)(ctu * kNVec + D::O1 + 4 * g);
    int a_base[2];
    if (D::COLSWZ) { a_base[0] = 4 * g * D::N2 + col + 16 * (g & 1); a_base[1] = 4 * g * D::N2 + col - 16 * (g & 1); }
    else { a_base[0] = 4 * g * D::N2 + col + D::N2 * (g & 1); a_base[1] = 4 * g * D::N2 + col - D::N2 * (g & 1); }
    const int j0 = (int)wvu * TPW;  // this wave's tiles j0 .. j0 + TPW - 1 (those < NT)
#define HS_DMA(voff, sbase, lds_byte_off)                                                                \
    {                                                                                                    \
        unsigned keep_;                                                                                  \
        const unsigned dst_ = __builtin_amdgcn_readfirstlane(lds_base + (lds_byte_off));                 \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(voff), "s"(sbase), "s"(dst_) : "memory");                      \
    }
#define HS_ISSUE(kc, st)                                                                                 \
    {                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < D::B_PER; ++i)                                             \
            HS_DMA(b_off[i], W2 + (size_t)(kc) * 16 * D::N2,                                             \
                   4u * ((st) * kHeadsStage + min(wvu + i * 4, (unsigned)(D::B_INST - 1)) * 256));       \
    }
    const __amdgpu_buffer_rsrc_t rH1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(H1), 0, -1, 0x00020000);
    f32x4 hq[D::NK];
#pragma unroll
    for (int kc = 0; kc < D::NK; ++kc) hq[kc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rH1, a_off, kc * 64, kAuxSc1));
    // wave 0 runs FC3 at the end: ALL of its W3 operands are requested now (a tile-ahead prefetch would expose one L2 round
    // trip per tile -- 12 x ~0.6 us for head 16 -- when nothing else runs on the CU); columns >= N3 read as 0
    const __amdgpu_buffer_rsrc_t rW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W3), 0, (D::N2 + 1) * D::N3 * 4, 0x00020000);
    float w3q[D::NT][4];
    if (wvu == 0) {
        const int w3off = (4 * g * D::N3 + col) * 4;  // lane part of W3[(16 j + 4 g + r) * N3 + col]
#pragma unroll
        for (int j = 0; j < D::NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                w3q[j][r] = (col < D::N3) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW3, w3off, (16 * j + r) * D::N3 * 4, 0)) : 0.0f;
    }
    f32x4 acc[TPW];
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) acc[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    HS_ISSUE(0, 0);
    if (D::NK > 1) { HS_ISSUE(1, 1); }
    if (D::NK > 1) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D::B_PER) : "memory"); }  // all but the newest group: h1 + chunk 0
    else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int kc = 0; kc < D::NK; ++kc) {
        const int st = kc % kHeadsLatStages;
        if (kc + 2 < D::NK) { HS_ISSUE(kc + 2, (kc + 2) % kHeadsLatStages); }
        const float* bsE = smem + st * kHeadsStage + a_base[0];
        const float* bsO = smem + st * kHeadsStage + a_base[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float hv = hq[kc][e];
#pragma unroll
            for (int jj = 0; jj < TPW; ++jj) {
                const int j = min(j0 + jj, D::NT - 1);  // (idle slots recompute the last tile; never stored)
                acc[jj] = MFMA16((((D::COLSWZ ? j : e) & 1) ? bsO : bsE)[e * D::N2 + 16 * j], hv, acc[jj]);
            }
        }
        if (kc + 2 < D::NK) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D::B_PER) : "memory"); }
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#undef HS_DMA
#undef HS_ISSUE
    // FC2 epilogue of this wave's tiles, then hand them to wave 0
    const __amdgpu_buffer_rsrc_t rW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W2), 0, (D::N1 + 1) * D::N2 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hp.b2[H]), 0, D::N2 * 4, 0x00020000);
#pragma unroll
    for (int jj = 0; jj < TPW; ++jj) {
        const int j = j0 + jj;
        if (j < D::NT) {  // wave-uniform
            const f32x4 wq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW2, 16 * g, (D::N1 * D::N2 + 16 * j) * 4, 0));
            const f32x4 bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB2, 16 * g, 64 * j, 0));
            f32x4 a = acc[jj];
            a[0] = lrelu_h(fmaf(qn, wq.x, a[0]) + bv.x);
            a[1] = lrelu_h(fmaf(qn, wq.y, a[1]) + bv.y);
            a[2] = lrelu_h(fmaf(qn, wq.z, a[2]) + bv.z);
            a[3] = lrelu_h(fmaf(qn, wq.w, a[3]) + bv.w);
            if (valid && h2row) *reinterpret_cast<f32x4*>(h2row + D::O2 + 16 * j + 4 * g) = a;
            h2T[j * 64 + lane] = a;
        }
    }
    __syncthreads();
    if (wvu != 0) return;
    f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < D::NT; ++j) {
        const f32x4 hv = h2T[j * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) z = MFMA16(w3q[j][r], hv[r], z);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 4 * g + r;
        if (o < D::N3 && valid) {
            const float zz = fmaf(qn, W3[D::N2 * D::N3 + o], z[r]) + hp.b3[H][o];
            const float p = 1.0f / (1.0f + expf_canonical_h(-zz));
            const size_t idx = (size_t)ctu * kNOut + D::O3 + o;
            if (logits) logits[idx] = zz;
            if (raw) raw[idx] = p;
            __hip_atomic_store(&probs[idx], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (H == 0 && p > thr1 && __hip_atomic_load(flag32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __hip_atomic_store(flag32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (H == 1 && p > thr2 && __hip_atomic_load(flag16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __hip_atomic_store(flag16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


// The tf.cond gates (net_CNN.py:175,187) applied INSIDE the launch that computes the probabilities.  Every heads block
// publishes its probabilities and predicates with agent-scope stores, waits for them (s_waitcnt vmcnt(0)), and then adds 1 to
// the ARRIVAL COUNTER of every gate sub-batch its 64 CTUs touch; the block whose add completes a sub-batch (3 heads x the
// 64-CTU tiles overlapping it) reads that sub-batch's predicates and zero-fills it if a gate is closed (usually it is not).
// One counter per sub-batch, not one ticket per launch: 4782 blocks adding to ONE word serialise at the memory side (measured:
// the C3 heads stage 0.150 -> 0.224 ms with a single ticket).  Same scheme as k_lstm_heads; memory-model argument in DESIGN.md
// ("hand-offs inside a launch").  sync = [2 * nchunks predicates][nchunks arrival counters]..., zero on entry.
struct GateArrive {
    int n;        // sub-batches this block completed
    int ch[66];   // their indices (a 64-CTU tile touches <= 65 sub-batches: one-CTU frames)
};
// CTU range [c0, c1) of sub-batch ch of the pass: the inverse of gate_chunk
__device__ __forceinline__ void gate_chunk_range(const GateIndex& gi, int ch, int N, int& c0, int& c1) {
    const int cc = ch + gi.c0, f = cc / gi.cpf, k = cc - f * gi.cpf;
    const int u0 = f * gi.nctu + k * kSubBatch, u1 = f * gi.nctu + min((k + 1) * kSubBatch, gi.nctu);
    c0 = max(u0 - gi.r0, 0);
    c1 = min(u1 - gi.r0, N);
}
// first_ctu: first CTU of this block's 64-CTU tile.  All threads of the block must call it (barriers inside).
// SELF_CLEAN (the single-launch small pass): the completer hands the sub-batch's predicate and arrival words back as zeros.
// TILE: CTUs per heads block (64; 16 in the single-launch small pass).
template <bool SELF_CLEAN = false, int TILE = 64>
__device__ __forceinline__ void heads_gates_arrive(int* sync, int* arrived, const GateIndex& gi, int N, int first_ctu, float thr2,
                                                   float* probs, GateArrive* ga) {
    ETHCNN_HANDOFF_RELEASE();  // this block's probabilities and predicates have completed
    __syncthreads();
    if (threadIdx.x == 0) {
        const int last_ctu = min(first_ctu + TILE - 1, N - 1);
        const int ca = gate_chunk(gi, first_ctu), cb = gate_chunk(gi, last_ctu);
        int n = 0;
        for (int ch = ca; ch <= cb; ++ch) {
            int c0, c1;
            gate_chunk_range(gi, ch, N, c0, c1);
            const int expected = 3 * ((c1 - 1) / TILE - c0 / TILE + 1);  // 3 heads x tiles overlapping [c0, c1)
            if (__hip_atomic_fetch_add(arrived + ch, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == expected) ga->ch[n++] = ch;
        }
        ga->n = n;
    }
    __syncthreads();
    if (ga->n > 0) ETHCNN_HANDOFF_ACQUIRE();  // the completer reads the other blocks' predicates and may overwrite their probabilities
    for (int i = 0; i < ga->n; ++i) {
        const int ch = ga->ch[i];
        const bool open32 = __hip_atomic_load(sync + 2 * ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        // y16 is gated on the GATED y32: a closed L1 gate leaves zeros, and any(0 > thr2) decides
        const bool open16 = open32 ? (__hip_atomic_load(sync + 2 * ch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) : (0.0f > thr2);
        if (SELF_CLEAN) {
            __syncthreads();  // every thread has read the predicates
            if (threadIdx.x == 0) {
                __hip_atomic_store(sync + 2 * ch, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sync + 2 * ch + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(arrived + ch, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (open32 && open16) continue;
        int c0, c1;
        gate_chunk_range(gi, ch, N, c0, c1);
        for (int idx = c0 * kNOut + (int)threadIdx.x; idx < c1 * kNOut; idx += (int)blockDim.x) {
            const int j = idx % kNOut;
            if (j != 0 && (j < 5 ? !open32 : !open16))
                __hip_atomic_store(probs + idx, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace ethcnn
