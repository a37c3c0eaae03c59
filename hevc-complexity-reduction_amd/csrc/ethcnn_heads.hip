// ethcnn_heads.hip -- the three QP-conditioned FC heads after FC1, fused in one kernel:
//   h2 = lrelu([h1, qp] W2 + b2)   (net_CNN.py:159,167,180)
//   y  = sigmoid([h2, qp] W3 + b3) (net_CNN.py:161,169,182)  + the gate predicates (:175,187)
// gfx950, v_mfma_f32_16x16x4_f32.
//
// One wave = 16 CTUs, "transposed": MFMA rows = output features, columns = CTUs.
//   FC2^T: A operand = W2 (16-k chunks by LDS-DMA, shared by the block's 4 waves, 2 stages),
//          B operand = this wave's h1 rows, one float4 per lane per chunk (element e feeds MFMA
//          step e -> k order 16c + 4g + e, the canonical FC order; head 16 fetches it straight into
//          registers a chunk ahead, heads 32 / 64 through the stage).  One block = one head of a
//          64-CTU tile (blockIdx.y = head, 16 first): short per-block latency, three times the blocks;
//          24 KB of LDS and 77 VGPRs: six blocks per CU cover each other's DMA latency.
//   The FC2^T accumulator of lane (ctu, g) holds h2[ctu][16t + 4g + r]: exactly the B operand
//   FC3^T needs for step (t, r) -- so FC2 -> FC3 chains in registers (same k order), no LDS,
//   no HBM round trip.  FC3^T's A operand (W3, 3525 floats in all) comes straight from L1/L2.
// Where the stage's time goes (phase stamps, device timeline, the rebuilt "wide" variant that was measured and dropped):
// profiles/r02_heads_timeline.txt, scripts/ubench/heads_probe.hip.
#include <hip/hip_runtime.h>

#include "ethcnn_kernels.h"

namespace ethcnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

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
        if (!D::H1REG) HP_DMA(a_off, H1 + (kc) * 16, 4u * ((st) * kHeadsStage + D::H1_AT + wvu * 256));  \
    }
#define HP_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

    f32x4 acc[D::NT];
#pragma unroll
    for (int j = 0; j < D::NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const __amdgpu_buffer_rsrc_t rH1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(H1), 0, -1, 0x00020000);
    f32x4 avr = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (D::H1REG) avr = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rH1, a_off, 0, 0));
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
            if (kc + 1 < D::NK) avn = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rH1, a_off, (kc + 1) * 64, 0));
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

__global__ __launch_bounds__(256) void k_heads(const float* __restrict__ H1, HeadsParams hp, float qn, int N, GateIndex gi,
                                               float thr1, float thr2, float* __restrict__ H2,
                                               float* __restrict__ logits, float* __restrict__ raw,
                                               float* __restrict__ probs, int* __restrict__ flags) {
    __shared__ __attribute__((aligned(16))) float smem[kHeadsStages * kHeadsStage];  // the ONLY LDS object (24 KB)
    HEADS_STAMP(0);
    const int lane = threadIdx.x & 63;
    const unsigned wvu = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15;
    const int tile_ = blockIdx.x, head_ = blockIdx.y;
    const int ctu_raw = (tile_ * 4 + (int)wvu) * 16 + col;
    const bool valid = ctu_raw < N;
    const int ctu = min(ctu_raw, N - 1);  // clamped rows are loaded, never stored
    float* h2row = H2 ? H2 + (size_t)ctu * kNFc2 : nullptr;
    // only heads 64 / 32 raise predicates (blockIdx.y = 2 / 1); head 16 (most of the blocks) skips the index arithmetic
    int* fl = flags;
    if (head_ != 0) fl += 2 * gate_chunk(gi, ctu);
    // blockIdx.y selects the head: the three heads of a 64-CTU tile are independent (each reads its own
    // column slice of h1), so they run as separate blocks -- head 16 (16 K chunks) is dispatched first,
    // the short heads 32 / 64 fill in behind it.  A third of the per-block latency, three times the blocks.
    if (head_ == 0)
        head_pass<2>(smem, H1, hp, qn, lane, wvu, valid, ctu, h2row, logits, raw, probs, fl, fl + 1, thr1, thr2);
    else if (head_ == 1)
        head_pass<1>(smem, H1, hp, qn, lane, wvu, valid, ctu, h2row, logits, raw, probs, fl, fl + 1, thr1, thr2);
    else
        head_pass<0>(smem, H1, hp, qn, lane, wvu, valid, ctu, h2row, logits, raw, probs, fl, fl + 1, thr1, thr2);
    HEADS_STAMP(4);
}

void launch_heads(const Workspace& ws, const DeviceWeights& w, int n, float qn, int nctu, long ctu0, float thr1,
                  float thr2, float* d_probs, hipStream_t s) {
    HeadsParams hp;
    for (int h = 0; h < 3; ++h) {
        hp.w2[h] = w.fc2_w[h];
        hp.b2[h] = w.fc2_b[h];
        hp.w3[h] = w.fc3_w[h];
        hp.b3[h] = w.fc3_b[h];
    }
    const GateIndex gi = make_gate_index(nctu, ctu0);
    const dim3 grid((n + 63) / 64, 3);
    hipLaunchKernelGGL(k_heads, grid, dim3(256), 0, s, ws.h1, hp, qn, n, gi, thr1, thr2, ws.h2, ws.logits, ws.raw, d_probs,
                       ws.flags);
}

}  // namespace ethcnn
