// ethcnn_heads_fast.hip -- plan 3 (opt-in, ethcnn_set_fc1_plan(ctx, 3)), round 5: the three QP-conditioned heads after FC1
//   h2 = lrelu([h1, qp] W2 + b2)   (net_CNN.py:159,167,180)
//   y  = sigmoid([h2, qp] W3 + b3) (net_CNN.py:161,169,182)  + the gate predicates (:175,187)
// on the 16-BIT matrix pipe, operands as fp16 x 2 splits of power-of-two scaled values like FC1's (ethcnn_fc1_fast.hip) and the
// trunk's (ethcnn_trunk_fast.hip), fp32 accumulation.  With FC1 and the trunk on that pipe the exact-fp32 heads (1008 + 21
// v_mfma_f32_16x16x4_f32 of 32 cycles per 16 CTUs, 0.16 ms per C3 step at the throttled clock) were 14 % of a plan-3 step.
//
// Same decomposition as k_heads (ethcnn_heads.hip): one block = one head of a 64-CTU tile (blockIdx.y: head 16 first), one wave = 16
// CTUs, "transposed" (MFMA rows = output features, columns = CTUs) so that the FC2 accumulators are FC3's B operand in registers.
//   FC2: v_mfma_f32_16x16x32_f16, K walked in chunks of 32: the B operand of lane (ctu, kg) is h1[ctu][O1 + 32 c + 8 kg .. + 7] --
//        two float4 loads straight into registers a chunk ahead, scaled by S1 (a power of two: exact) and split hi = fp16(v),
//        lo = fp16((v - hi) 2^11) (a SCALED residual, see split8r) -- three products per tile: hi W0 into the main accumulator, lo W0 and
//        hi W1 (W1 likewise a scaled residual) into a correction accumulator that is added back with 2^-11.  The A operands (W2 as
//        two fp16 pieces, packed per lane at weight load: pack_heads_f16) reach the block's four waves through a 2-stage LDS ring by
//        LDS-DMA (linear 1 KiB per (tile, piece): one conflict-free ds_read_b128 per lane).
//   FC3: the FC2 epilogue (x U2, + qp row, + bias, leaky-ReLU) leaves h2 of tiles (2 p, 2 p + 1) as the eight k values of step p; split
//        again (scale S2), three products against the W3 pieces (rows >= n3 are zero), fetched from L2 while the epilogue runs.
// 126 + 2..6 steps of 3 matrix instructions of 16 cycles per 16 CTUs instead of 1029 of 32.  Numerics: fp32-class like the other
// fast stages (different rounding points: not bit-identical to the oracle; bar = the north star's 1e-4, tests/test_gpu_fast_plan.py).
#include <hip/hip_runtime.h>

#include "ethcnn_heads_pass.h"
#include "ethcnn_kernels.h"

namespace ethcnn {

typedef _Float16 hh8 __attribute__((ext_vector_type(8)));
typedef float ff2 __attribute__((ext_vector_type(2)));
typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
#define MFMA32HF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// eight (already scaled) fp32 values -> fp16 pieces hi = fp16(v), lo = fp16((v - hi) * 2^11): v = hi + lo * 2^-11 to 2^-22 relative
// WHEREVER hi is a normal number.  The residual is scaled because the heads' activation scales come from worst-case bounds that are
// loose by ~2^15 (|h1| <= |b1| + bound(features) * sum |W1|): a typical scaled value sits near 2^-6 .. 2^0, its plain residual (2^-11
// of it) would fall into fp16's subnormal range and keep only a few bits (measured with the plain split: 4x the exact plan's error
// against float64).  The scaled pieces feed a second accumulator (see head_pass_f16).
constexpr float kResidualScale = 2048.0f, kResidualUnscale = 1.0f / 2048.0f;
__device__ __forceinline__ void split8r(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const hh2 h = __builtin_convertvector((ff2){v[2 * i], v[2 * i + 1]}, hh2);
        hi[i] = __builtin_bit_cast(unsigned, h);
        float r0, r1;  // v - hi, exact (the half read straight out of the packed register)
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi[i]), "v"(v[2 * i]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi[i]), "v"(v[2 * i + 1]));
        const hh2 l = __builtin_convertvector((ff2){r0 * kResidualScale, r1 * kResidualScale}, hh2);
        lo[i] = __builtin_bit_cast(unsigned, l);
    }
}

struct Heads16Params {
    const uint16_t* img;  // DeviceWeights::heads16_w
    Heads16Scalars sc;
};
constexpr int kHeads16Stage = 24 * 1024;  // bytes: the widest chunk (head 16: 12 tiles x 2 pieces x 1 KiB)
constexpr int kHeads16Stages = 2;

template <int H>
__device__ __forceinline__ void head_pass_f16(char* smem, const float* __restrict__ H1, const HeadsParams& hp, const Heads16Params& fp, float qn,
                                              int lane, unsigned wvu, bool valid, int ctu, float* __restrict__ h2row, float* __restrict__ logits,
                                              float* __restrict__ raw, float* __restrict__ probs, int* flag32, int* flag16, float thr1, float thr2) {
    using D = Hd<H>;
    constexpr int NK = D::N1 / 32;             // chunks of 32 k: 2 / 4 / 8
    constexpr int NT = D::NT;                  // output tiles: 3 / 6 / 12
    constexpr int INST = NT * 2;               // 1 KiB LDS-DMA instructions per chunk: 6 / 12 / 24
    constexpr int PER = (INST + 3) / 4;        // per wave (the tail duplicates the last piece)
    constexpr int NP = (NT + 1) / 2;           // FC3 steps
    const int g = lane >> 4;
    const float S1 = fp.sc.S1[H], U2 = fp.sc.U2[H], S2 = fp.sc.S2[H], U3 = fp.sc.U3[H];
    const uint16_t* W2i = fp.img + heads16_fc2_at(H);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)smem);
    const unsigned voff = (unsigned)lane * 16u;

#define HF_DMA(sbase, lds_byte_off)                                                                      \
    {                                                                                                    \
        unsigned keep_;                                                                                  \
        const unsigned dst_ = __builtin_amdgcn_readfirstlane(lds_base + (lds_byte_off));                 \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(voff), "s"(sbase), "s"(dst_) : "memory");                      \
    }
#define HF_ISSUE(kc, st)                                                                                 \
    {                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                                \
            const unsigned q_ = min(wvu + i * 4, (unsigned)(INST - 1));                                  \
            HF_DMA(W2i + ((size_t)(kc) * INST + q_) * 512, (unsigned)((st) * kHeads16Stage) + q_ * 1024u); \
        }                                                                                                \
    }
    const __amdgpu_buffer_rsrc_t rH1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(H1), 0, -1, 0x00020000);
    const unsigned a_off = 4u * (unsigned)(ctu * kNVec + D::O1 + 8 * g);
    // acc: hi x W0; cor: the two cross terms lo' x W0 + hi x W1', both carried at 2^11 (scaled residuals): result = acc + cor 2^-11
    f32x4 acc[NT], cor[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = cor[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 hA = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rH1, a_off, 0, 0));
    f32x4 hB = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rH1, a_off, 16, 0));
    __builtin_amdgcn_s_barrier();  // (a block runs one head: nothing of an earlier pass is in the stages; kept for the raw-barrier pairing below)
    HF_ISSUE(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int st = 0;
#pragma unroll 1
    for (int kc = 0; kc < NK; ++kc) {
        const int st2 = st ^ 1;
        if (kc + 1 < NK) { HF_ISSUE(kc + 1, st2); }
        float v[8] = {hA[0] * S1, hA[1] * S1, hA[2] * S1, hA[3] * S1, hB[0] * S1, hB[1] * S1, hB[2] * S1, hB[3] * S1};
        if (kc + 1 < NK) {
            hA = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rH1, a_off, (kc + 1) * 128, 0));
            hB = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rH1, a_off, (kc + 1) * 128 + 16, 0));
        }
        u32x4 bhi, blo;
        split8r(v, bhi, blo);
        const char* sp = smem + st * kHeads16Stage + lane * 16;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const hh8 w0 = *reinterpret_cast<const hh8*>(sp + (j * 2 + 0) * 1024);
            const hh8 w1 = *reinterpret_cast<const hh8*>(sp + (j * 2 + 1) * 1024);
            acc[j] = MFMA32HF(w0, __builtin_bit_cast(hh8, bhi), acc[j]);
            cor[j] = MFMA32HF(w0, __builtin_bit_cast(hh8, blo), cor[j]);
            cor[j] = MFMA32HF(w1, __builtin_bit_cast(hh8, bhi), cor[j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        st = st2;
    }
#undef HF_DMA
#undef HF_ISSUE
    // FC3's A operands (W3 pieces, 2 KiB per step) are requested now and arrive under the FC2 epilogue
    const __amdgpu_buffer_rsrc_t rW3i = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(fp.img + heads16_fc3_at(H)), 0, NP * 2048, 0x00020000);
    u32x4 a3[NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        a3[p][0] = __builtin_amdgcn_raw_buffer_load_b128(rW3i, voff, (p * 2 + 0) * 1024, 0);
        a3[p][1] = __builtin_amdgcn_raw_buffer_load_b128(rW3i, voff, (p * 2 + 1) * 1024, 0);
    }
    // FC2 epilogue in place: lane (ctu = col, g) holds h2[ctu][16 j + 4 g + r]
    const float* W2 = hp.w2[H];
    const float* W3 = hp.w3[H];
    const __amdgpu_buffer_rsrc_t rW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W2), 0, (D::N1 + 1) * D::N2 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hp.b2[H]), 0, D::N2 * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const f32x4 wq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW2, 16 * g, (D::N1 * D::N2 + 16 * j) * 4, 0));
        const f32x4 bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB2, 16 * g, 64 * j, 0));
        acc[j][0] = lrelu_h(fmaf(fmaf(cor[j][0], kResidualUnscale, acc[j][0]), U2, fmaf(qn, wq.x, bv.x)));
        acc[j][1] = lrelu_h(fmaf(fmaf(cor[j][1], kResidualUnscale, acc[j][1]), U2, fmaf(qn, wq.y, bv.y)));
        acc[j][2] = lrelu_h(fmaf(fmaf(cor[j][2], kResidualUnscale, acc[j][2]), U2, fmaf(qn, wq.z, bv.z)));
        acc[j][3] = lrelu_h(fmaf(fmaf(cor[j][3], kResidualUnscale, acc[j][3]), U2, fmaf(qn, wq.w, bv.w)));
        if (valid && h2row) *reinterpret_cast<f32x4*>(h2row + D::O2 + 16 * j + 4 * g) = acc[j];
    }
    // FC3^T: rows = outputs (n3 of 16 used), columns = CTUs; step p consumes the quads of tiles 2 p, 2 p + 1
    f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f}, zc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (2 * p + (i >> 2) < NT) ? acc[(2 * p + (i >> 2) < NT) ? 2 * p + (i >> 2) : 0][i & 3] * S2 : 0.0f;
        u32x4 bhi, blo;
        split8r(v, bhi, blo);
        z = MFMA32HF(__builtin_bit_cast(hh8, a3[p][0]), __builtin_bit_cast(hh8, bhi), z);
        zc = MFMA32HF(__builtin_bit_cast(hh8, a3[p][0]), __builtin_bit_cast(hh8, blo), zc);
        zc = MFMA32HF(__builtin_bit_cast(hh8, a3[p][1]), __builtin_bit_cast(hh8, bhi), zc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = fmaf(zc[r], kResidualUnscale, z[r]);
    // lane (ctu = col, g) holds outputs 4 g + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 4 * g + r;
        if (o < D::N3 && valid) {
            const float zz = fmaf(z[r], U3, fmaf(qn, W3[D::N2 * D::N3 + o], hp.b3[H][o]));
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

__global__ __launch_bounds__(256, 3) void k_heads_f16(const float* __restrict__ H1, HeadsParams hp, Heads16Params fp, float qn, int N, GateIndex gi,
                                                   float thr1, float thr2, float* __restrict__ H2, float* __restrict__ logits,
                                                   float* __restrict__ raw, float* __restrict__ probs, int* __restrict__ flags) {
    __shared__ __attribute__((aligned(16))) char smem[kHeads16Stages * kHeads16Stage];  // 48 KB
    const int lane = threadIdx.x & 63;
    const unsigned wvu = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15;
    const int tile_ = blockIdx.x, head_ = blockIdx.y;
    const int ctu_raw = (tile_ * 4 + (int)wvu) * 16 + col;
    const bool valid = ctu_raw < N;
    const int ctu = min(ctu_raw, N - 1);  // clamped rows are loaded, never stored
    float* h2row = H2 ? H2 + (size_t)ctu * kNFc2 : nullptr;
    int* fl = flags;
    if (head_ != 0) fl += 2 * gate_chunk(gi, ctu);
    if (head_ == 0)
        head_pass_f16<2>(smem, H1, hp, fp, qn, lane, wvu, valid, ctu, h2row, logits, raw, probs, fl, fl + 1, thr1, thr2);
    else if (head_ == 1)
        head_pass_f16<1>(smem, H1, hp, fp, qn, lane, wvu, valid, ctu, h2row, logits, raw, probs, fl, fl + 1, thr1, thr2);
    else
        head_pass_f16<0>(smem, H1, hp, fp, qn, lane, wvu, valid, ctu, h2row, logits, raw, probs, fl, fl + 1, thr1, thr2);
}

void launch_heads_f16(const Workspace& ws, const DeviceWeights& w, int n, float qn, int nctu, long ctu0, float thr1, float thr2,
                      float* d_probs, hipStream_t s) {
    HeadsParams hp;
    for (int h = 0; h < 3; ++h) {
        hp.w2[h] = w.fc2_w[h];
        hp.w2lane[h] = w.fc2_lane[h];
        hp.b2[h] = w.fc2_b[h];
        hp.w3[h] = w.fc3_w[h];
        hp.b3[h] = w.fc3_b[h];
    }
    Heads16Params fp;
    fp.img = w.heads16_w;
    fp.sc = w.heads16_s;
    const GateIndex gi = make_gate_index(nctu, ctu0);
    const dim3 grid((n + 63) / 64, 3);
    hipLaunchKernelGGL(k_heads_f16, grid, dim3(256), 0, s, ws.h1, hp, fp, qn, n, gi, thr1, thr2, ws.h2, ws.logits, ws.raw, d_probs, ws.flags);
}

}  // namespace ethcnn
