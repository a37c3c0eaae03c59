// ethcnn_heads.hip -- the three QP-conditioned FC heads after FC1, fused in one kernel:
//   h2 = lrelu([h1, qp] W2 + b2)   (net_CNN.py:159,167,180)
//   y  = sigmoid([h2, qp] W3 + b3) (net_CNN.py:161,169,182)  + the gate predicates (:175,187)
// gfx950, v_mfma_f32_16x16x4_f32.
//
// One wave = 16 CTUs, "transposed": MFMA rows = output features, columns = CTUs.
//   FC2^T: A operand = W2 (one 16-k chunk of every still-active head staged per iteration by
//          LDS-DMA, shared by the block's 4 waves; ONE barrier per iteration for all heads:
//          head 16 has 16 chunks, head 32 the first 8, head 64 the first 4), B operand = this
//          wave's h1 rows, one float4 per lane per chunk and head (element e feeds MFMA step e ->
//          k order 16c + 4g + e, the canonical FC order).
//   The FC2^T accumulator of lane (ctu, g) holds h2[ctu][16t + 4g + r]: exactly the B operand
//   FC3^T needs for step (t, r) -- so FC2 -> FC3 chains in registers (same k order), no LDS,
//   no HBM round trip.  FC3^T's A operand (W3, 3525 floats in all) comes straight from L1/L2.
#include <hip/hip_runtime.h>

#include "ethcnn_kernels.h"

namespace ethcnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
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

struct HeadsParams {
    const float* w2[3];
    const float* b2[3];
    const float* w3[3];
    const float* b3[3];
};

__device__ __forceinline__ long gchunk(long gn, int nctu, int cpf) {
    const long f = gn / nctu;
    return f * cpf + (gn - f * nctu) / kSubBatch;
}

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
    static constexpr int LDS_OFF = (H == 2) ? 0 : (H == 1 ? 16 * 192 : 16 * (192 + 96));  // inside a stage
    static constexpr bool COLSWZ = (N2 % 32 == 0);
};
constexpr int kHeadsStage = 16 * (192 + 96 + 48);  // floats per LDS stage (21.5 KB)

// per-head, per-lane state kept across the merged K loop
template <int H>
struct HeadState {
    f32x4 acc[Hd<H>::NT];
    float4 a_cur, a_nxt;
    const float* a_src;   // this lane's h1 float4 stream
    const float* b_src;   // LDS-DMA source of this lane (instructions wv, wv+4, ...)
    int bcol[Hd<H>::NT], brow[4];
};

template <int H>
__device__ __forceinline__ void head_init(HeadState<H>& st, const float* h1row, const float* W2, int lane, int wv) {
    using D = Hd<H>;
    const int col = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < D::NT; ++j) st.acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    st.a_src = h1row + D::O1 + 4 * g;
    // same permuted LDS image as ethcnn_dense.hip: position (p, c) holds W2[p'][c'] with rows /
    // column groups of odd (k>>2) swapped; here the permutation is applied to the DMA source
    {
        const int e = wv * 64 + lane;  // float4 index of instruction wv (others: + 4*64 per step)
        (void)e;
    }
    st.b_src = W2;
#pragma unroll
    for (int j = 0; j < D::NT; ++j) st.bcol[j] = (j * 16 + col) ^ (D::COLSWZ ? ((g & 1) << 4) : 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) st.brow[e] = (D::COLSWZ ? e : (e ^ (g & 1))) * D::N2;
}

// issue this wave's share of the LDS-DMA of chunk kc of head H into stage `buf`
template <int H>
__device__ __forceinline__ void head_issue(const HeadState<H>& st, float* smem, int kc, int buf, int lane, int wv) {
    using D = Hd<H>;
#pragma unroll
    for (int i = 0; i < (D::B_INST + 3) / 4; ++i) {
        const int q = wv + i * 4;
        if ((i + 1) * 4 <= D::B_INST || q < D::B_INST) {
            const int e = q * 64 + lane;
            int row = (e / (D::N2 / 4)) % 16;
            int c4 = e % (D::N2 / 4);
            if (D::COLSWZ) c4 ^= ((row >> 2) & 1) << 2;
            else row ^= (row >> 2) & 1;
            __builtin_amdgcn_global_load_lds((glb_void*)(st.b_src + (size_t)(kc * 16 + row) * D::N2 + c4 * 4),
                                             (lds_void*)(smem + buf * kHeadsStage + D::LDS_OFF + q * 256), 16, 0, 0);
        }
    }
}

template <int H>
__device__ __forceinline__ void head_mfma(HeadState<H>& st, const float* smem, int buf, int g) {
    using D = Hd<H>;
    const float* bs = smem + buf * kHeadsStage + D::LDS_OFF + 4 * g * D::N2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float hv = (e == 0) ? st.a_cur.x : (e == 1) ? st.a_cur.y : (e == 2) ? st.a_cur.z : st.a_cur.w;
#pragma unroll
        for (int j = 0; j < D::NT; ++j) st.acc[j] = MFMA16(bs[st.brow[e] + st.bcol[j]], hv, st.acc[j]);  // rows = W2 columns
    }
}

// FC2 epilogue (qp column, bias, leaky) in place, optional h2 store, FC3^T, sigmoid, outputs
template <int H>
__device__ __forceinline__ void head_finish(HeadState<H>& st, const HeadsParams& hp, float qn, int lane, bool valid,
                                            int ctu, float* __restrict__ h2row, float* __restrict__ logits,
                                            float* __restrict__ raw, float* __restrict__ probs, int* flag32,
                                            int* flag16, float thr1, float thr2) {
    using D = Hd<H>;
    const int col = lane & 15, g = lane >> 4;
    const float* W2 = hp.w2[H];
    const float* W3 = hp.w3[H];
    // FC3^T A operand: W3[k = 16 j + 4 g + r][out = col]; all loads issued before the first use
    float w3r[D::NT][4];
#pragma unroll
    for (int j = 0; j < D::NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) w3r[j][r] = (col < D::N3) ? W3[(16 * j + 4 * g + r) * D::N3 + col] : 0.0f;
    // lane (ctu = col, g) holds h2[ctu][16 j + 4 g + r]
#pragma unroll
    for (int j = 0; j < D::NT; ++j) {
        const int n = 16 * j + 4 * g;
        const float4 wq = *reinterpret_cast<const float4*>(W2 + (size_t)D::N1 * D::N2 + n);
        const float4 bv = *reinterpret_cast<const float4*>(hp.b2[H] + n);
        st.acc[j][0] = lrelu_h(fmaf(qn, wq.x, st.acc[j][0]) + bv.x);
        st.acc[j][1] = lrelu_h(fmaf(qn, wq.y, st.acc[j][1]) + bv.y);
        st.acc[j][2] = lrelu_h(fmaf(qn, wq.z, st.acc[j][2]) + bv.z);
        st.acc[j][3] = lrelu_h(fmaf(qn, wq.w, st.acc[j][3]) + bv.w);
        if (valid && h2row) *reinterpret_cast<f32x4*>(h2row + D::O2 + n) = st.acc[j];
    }
    // FC3^T: rows = outputs (N3 of 16 used), columns = CTUs; step (j, r) consumes k = 16 j + 4 g + r
    f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < D::NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) z = MFMA16(w3r[j][r], st.acc[j][r], z);
    // lane (ctu = col, g) holds outputs 4 g + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 4 * g + r;
        if (o < D::N3 && valid) {
            const float zz = fmaf(qn, W3[D::N2 * D::N3 + o], z[r]) + hp.b3[H][o];
            const float p = 1.0f / (1.0f + expf_canonical_h(-zz));
            const size_t idx = (size_t)ctu * kNOut + D::O3 + o;
            logits[idx] = zz;
            raw[idx] = p;
            probs[idx] = p;
            if (H == 0 && p > thr1 && __hip_atomic_load(flag32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __hip_atomic_store(flag32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // any(y64 > THR_L1_LOWER)
            if (H == 1 && p > thr2 && __hip_atomic_load(flag16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __hip_atomic_store(flag16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // any(y32_tmp > THR_L2_LOWER)
        }
    }
}

__global__ __launch_bounds__(256) void k_heads(const float* __restrict__ H1, HeadsParams hp, float qn, int N, int nctu,
                                               int cpf, long ctu0, float thr1, float thr2, float* __restrict__ H2,
                                               float* __restrict__ logits, float* __restrict__ raw,
                                               float* __restrict__ probs, int* __restrict__ flags) {
    __shared__ __attribute__((aligned(16))) float smem[2 * kHeadsStage];  // the ONLY LDS object
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 15, g = lane >> 4;
    const int ctu_raw = (blockIdx.x * 4 + wv) * 16 + col;
    const bool valid = ctu_raw < N;
    const int ctu = min(ctu_raw, N - 1);  // clamped rows are loaded, never stored
    const float* h1row = H1 + (size_t)ctu * kNVec;
    int* fl = flags + 2 * (gchunk(ctu0 + ctu, nctu, cpf) - gchunk(ctu0, nctu, cpf));

    HeadState<0> s0;
    HeadState<1> s1;
    HeadState<2> s2;
    head_init<0>(s0, h1row, hp.w2[0], lane, wv);
    head_init<1>(s1, h1row, hp.w2[1], lane, wv);
    head_init<2>(s2, h1row, hp.w2[2], lane, wv);

    head_issue<2>(s2, smem, 0, 0, lane, wv);
    head_issue<1>(s1, smem, 0, 0, lane, wv);
    head_issue<0>(s0, smem, 0, 0, lane, wv);
    s2.a_cur = *reinterpret_cast<const float4*>(s2.a_src);
    s1.a_cur = *reinterpret_cast<const float4*>(s1.a_src);
    s0.a_cur = *reinterpret_cast<const float4*>(s0.a_src);
    s2.a_nxt = s2.a_cur; s1.a_nxt = s1.a_cur; s0.a_nxt = s0.a_cur;
    __syncthreads();  // hipcc drains vmcnt here (LDS-DMA pending): chunk 0 has landed
#pragma unroll 1
    for (int kc = 0; kc < 16; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < 16) {
            head_issue<2>(s2, smem, kc + 1, buf ^ 1, lane, wv);
            s2.a_nxt = *reinterpret_cast<const float4*>(s2.a_src + (kc + 1) * 16);
            if (kc + 1 < 8) {
                head_issue<1>(s1, smem, kc + 1, buf ^ 1, lane, wv);
                s1.a_nxt = *reinterpret_cast<const float4*>(s1.a_src + (kc + 1) * 16);
            }
            if (kc + 1 < 4) {
                head_issue<0>(s0, smem, kc + 1, buf ^ 1, lane, wv);
                s0.a_nxt = *reinterpret_cast<const float4*>(s0.a_src + (kc + 1) * 16);
            }
        }
        head_mfma<2>(s2, smem, buf, g);
        if (kc < 8) head_mfma<1>(s1, smem, buf, g);
        if (kc < 4) head_mfma<0>(s0, smem, buf, g);
        s2.a_cur = s2.a_nxt;
        s1.a_cur = s1.a_nxt;
        s0.a_cur = s0.a_nxt;
        __syncthreads();
    }
    float* h2row = H2 ? H2 + (size_t)ctu * kNFc2 : nullptr;
    head_finish<0>(s0, hp, qn, lane, valid, ctu, h2row, logits, raw, probs, fl, fl + 1, thr1, thr2);
    head_finish<1>(s1, hp, qn, lane, valid, ctu, h2row, logits, raw, probs, fl, fl + 1, thr1, thr2);
    head_finish<2>(s2, hp, qn, lane, valid, ctu, h2row, logits, raw, probs, fl, fl + 1, thr1, thr2);
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
    hipLaunchKernelGGL(k_heads, dim3((n + 63) / 64), dim3(256), 0, s, ws.h1, hp, qn, n, nctu, chunks_per_frame(nctu),
                       ctu0, thr1, thr2, ws.h2, ws.logits, ws.raw, d_probs, ws.flags);
}

}  // namespace ethcnn
