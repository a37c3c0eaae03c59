// bf16x3_probe.hip -- what the bf16 matrix pipe gives an fp32 product that is carried as an EXACT three-way bf16 split
// (a = a0 + a1 + a2, b likewise; the six cross products with i + j <= 2 on v_mfma_f32_{32x32x16,16x16x32}_bf16, fp32
// accumulate): (1) the rate of both MFMA shapes with FC1-like register tiles, (2) the numerics of that sum against a float64
// reference, beside the exact-fp32 fmaf chain (what v_mfma_f32_16x16x4_f32 computes) on the same data, for several orders
// of the six products, (3) that the device-side split (v_cvt_pk_bf16_f32, round to nearest even) is exact.
// build: hipcc --offload-arch=gfx950 -O2 -mllvm -amdgpu-mfma-vgpr-form bf16x3_probe.hip -o bf16x3_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MF32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------- (1) rate ---------------------------------------------
template <int SHAPE>  // 0: 32x32x16 with 7 accumulator tiles (112 regs); 1: 16x16x32 with 28 tiles (112 regs)
__global__ __launch_bounds__(512) void k_rate(int iters, float* sink) {
    bf16x8 a[3], b[3];
    for (int p = 0; p < 3; ++p) {
        u32x4 va = {threadIdx.x * 7u + p, 0x3f803f80u, 0x3f003f00u + p, 0x3e803e80u};
        u32x4 vb = {threadIdx.x * 3u + p, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u + p};
        a[p] = __builtin_bit_cast(bf16x8, va);
        b[p] = __builtin_bit_cast(bf16x8, vb);
    }
    float s = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[7];
        for (int j = 0; j < 7; ++j)
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                acc[j] = MF32(a[0], b[0], acc[j]);
                acc[j] = MF32(a[0], b[1], acc[j]);
                acc[j] = MF32(a[1], b[0], acc[j]);
                acc[j] = MF32(a[1], b[1], acc[j]);
                acc[j] = MF32(a[0], b[2], acc[j]);
                acc[j] = MF32(a[2], b[0], acc[j]);
            }
        }
        for (int j = 0; j < 7; ++j) s += acc[j][j];
    } else {
        f32x4 acc[28];
        for (int j = 0; j < 28; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 28; ++j) {
                acc[j] = MF16(a[0], b[0], acc[j]);
                acc[j] = MF16(a[0], b[1], acc[j]);
                acc[j] = MF16(a[1], b[0], acc[j]);
                acc[j] = MF16(a[1], b[1], acc[j]);
                acc[j] = MF16(a[0], b[2], acc[j]);
                acc[j] = MF16(a[2], b[0], acc[j]);
            }
        }
        for (int j = 0; j < 28; ++j) s += acc[j][j & 3];
    }
    if (s == 123.456f) *sink = s;
}

// ---------------------------------------------------------------- (2) numerics -----------------------------------------
// exact three-way split of an fp32 value into bf16 pieces, round to nearest even at each step (what the trunk epilogue of the
// fast plan does): returns the three pieces as bf16 bit patterns
__device__ __forceinline__ void split3(float x, unsigned short& p0, unsigned short& p1, unsigned short& p2) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto cvt = [](float v) -> unsigned short {
        f32x2 in = {v, 0.f};
        bf16x2 o = __builtin_convertvector(in, bf16x2);
        return (unsigned short)(__builtin_bit_cast(unsigned, o) & 0xffffu);
    };
    p0 = cvt(x);
    const float r1 = x - __builtin_bit_cast(float, (unsigned)p0 << 16);
    p1 = cvt(r1);
    const float r2 = r1 - __builtin_bit_cast(float, (unsigned)p1 << 16);
    p2 = cvt(r2);
}

// A: [32][K] fp32 row-major, B: [K][32] fp32.  One wave.  order: permutation id of the six products.
// out: [32][32].  SHAPE 0: one 32x32 tile, K step 16; SHAPE 1: 2x2 16x16 tiles, K step 32.
// split_bad: counts lanes whose pieces do not add back to the input exactly.
template <int SHAPE>
__global__ __launch_bounds__(64) void k_num(const float* A, const float* B, int K, int order, float* out, int* split_bad) {
    const int lane = threadIdx.x;
    static const int ordtab[4][6][2] = {
        {{0, 0}, {0, 1}, {1, 0}, {1, 1}, {0, 2}, {2, 0}},   // big first
        {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}},   // small first
        {{0, 0}, {1, 0}, {0, 1}, {2, 0}, {1, 1}, {0, 2}},
        {{0, 0}, {0, 1}, {1, 0}, {0, 0}, {0, 0}, {0, 0}}};  // (3: only three products: bf16x2, for scale; uses first 3)
    const int nprod = (order == 3) ? 3 : 6;
    int bad = 0;
    if (SHAPE == 0) {
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int rc = lane & 31, kh = lane >> 5;
        for (int k0 = 0; k0 < K; k0 += 16) {
            unsigned short pa[3][8], pb[3][8];
            for (int i = 0; i < 8; ++i) {
                const float av = A[(size_t)rc * K + k0 + 8 * kh + i], bv = B[(size_t)(k0 + 8 * kh + i) * 32 + rc];
                split3(av, pa[0][i], pa[1][i], pa[2][i]);
                split3(bv, pb[0][i], pb[1][i], pb[2][i]);
                auto f = [](unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); };
                if ((f(pa[0][i]) + f(pa[1][i])) + f(pa[2][i]) != av) ++bad;
                if ((f(pb[0][i]) + f(pb[1][i])) + f(pb[2][i]) != bv) ++bad;
            }
            bf16x8 fa[3], fb[3];
            for (int p = 0; p < 3; ++p) {
                u32x4 va, vb;
                for (int i = 0; i < 4; ++i) {
                    va[i] = (unsigned)pa[p][2 * i] | ((unsigned)pa[p][2 * i + 1] << 16);
                    vb[i] = (unsigned)pb[p][2 * i] | ((unsigned)pb[p][2 * i + 1] << 16);
                }
                fa[p] = __builtin_bit_cast(bf16x8, va);
                fb[p] = __builtin_bit_cast(bf16x8, vb);
            }
            for (int q = 0; q < nprod; ++q) {
                const int i = ordtab[order][q][0], j = ordtab[order][q][1];
                // (runtime piece selection through a switch keeps the operands in registers)
                bf16x8 xa = i == 0 ? fa[0] : (i == 1 ? fa[1] : fa[2]);
                bf16x8 xb = j == 0 ? fb[0] : (j == 1 ? fb[1] : fb[2]);
                acc = MF32(xa, xb, acc);
            }
        }
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
            out[row * 32 + col] = acc[r];
        }
    } else {
        f32x4 acc[2][2];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int rc = lane & 15, kb = lane >> 4;
        for (int k0 = 0; k0 < K; k0 += 32) {
            bf16x8 fa[2][3], fb[2][3];
            for (int t = 0; t < 2; ++t) {
                unsigned short pa[3][8], pb[3][8];
                for (int i = 0; i < 8; ++i) {
                    split3(A[(size_t)(16 * t + rc) * K + k0 + 8 * kb + i], pa[0][i], pa[1][i], pa[2][i]);
                    split3(B[(size_t)(k0 + 8 * kb + i) * 32 + 16 * t + rc], pb[0][i], pb[1][i], pb[2][i]);
                }
                for (int p = 0; p < 3; ++p) {
                    u32x4 va, vb;
                    for (int i = 0; i < 4; ++i) {
                        va[i] = (unsigned)pa[p][2 * i] | ((unsigned)pa[p][2 * i + 1] << 16);
                        vb[i] = (unsigned)pb[p][2 * i] | ((unsigned)pb[p][2 * i + 1] << 16);
                    }
                    fa[t][p] = __builtin_bit_cast(bf16x8, va);
                    fb[t][p] = __builtin_bit_cast(bf16x8, vb);
                }
            }
            for (int q = 0; q < nprod; ++q) {
                const int i = ordtab[order][q][0], j = ordtab[order][q][1];
                for (int ti = 0; ti < 2; ++ti)
                    for (int tj = 0; tj < 2; ++tj) {
                        bf16x8 xa = i == 0 ? fa[ti][0] : (i == 1 ? fa[ti][1] : fa[ti][2]);
                        bf16x8 xb = j == 0 ? fb[tj][0] : (j == 1 ? fb[tj][1] : fb[tj][2]);
                        acc[ti][tj] = MF16(xa, xb, acc[ti][tj]);
                    }
            }
        }
        for (int ti = 0; ti < 2; ++ti)
            for (int tj = 0; tj < 2; ++tj)
                for (int r = 0; r < 4; ++r) out[(16 * ti + 4 * (lane >> 4) + r) * 32 + 16 * tj + (lane & 15)] = acc[ti][tj][r];
    }
    if (bad) atomicAdd(split_bad, bad);
}

// fp16 x 2: a = (a0 + a1) 2^-sa with a0 = fp16(a 2^sa), a1 = fp16(a 2^sa - a0) (an 11-bit significand each: 2^-24 relative when a1 is a
// normal number); products a0 b0, a0 b1, a1 b0 (nprod 3) or also a1 b1 (nprod 4) on v_mfma_f32_32x32x16_f16; result scaled back by 2^-(sa + sb)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(64) void k_num_f16(const float* A, const float* B, int K, int nprod, float sa, float sb, float* out) {
    const int lane = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int rc = lane & 31, kh = lane >> 5;
    for (int k0 = 0; k0 < K; k0 += 16) {
        f16x8 fa[2], fb[2];
        for (int i = 0; i < 8; ++i) {
            const float av = A[(size_t)rc * K + k0 + 8 * kh + i] * sa, bv = B[(size_t)(k0 + 8 * kh + i) * 32 + rc] * sb;
            const _Float16 a0 = (_Float16)av, b0 = (_Float16)bv;
            fa[0][i] = a0; fa[1][i] = (_Float16)(av - (float)a0);
            fb[0][i] = b0; fb[1][i] = (_Float16)(bv - (float)b0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0], fb[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1], fb[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0], fb[1], acc, 0, 0, 0);
        if (nprod == 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1], fb[1], acc, 0, 0, 0);
    }
    const float inv = 1.0f / (sa * sb);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        out[row * 32 + col] = acc[r] * inv;
    }
}

static uint64_t sm64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double urand(uint64_t& s) { return (double)(sm64(s) >> 11) * (1.0 / 9007199254740992.0); }

int main() {
    float* sink;
    hipMalloc((void**)&sink, 4);
    // ---- rate
    for (int shape = 0; shape < 2; ++shape) {
        const int iters = 2000, blocks = 256;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            for (int l = 0; l < 5; ++l) {
                if (shape == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(512), 0, 0, iters, sink);
                else hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(512), 0, 0, iters, sink);
            }
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double mfmas = 5.0 * blocks * 8 * iters * (shape == 0 ? 42 : 168);
            const double flop = mfmas * (shape == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32);
            printf("rate %s, 8 waves/CU, 6-product register tile: %.3f ms, %.0f TF/s (bf16 flop), cycles/MFMA/SIMD at 2.4 GHz: %.1f\n",
                   shape == 0 ? "32x32x16" : "16x16x32", ms, flop / ms * 1e-9, ms * 1e-3 * 2.4e9 / (5.0 * iters * (shape == 0 ? 42 : 168) * 2));
        }
    }
    // ---- numerics
    const int K = 2688;
    for (int dist = 0; dist < 3; ++dist) {
        std::vector<float> A(32 * K), B(K * 32);
        uint64_t s = 1234 + dist;
        for (auto& v : A) {
            // dist 0: leaky-ReLU-like features (mostly positive, a fifth small negatives); 1: uniform +-1; 2: wide log-uniform magnitudes
            double u = urand(s) * 2 - 1;
            if (dist == 0) u = u > 0 ? u : 0.2 * u;
            if (dist == 2) u = (u > 0 ? 1 : -1) * std::exp2(-12.0 * urand(s));
            v = (float)u;
        }
        const double wscale = std::sqrt(3.0 / K);
        for (auto& v : B) v = (float)((urand(s) * 2 - 1) * wscale * (dist == 2 ? std::exp2(-8.0 * urand(s)) : 1.0));
        std::vector<double> ref(32 * 32), mag(32 * 32);
        std::vector<float> chain(32 * 32);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double r = 0, m = 0;
                float c = 0.f;
                for (int k = 0; k < K; ++k) {
                    r += (double)A[i * K + k] * (double)B[k * 32 + j];
                    m += std::fabs((double)A[i * K + k] * (double)B[k * 32 + j]);
                    c = std::fmaf(A[i * K + k], B[k * 32 + j], c);
                }
                ref[i * 32 + j] = r;
                mag[i * 32 + j] = m;
                chain[i * 32 + j] = c;
            }
        auto report = [&](const char* name, const float* got) {
            double emax = 0, erms = 0, ebias = 0, rel = 0;
            for (int i = 0; i < 1024; ++i) {
                const double e = (double)got[i] - ref[i];
                emax = std::max(emax, std::fabs(e));
                erms += e * e;
                ebias += e;
                rel = std::max(rel, std::fabs(e) / mag[i]);
            }
            printf("  dist %d  %-34s max|err| %.3e  rms %.3e  mean(signed) %+.3e  max err/sum|ab| %.3e\n", dist, name, emax, std::sqrt(erms / 1024),
                   ebias / 1024, rel);
        };
        report("fp32 fmaf chain (exact path)", chain.data());
        float *dA, *dB, *dO;
        int* dbad;
        hipMalloc((void**)&dA, A.size() * 4);
        hipMalloc((void**)&dB, B.size() * 4);
        hipMalloc((void**)&dO, 1024 * 4);
        hipMalloc((void**)&dbad, 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> got(1024);
        for (int shape = 0; shape < 2; ++shape)
            for (int order = 0; order < 4; ++order) {
                hipMemset(dbad, 0, 4);
                if (shape == 0) hipLaunchKernelGGL(k_num<0>, dim3(1), dim3(64), 0, 0, dA, dB, K, order, dO, dbad);
                else hipLaunchKernelGGL(k_num<1>, dim3(1), dim3(64), 0, 0, dA, dB, K, order, dO, dbad);
                hipDeviceSynchronize();
                hipMemcpy(got.data(), dO, 1024 * 4, hipMemcpyDeviceToHost);
                int bad = 0;
                hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
                char name[96];
                snprintf(name, sizeof name, "bf16x%d %s order %d (split bad %d)", order == 3 ? 2 : 3, shape == 0 ? "32x32x16" : "16x16x32", order, bad);
                report(name, got.data());
            }
        // fp16 x 2 at several operand scales: amax * sa = 2^14 is "just below overflow"; each factor 2^-4 pushes more low pieces into fp16's
        // subnormal range (absolute precision 2^-25 there)
        float amax = 0.f, bmax = 0.f;
        for (float v : A) amax = std::max(amax, std::fabs(v));
        for (float v : B) bmax = std::max(bmax, std::fabs(v));
        for (int down = 0; down <= 12; down += 4)
            for (int nprod = 3; nprod <= 4; ++nprod) {
                const float sa = std::exp2f(14 - down - std::ceil(std::log2(amax))), sb = std::exp2f(14 - down - std::ceil(std::log2(bmax)));
                hipLaunchKernelGGL(k_num_f16, dim3(1), dim3(64), 0, 0, dA, dB, K, nprod, sa, sb, dO);
                hipDeviceSynchronize();
                hipMemcpy(got.data(), dO, 1024 * 4, hipMemcpyDeviceToHost);
                char name[96];
                snprintf(name, sizeof name, "fp16x2 %d products, max scaled to 2^%d", nprod, 14 - down);
                report(name, got.data());
            }
        hipFree(dA);
        hipFree(dB);
        hipFree(dO);
        hipFree(dbad);
    }
    return 0;
}
