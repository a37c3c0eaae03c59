// Microbenchmark (round 5): do VALU instructions of ONE wave hide under 16-bit MFMAs of ANOTHER wave on the same SIMD of gfx950?
// (Round 1 measured fp32 MFMAs: they share the VALU datapath, the costs add -- profiles/r01_ubench_gfx950_issue_costs.txt.)
// Blocks of 512 threads = 2 waves per SIMD, one block per CU.  Per iteration a wave issues M matrix instructions (4 independent chains)
// and / or V VALU (v_fma_f32, 8 independent chains):
//   "mfma | idle"   waves 0..3 MFMA, waves 4..7 leave at once           -> cost of the MFMA stream alone
//   "idle | valu"   waves 0..3 leave, waves 4..7 VALU                   -> cost of the VALU stream alone
//   "mfma | valu"   waves 0..3 MFMA, waves 4..7 VALU                    -> max (they overlap) or sum (they share the issue / datapath)?
//   "both | both"   every wave M MFMA then V VALU (the trunk's shape)   -> 2 (M + V) if nothing overlaps
// hipcc --offload-arch=gfx950 -O3 -o mfma16_valu mfma16_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

template <int KIND>
__device__ __forceinline__ f32x4 mfma(f32x4 c, float seed) {
    if constexpr (KIND == 0) return __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed * 0.5f, c, 0, 0, 0);
    else if constexpr (KIND == 1) {
        const h4 a = {(_Float16)seed, (_Float16)1, (_Float16)2, (_Float16)3};
        return __builtin_amdgcn_mfma_f32_16x16x16f16(a, a, c, 0, 0, 0);
    } else {
        const h8 a = {(_Float16)seed, (_Float16)1, (_Float16)2, (_Float16)3, (_Float16)4, (_Float16)5, (_Float16)6, (_Float16)7};
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c, 0, 0, 0);
    }
}

// role: bit 0 = MFMA, bit 1 = VALU; roleA for waves 0..3, roleB for waves 4..7
template <int KIND, int M, int V>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a0, int roleA, int roleB) {
    const int role = (threadIdx.x < 256) ? roleA : roleB;
    if (role == 0) return;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a0 + threadIdx.x * 0.001f + i;
    const float a = a0, b = a0 * 0.5f;
    for (int it = 0; it < iters; ++it) {
        if (role & 1) {
#pragma unroll
            for (int m = 0; m < M; ++m) acc[m & 3] = mfma<KIND>(acc[m & 3], a0);
        }
        if (role & 2) {
#pragma unroll
            for (int m = 0; m < V; ++m) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[m & 7]) : "v"(a), "v"(b));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND, int M, int V>
static double run(int roleA, int roleB) {
    static float* d = nullptr;
    if (!d) hipMalloc(&d, 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<KIND, M, V><<<256, 512>>>(d, 50, 1.0f, roleA, roleB);
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        k<KIND, M, V><<<256, 512>>>(d, iters, 1.0f, roleA, roleB);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3 / iters;  // us per iteration
}

template <int KIND, int M, int V>
static void table(const char* name) {
    const double m = run<KIND, M, V>(1, 0), v = run<KIND, M, V>(0, 2), mv = run<KIND, M, V>(1, 2), bb = run<KIND, M, V>(3, 3), b1 = run<KIND, M, V>(3, 0);
    printf("%-26s M %3d V %3d | mfma|idle %7.3f  idle|valu %7.3f  mfma|valu %7.3f (sum %7.3f max %7.3f)  both|idle %7.3f  both|both %7.3f  us/iter\n", name, M, V,
           m, v, mv, m + v, m > v ? m : v, b1, bb);
}

int main() {
    table<0, 16, 64>("v_mfma_f32_16x16x4_f32");
    table<1, 32, 64>("v_mfma_f32_16x16x16_f16");
    table<2, 32, 64>("v_mfma_f32_16x16x32_f16");
    table<2, 32, 128>("v_mfma_f32_16x16x32_f16");
    table<2, 16, 128>("v_mfma_f32_16x16x32_f16");
    table<1, 32, 128>("v_mfma_f32_16x16x16_f16");
    return 0;
}
