// Microbenchmark: does VALU work hide under fp32 MFMAs on gfx950, (a) interleaved inside one wave,
// (b) in separate phases of one wave, with 1..4 waves per SIMD?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// MODE 0: MFMA only (NCH independent chains). 1: + VPM VALU per MFMA interleaved (asm block per MFMA).
// 2: phased: 16 MFMAs then 16*VPM VALU.  3: VALU only.
template <int MODE, int VPM, int NCH>
__global__ void k(float* out, int iters, float a0) {
    f32x4 acc[NCH];
    for (int i = 0; i < NCH; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a0 + threadIdx.x * 0.001f + i;
    const float a = a0, b = a0 * 0.5f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
#pragma unroll
            for (int m = 0; m < 16; ++m)
                if (MODE != 3) acc[m % NCH] = MFMA16(a, b, acc[m % NCH]);
#pragma unroll
            for (int m = 0; m < 16 * VPM; ++m)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[m % 8]) : "v"(a), "v"(b));
        } else {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (MODE != 3) acc[m % NCH] = MFMA16(a, b, acc[m % NCH]);
                if (MODE == 1 || MODE == 3) {
#pragma unroll
                    for (int q = 0; q < VPM; ++q)
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(m * VPM + q) % 8]) : "v"(a), "v"(b));
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < NCH; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE, int VPM, int NCH>
static void run(const char* name, int waves_per_simd) {
    float* d;
    hipMalloc(&d, 4);
    const int iters = 20000;
    const int blocks = 256, threads = 64 * 4 * waves_per_simd;  // one block per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE, VPM, NCH><<<blocks, threads>>>(d, 100, 1.0f);
    hipEventRecord(e0);
    k<MODE, VPM, NCH><<<blocks, threads>>>(d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = (MODE == 3) ? 0 : 16.0 * iters * waves_per_simd;  // MFMAs per SIMD
    const double cyc_per_mfma = mf ? ms * 1e-3 * 2.4e9 / mf : 0;
    const double tf = (MODE == 3) ? 0 : 2048.0 * 16 * iters * (double)blocks * 4 * waves_per_simd / (ms * 1e-3) / 1e12;
    printf("%-34s waves/SIMD %d  %8.3f ms  %6.1f TF  %5.1f cyc/MFMA(@2.4GHz)\n", name, waves_per_simd, ms, tf, cyc_per_mfma);
    hipFree(d);
}

int main() {
    for (int w = 1; w <= 4; ++w) {
        run<0, 0, 4>("mfma only, 4 chains", w);
        run<0, 0, 2>("mfma only, 2 chains", w);
        run<0, 0, 1>("mfma only, 1 chain", w);
        run<1, 2, 4>("interleaved 2 VALU/MFMA, 4 ch", w);
        run<1, 4, 4>("interleaved 4 VALU/MFMA, 4 ch", w);
        run<2, 2, 4>("phased 16 MFMA | 32 VALU, 4 ch", w);
        run<2, 2, 2>("phased 16 MFMA | 32 VALU, 2 ch", w);
        run<3, 2, 4>("VALU only (32 per iter)", w);
    }
    return 0;
}
