// Microbenchmark 2: matrix-pipe cost of ONE extra instruction of a given kind per fp32 MFMA (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int KIND, int PER>
__global__ void k(float* out, const float* src, int iters, float a0) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = a0 + i;
    __syncthreads();
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float v[8];
    f32x2 p[4];
    for (int i = 0; i < 8; ++i) v[i] = a0 + threadIdx.x * 0.001f + i;
    for (int i = 0; i < 4; ++i) p[i] = (f32x2){v[i], v[i + 4]};
    unsigned u = threadIdx.x * 2654435761u;
    unsigned su = (unsigned)iters;
    const float a = a0, b = a0 * 0.5f;
    const f32x2 pa = (f32x2){a, b};
    const float* gp = src + (threadIdx.x & 63);
    const unsigned la = (threadIdx.x & 63) * 4;
    float ld = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m & 3] = MFMA16(a, b, acc[m & 3]);
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int j = (m * PER + q) & 7;
                if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(a));
                if (KIND == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j]) : "v"(a));
                if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j & 3]) : "v"(pa));
                if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j & 3]) : "v"(pa));
                if (KIND == 5) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(v[j]) : "v"(u));
                if (KIND == 6) asm volatile("ds_read_b32 %0, %1" : "=v"(v[j]) : "v"(la) : "memory");
                if (KIND == 7) asm volatile("s_mul_i32 %0, %0, 3" : "+s"(su));
                if (KIND == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j]) : "v"(a));
                if (KIND == 9) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u) : "v"(0x7fffffffu));
            }
        }
        if (KIND == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = ld;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + p[i][0] + p[i][1];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f + u + su) out[0] = s;
}

template <int KIND, int PER>
static double run(int wps) {
    float *d, *src;
    hipMalloc(&d, 4);
    hipMalloc(&src, 4096);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<KIND, PER><<<256, 256 * wps>>>(d, src, 100, 1.0f);
    hipEventRecord(e0);
    k<KIND, PER><<<256, 256 * wps>>>(d, src, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    hipFree(src);
    return ms * 1e-3 * 2.4e9 / (16.0 * iters * wps);  // cycles per MFMA slot per SIMD
}

#define ROW(KIND, name)                                                                                     \
    {                                                                                                       \
        printf("%-22s", name);                                                                              \
        for (int w = 1; w <= 3; ++w) {                                                                      \
            const double base = run<0, 1>(w), one = run<KIND, 1>(w), two = run<KIND, 2>(w);                 \
            printf("  w%d: +%4.1f /1  +%4.1f /2 ", w, one - base, (two - base) / 2);                        \
        }                                                                                                   \
        printf("\n");                                                                                       \
    }
int main() {
    printf("extra matrix-pipe cycles (at 2.4 GHz nominal) per added instruction, 1 or 2 per fp32 MFMA; w = waves/SIMD\n");
    printf("baseline cycles per MFMA: %.1f (1 wave/SIMD)\n", run<0, 1>(1));
    ROW(1, "v_mul_f32");
    ROW(2, "v_max_f32");
    ROW(3, "v_pk_mul_f32");
    ROW(4, "v_pk_fma_f32");
    ROW(5, "v_cvt_f32_ubyte0");
    ROW(8, "v_mov_b32");
    ROW(9, "v_and_b32");
    ROW(6, "ds_read_b32");
    ROW(7, "s_mul_i32");
    return 0;
}
