// Microbenchmark 3: matrix-pipe cost of VMEM instructions beside fp32 MFMAs (gfx950): NV of them per 16 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// KIND 0 none, 1 global_load_dwordx4 -> VGPR, 2 global_load_lds_dwordx4 (LDS-DMA), 3 global_load_dword
template <int KIND, int NV>
__global__ void k(float* out, const float* src, int iters, float a0) {
    __shared__ __attribute__((aligned(16))) float lds[8 * 1024];
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a = a0, b = a0 * 0.5f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* gp = src + ((size_t)blockIdx.x * 16 + wave) * 4096 + lane * 4;  // cache-hot after the first pass
    const unsigned lbase = (unsigned)(size_t)(lds_void*)lds + (wave & 7) * 4096;
    f32x4 r[4];
    for (int i = 0; i < 4; ++i) r[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m & 3] = MFMA16(a, b, acc[m & 3]);
            if (m < NV) {
                const float* g = gp + (m & 3) * 256;
                if (KIND == 1) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r[m & 3]) : "v"(g) : "memory");
                if (KIND == 3) asm volatile("global_load_dword %0, %1, off" : "=&v"(r[m & 3][0]) : "v"(g) : "memory");
                if (KIND == 2) {
                    const unsigned dst = __builtin_amdgcn_readfirstlane(lbase + (m & 3) * 1024);
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(dst) : "memory", "m0");
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float s = lds[threadIdx.x];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + r[i][0] + r[i][1] + r[i][2] + r[i][3];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND, int NV>
static double run(int wps, float* src) {
    float* d;
    hipMalloc(&d, 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<KIND, NV><<<256, 256 * wps>>>(d, src, 100, 1.0f);
    hipEventRecord(e0);
    k<KIND, NV><<<256, 256 * wps>>>(d, src, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return ms * 1e-3 * 2.4e9 / (1.0 * iters * wps);  // cycles per 16-MFMA iteration per wave-slot
}

#define ROW(KIND, name)                                                                           \
    {                                                                                             \
        printf("%-28s", name);                                                                    \
        for (int w = 1; w <= 3; ++w) {                                                            \
            const double base = run<0, 0>(w, src), one = run<KIND, 1>(w, src), four = run<KIND, 4>(w, src); \
            printf("  w%d: +%5.1f (1/16)  +%5.1f each (4/16)", w, one - base, (four - base) / 4);  \
        }                                                                                         \
        printf("\n");                                                                             \
    }
int main() {
    float* src;
    hipMalloc(&src, (size_t)256 * 16 * 4096 * 4);
    hipMemset(src, 0, (size_t)256 * 16 * 4096 * 4);
    printf("extra cycles per 16-MFMA iteration per added VMEM instruction (nominal 2.4 GHz), incl. the vmcnt(0) at the end of the iteration\n");
    printf("baseline: %.1f cycles per 16 MFMAs (1 wave/SIMD)\n", run<0, 0>(1, src));
    ROW(1, "global_load_dwordx4 -> VGPR");
    ROW(2, "global_load_lds_dwordx4");
    ROW(3, "global_load_dword -> VGPR");
    return 0;
}
