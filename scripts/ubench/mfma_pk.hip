// Microbenchmark 5: what does ONE instruction of a given kind cost the matrix pipe when it is issued beside fp32 MFMAs
// (two per MFMA; 1 and 3 waves per SIMD -- 3 is the trunk's occupancy)?  A cost table over the VALU / LDS opcodes the
// kernels use or could use: plain fp32, min/max, integer, conversions, SDWA, packed fp32, transcendental, cross-lane.
// The kernel below is generated from an op list (one `if (KIND == n) asm(...)` line each).   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int KIND, int VPM>
__global__ void k(float* out, int iters, float a0) {
    __shared__ float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = a0;
    __syncthreads();
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float v[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = a0 + threadIdx.x * 0.001f + i; p[i] = (f32x2){v[i], v[i] + 1.f}; }
    const float a = a0, b = a0 * 0.5f;
    const f32x2 a2 = {a, a}, b2 = {b, b};
    const unsigned la = (threadIdx.x & 63) * 4;
    unsigned sg = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m % 4] = MFMA16(a, b, acc[m % 4]);
#pragma unroll
            for (int q = 0; q < VPM; ++q) {
                const int i = (m * VPM + q) % 8;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
                if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
                if (KIND == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
                if (KIND == 4) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
                if (KIND == 5) asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(a), "v"(b));
                if (KIND == 6) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
                if (KIND == 7) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
                if (KIND == 8) asm volatile("v_maximum3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
                if (KIND == 9) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
                if (KIND == 10) asm volatile("v_max_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
                if (KIND == 11) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
                if (KIND == 12) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
                if (KIND == 13) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
                if (KIND == 14) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(v[i]) : "v"(a));
                if (KIND == 15) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(v[i]));
                if (KIND == 16) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(v[i]));
                if (KIND == 17) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
                if (KIND == 18) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a) : );
                if (KIND == 19) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(v[i]), "v"(a) : "vcc");
                if (KIND == 20) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(v[i]));
                if (KIND == 21) asm volatile("v_cvt_f32_ubyte3 %0, %0" : "+v"(v[i]));
                if (KIND == 22) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(v[i]));
                if (KIND == 23) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(v[i]));
                if (KIND == 24) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[i]));
                if (KIND == 25) asm volatile("v_mul_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(v[i]) : "v"(a));
                if (KIND == 26) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(v[i]) : "s"(a));
                if (KIND == 27) asm volatile("v_fmamk_f32 %0, %0, 0x39808081, %1" : "+v"(v[i]) : "v"(b));
                if (KIND == 28) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
                if (KIND == 29) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(a));
                if (KIND == 30) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(a));
                if (KIND == 31) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
                if (KIND == 32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(a2));
                if (KIND == 33) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a2), "v"(b2));
                if (KIND == 34) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(a2));
                if (KIND == 35) asm volatile("v_pk_mov_b32 %0, %1, %1" : "+v"(p[i]) : "v"(a2));
                if (KIND == 36) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                if (KIND == 37) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
                if (KIND == 38) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sg) : "v"(v[i]));
                if (KIND == 39) asm volatile("s_nop 0");
                if (KIND == 40) asm volatile("ds_read_b32 %0, %1" : "=v"(v[i]) : "v"(la) : "memory");
                if (KIND == 41) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(v[i]) : "v"(la), "v"(a) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = (float)sg;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
    if (s == 12345.678f) out[0] = s + lds[0];
}

template <int KIND, int VPM>
static void run(const char* name, int w) {
    float* d;
    hipMalloc(&d, 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<KIND, VPM><<<256, 256 * w>>>(d, 100, 1.0f);
    hipEventRecord(e0);
    k<KIND, VPM><<<256, 256 * w>>>(d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.39e9 / (16.0 * iters * w);
    printf("%-28s x%d per MFMA, %d waves/SIMD: %6.2f cycles per MFMA slot -> %5.2f cycles per instruction\n", name, VPM, w, cyc, (cyc - 32.0) / VPM);
    hipFree(d);
}

int main() {
    for (int w : {1, 3}) {
        run<0, 2>("v_fma_f32", w);
        run<1, 2>("v_mul_f32", w);
        run<2, 2>("v_add_f32", w);
        run<3, 2>("v_max_f32", w);
        run<4, 2>("v_min_f32", w);
        run<5, 2>("v_max_f32 (dst != src)", w);
        run<6, 2>("v_med3_f32", w);
        run<7, 2>("v_max3_f32", w);
        run<8, 2>("v_maximum3_f32", w);
        run<9, 2>("v_max_i32", w);
        run<10, 2>("v_max_u32", w);
        run<11, 2>("v_and_b32", w);
        run<12, 2>("v_xor_b32", w);
        run<13, 2>("v_add_u32", w);
        run<14, 2>("v_lshl_add_u32", w);
        run<15, 2>("v_lshrrev_b32", w);
        run<16, 2>("v_bfe_u32", w);
        run<17, 2>("v_perm_b32", w);
        run<18, 2>("v_cndmask_b32 (vcc)", w);
        run<19, 2>("v_cmp_gt_f32 (vcc)", w);
        run<20, 2>("v_cvt_f32_ubyte0", w);
        run<21, 2>("v_cvt_f32_ubyte3", w);
        run<22, 2>("v_cvt_f32_u32", w);
        run<23, 2>("v_cvt_f32_i32", w);
        run<24, 2>("v_cvt_f32_f16", w);
        run<25, 2>("v_mul_f32 sdwa BYTE_0", w);
        run<26, 2>("v_mul_f32 (e64, sgpr src)", w);
        run<27, 2>("v_fmamk_f32 (literal)", w);
        run<28, 2>("v_fmac_f32", w);
        run<29, 2>("v_mov_b32", w);
        run<30, 2>("v_mov_b32 dpp row_shr:1", w);
        run<31, 2>("v_dot4_u32_u8", w);
        run<32, 2>("v_pk_mul_f32", w);
        run<33, 2>("v_pk_fma_f32", w);
        run<34, 2>("v_pk_add_f32", w);
        run<35, 2>("v_pk_mov_b32", w);
        run<36, 2>("v_exp_f32", w);
        run<37, 2>("v_rcp_f32", w);
        run<38, 2>("v_readfirstlane_b32", w);
        run<39, 2>("s_nop 0 (SALU reference)", w);
        run<40, 2>("ds_read_b32", w);
        run<41, 2>("ds_bpermute_b32", w);
    }
    return 0;
}
