// Round 6 probe: joules per TFLOP of gfx950's matrix instructions by shape and type, on RANDOM operands (plan 3 is bound by the socket's power cap:
// profiles/r06_plan3_overlap.txt -- what would move it is fewer joules per product, so: does another instruction of the same arithmetic cost fewer?).
//   mfma_energy <kind> <seconds>     kind: 0 f32 16x16x4 | 1 f16 16x16x16 | 2 f16 16x16x32 | 3 f16 32x32x8 | 4 f16 32x32x16 | 5 bf16 32x32x16 | 6 fp8 32x32x64 | 7 i8 32x32x32
// One 4-wave block x 3 per CU (three waves per SIMD), 4 independent accumulators per wave, operands from a per-lane random table refreshed every
// iteration by a cheap xorshift on two of eight registers (so the multiplier sees changing bits, as FC1's does).  Prints achieved TFLOP/s; the wrapper
// (scripts/mfma_energy.py) samples socket power and shader clock meanwhile.
// hipcc --offload-arch=gfx950 -O3 -o mfma_energy mfma_energy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, long iters, unsigned seed) {
    unsigned r[8];
    unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    for (int i = 0; i < 8; ++i) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; r[i] = (s & 0x3fff3fffu) | 0x30003000u; }  // two fp16 in [0.125, 2): finite, mixed bits
    f32x4 a4[4];
    f32x16 a16[4];
    i32x16 ai[4];
    for (int i = 0; i < 4; ++i) { a4[i] = (f32x4){0, 0, 0, 0}; for (int j = 0; j < 16; ++j) { a16[i][j] = 0.f; ai[i][j] = 0; } }
    for (long it = 0; it < iters; ++it) {
        r[it & 7] ^= r[(it + 3) & 7] >> 3;  // keep the operands moving (cheap: one VALU per 4 MFMAs)
        r[it & 7] = (r[it & 7] & 0x3fff3fffu) | 0x30003000u;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if constexpr (KIND == 0) a4[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, (r[m] & 0x3fffffffu) | 0x3e000000u), __builtin_bit_cast(float, (r[m + 4] & 0x3fffffffu) | 0x3e000000u), a4[m], 0, 0, 0);
            else if constexpr (KIND == 1) { const h4 x = __builtin_bit_cast(h4, (unsigned long long)r[m] | ((unsigned long long)r[m + 1 & 7] << 32)), y = __builtin_bit_cast(h4, (unsigned long long)r[m + 4] | ((unsigned long long)r[(m + 5) & 7] << 32)); a4[m] = __builtin_amdgcn_mfma_f32_16x16x16f16(x, y, a4[m], 0, 0, 0); }
            else {
                const i32x4 xa = {(int)r[m], (int)r[(m + 1) & 7], (int)r[(m + 2) & 7], (int)r[(m + 3) & 7]}, ya = {(int)r[(m + 4) & 7], (int)r[(m + 5) & 7], (int)r[(m + 6) & 7], (int)r[(m + 7) & 7]};
                if constexpr (KIND == 2) a4[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, xa), __builtin_bit_cast(h8, ya), a4[m], 0, 0, 0);
                else if constexpr (KIND == 3) { const h4 x = __builtin_bit_cast(h4, (unsigned long long)r[m] | ((unsigned long long)r[(m + 1) & 7] << 32)), y = __builtin_bit_cast(h4, (unsigned long long)r[m + 4] | ((unsigned long long)r[(m + 5) & 7] << 32)); a16[m] = __builtin_amdgcn_mfma_f32_32x32x8f16(x, y, a16[m], 0, 0, 0); }
                else if constexpr (KIND == 4) a16[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, xa), __builtin_bit_cast(h8, ya), a16[m], 0, 0, 0);
                else if constexpr (KIND == 5) a16[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, xa), __builtin_bit_cast(b8, ya), a16[m], 0, 0, 0);
                else if constexpr (KIND == 6) { const i32x8 x8 = {xa[0], xa[1], xa[2], xa[3], ya[0], ya[1], ya[2], ya[3]}; a16[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x8, x8, a16[m], 0, 0, 0, 0x7f, 0, 0x7f); }
                else ai[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa, ya, ai[m], 0, 0, 0);
            }
        }
    }
    float sum = 0.f;
    for (int i = 0; i < 4; ++i) { sum += a4[i][0] + a16[i][0] + (float)ai[i][0]; }
    if (sum == 12345.678f) out[0] = sum;
}

template <int KIND>
static void run(double seconds, double flop_per_inst) {
    float* d;
    hipMalloc(&d, 4);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 3;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    long iters = 20000;
    float ms = 0;
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    iters = (long)(iters * (seconds * 1e3 / 4.0) / ms);  // four launches of seconds / 4 each
    double best = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 7u + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double tf = (double)blocks * 4 * iters * 4 * flop_per_inst / (ms * 1e-3) * 1e-12;
        if (tf > best) best = tf;
    }
    printf("%.1f\n", best);
}

int main(int argc, char** argv) {
    const int kind = argc > 1 ? atoi(argv[1]) : 4;
    const double sec = argc > 2 ? atof(argv[2]) : 3.0;
    switch (kind) {
        case 0: run<0>(sec, 2.0 * 16 * 16 * 4); break;
        case 1: run<1>(sec, 2.0 * 16 * 16 * 16); break;
        case 2: run<2>(sec, 2.0 * 16 * 16 * 32); break;
        case 3: run<3>(sec, 2.0 * 32 * 32 * 8); break;
        case 4: run<4>(sec, 2.0 * 32 * 32 * 16); break;
        case 5: run<5>(sec, 2.0 * 32 * 32 * 16); break;
        case 6: run<6>(sec, 2.0 * 32 * 32 * 64); break;
        default: run<7>(sec, 2.0 * 32 * 32 * 32); break;
    }
    return 0;
}
