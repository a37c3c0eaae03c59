// chain_probe.hip -- what ONE dependent chain of v_mfma_f32_16x16x4_f32 costs per link, and at which shader clock, when the GPU
// runs short single-picture launches (224 blocks x 4 waves, 672 links per wave = FC1's serial K chain of the 64 x 16 shape)
// separated by host round trips -- the single-launch small pass's situation -- against the same launch repeated without gaps.
// build: hipcc --offload-arch=gfx950 -O2 -mllvm -amdgpu-mfma-vgpr-form chain_probe.hip -o chain_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_chain(int links, int chains, unsigned long long* out, float* sink) {
    unsigned long long c0, c1, r0, r1;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    if (chains == 1) {
        for (int i = 0; i < links; ++i) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
    } else if (chains == 2) {
        for (int i = 0; i < links; ++i) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[1], 0, 0, 0);
        }
    } else {
        for (int i = 0; i < links; ++i) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[3], 0, 0, 0);
        }
    }
    float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    asm volatile("s_nop 0" ::"v"(s));
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = c1 - c0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
    if (s == 123.456f) *sink = s;
}

int main() {
    unsigned long long* d;
    float* sink;
    const int blocks = 224, links = 672;
    hipMalloc((void**)&d, blocks * 16);
    hipMalloc((void**)&sink, 4);
    std::vector<unsigned long long> h(blocks * 2);
    for (int chains = 1; chains <= 4; chains *= 2)
        for (int mode = 0; mode < 2; ++mode) {  // 0: launch + synchronise per call (latency loop); 1: 200 launches back to back
            for (int i = 0; i < 300; ++i) {
                hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, links, chains, d, sink);
                if (mode == 0) hipStreamSynchronize(0);
            }
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost);
            double cyc = 0, real = 0;
            for (int b = 0; b < blocks; ++b) { cyc += (double)h[2 * b]; real += (double)h[2 * b + 1]; }
            cyc /= blocks; real /= blocks;
            printf("%d chain(s) per wave, %s: %.0f shader cycles = %.2f us for %d links x %d -> %.1f cycles per MFMA, shader clock %.2f GHz\n", chains,
                   mode == 0 ? "launch + synchronise per call" : "launches back to back      ", cyc, real / 100.0, links, chains,
                   cyc / (links * chains), cyc / (real * 10.0));
        }
    return 0;
}
