// Microbenchmark (round 5): how many 256-thread blocks does a gfx950 CU really hold, by LDS bytes per block and by registers per wave?
// Every block waits a fixed number of shader-clock ticks; 512 blocks are launched on 256 CUs: the launch takes one wait if two blocks are
// resident per CU and two waits if only one is.  (k1_trunk_f16_foldall: 78,912 B of LDS, 255 VGPRs -- built for two blocks per CU.)
// hipcc --offload-arch=gfx950 -O3 -o occupancy_probe occupancy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NV> __global__ void k(long long ticks, int* out);
#define PROBE(NV)                                                                                                        \
    template <> __global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(NV))) void k<NV>(long long ticks, int* out) { \
        extern __shared__ int dyn[];                                                                                      \
        if (threadIdx.x == 0) dyn[0] = 1;                                                                                 \
        const long long t0 = wall_clock64();                                                                              \
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);                                                  \
        if (dyn[0] == 12345) out[0] = 1;                                                                                  \
    }
PROBE(64)
PROBE(128)
PROBE(256)

template <int NV>
static void sweep(int blocks) {
    int* d;
    hipMalloc(&d, 4);
    const long long ticks = 100 * 100;  // wall_clock64: 100 MHz -> 100 us
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k<NV>));
    printf("registers per lane (as built) %d, %d blocks of 256 threads on 256 CUs, each waits 100 us:\n", fa.numRegs, blocks);
    const int sizes[] = {16 * 1024, 32 * 1024, 40 * 1024, 48 * 1024, 52 * 1024, 53 * 1024, 54 * 1024, 56 * 1024, 64 * 1024, 72 * 1024, 76 * 1024, 78 * 1024, 78912, 79 * 1024, 80 * 1024, 81 * 1024, 96 * 1024};
    for (int lds : sizes) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        k<NV><<<blocks, 256, lds>>>(ticks, d);
        hipEventRecord(e0);
        k<NV><<<blocks, 256, lds>>>(ticks, d);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("  LDS %6d B per block: %7.1f us -> %.1f rounds\n", lds, ms * 1e3, ms * 1e3 / 100.0);
    }
    hipFree(d);
}

int main() {
    sweep<64>(512);
    sweep<128>(512);
    sweep<256>(512);
    sweep<64>(768);
    sweep<64>(1024);
    return 0;
}
