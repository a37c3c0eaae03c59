// conv2_pad_probe.hip -- VERDICT r04 item 3, the first "exact-path lever" of DESIGN 6.1: conv2's 24 output channels occupy two 16-row
// tiles of v_mfma_f32_16x16x4_f32, so every position issues 32 MFMAs for 24 rows of work (240 per task for 208 useful).  The only
// pad-free form that keeps the trunk's "accumulators are the next layer's B operand" layout is the blocked shape
// v_mfma_f32_4x4x1_16B_f32 for channels 16..23: 16 independent 4x4 blocks per instruction, block (g, CTU quad) multiplying ITS lanes'
// channels (ci = 4 g + r) -- each block then holds a PARTIAL sum over a quarter of K, and the four g blocks of a CTU quad have to be
// added across lanes (g = lane >> 4: two cross-row exchanges + adds per value), which also replaces the single fmaf chain of the
// canonical order by four partial chains (oracle, single-launch pass and LDP front-end would have to follow).
// This probe prices the two forms per conv2 position at the trunk's occupancy (3 waves per SIMD, 256 CUs busy):
//   A  shipped: tile 0 (16 MFMA 16x16x4) + tile 1 (16 MFMA 16x16x4, half of its rows padding)
//   B  packed:  tile 0 (16 MFMA 16x16x4) + channels 16..23 as 2 x 16 MFMA 4x4x1_16B + the cross-g reduction of the 8 results per lane
// build: hipcc --offload-arch=gfx950 -O2 -mllvm -amdgpu-mfma-vgpr-form conv2_pad_probe.hip -o conv2_pad_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFMA4B(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

template <int FORM>
__global__ __launch_bounds__(256) void k_probe(int positions, unsigned long long* out, float* sink, const float* wsrc) {
    extern __shared__ float pad[];  // 48 KB requested at launch: three blocks per CU, the trunk's occupancy
    const int lane = threadIdx.x & 63;
    float w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = wsrc[i * 64 + lane];
    f32x4 c1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) c1[q] = (f32x4){1.0f + lane * 1e-3f, 0.5f, -0.25f + q, 0.125f};
    f32x4 keep = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
    for (int p = 0; p < positions; ++p) {
        f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
        for (int q1 = 0; q1 < 4; ++q1)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a0 = MFMA16(w[4 * q1 + r], c1[q1][r], a0);
                if (FORM == 0) a1 = MFMA16(w[(4 * q1 + r + 5) & 15], c1[q1][r], a1);
            }
        if (FORM == 1) {
            f32x4 b0 = (f32x4){0.f, 0.f, 0.f, 0.f}, b1 = b0;  // channels 16..19 / 20..23 of the lane's CTU, partial over its own g
#pragma unroll
            for (int q1 = 0; q1 < 4; ++q1)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    b0 = MFMA4B(w[4 * q1 + r], c1[q1][r], b0);
                    b1 = MFMA4B(w[(4 * q1 + r + 3) & 15], c1[q1][r], b1);
                }
            // sum over g (lanes l, l ^ 16, l ^ 32, l ^ 48): every lane ends with the full sums (what conv3's B operand needs)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                b0[r] += __shfl_xor(b0[r], 16);
                b0[r] += __shfl_xor(b0[r], 32);
                b1[r] += __shfl_xor(b1[r], 16);
                b1[r] += __shfl_xor(b1[r], 32);
            }
            a1 = b0 + b1;
        }
        keep += a0 + a1;
        c1[p & 3] = keep * 1e-3f;  // the next position depends on this one (as conv1 -> conv2 does through the registers)
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (keep[0] + keep[1] + keep[2] + keep[3] == 123.456f) *sink = keep[0] + pad[0];
}

int main() {
    unsigned long long* d;
    float *sink, *w;
    const int blocks = 768, positions = 2000;
    hipMalloc((void**)&d, blocks * 8);
    hipMalloc((void**)&sink, 4);
    hipMalloc((void**)&w, 16 * 64 * 4);
    std::vector<float> hw(16 * 64);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.01f * (float)((i * 37) % 19) - 0.09f;
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    std::vector<unsigned long long> h(blocks);
    for (int form = 0; form < 2; ++form) {
        for (int rep = 0; rep < 3; ++rep) {
            if (form == 0) hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(256), 48 * 1024, 0, positions, d, sink, w);
            else hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(256), 48 * 1024, 0, positions, d, sink, w);
        }
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
        double cyc = 0;
        for (int b = 0; b < blocks; ++b) cyc += (double)h[b];
        cyc /= blocks;
        printf("form %s: %.0f shader cycles per conv2 position per wave (3 waves per SIMD) -> %.0f per SIMD\n",
               form == 0 ? "A shipped (32 x 16x16x4, tile 1 half padding)        " : "B packed  (16 x 16x16x4 + 32 x 4x4x1_16B + g reduction)", cyc / positions,
               cyc / positions / 3.0);
    }
    return 0;
}
