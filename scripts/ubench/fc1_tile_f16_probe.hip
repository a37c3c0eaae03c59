// fc1_tile_f16_probe.hip -- VERDICT r04 item 7: what would the 16-bit matrix pipe buy the single-launch pass (the in-process encoder
// hook's one picture, every LDP frame)?  Its FC1 phase is the register-fed 64 x 16 tile of csrc/ethcnn_fc1_regs.h: four waves, one
// dependent chain of 672 v_mfma_f32_16x16x4_f32 per accumulator (~45 cycles per link when a picture's blocks are alone on their SIMDs).
// This probe launches exactly that phase -- ceil(n / 64) M tiles x 28 column blocks, 256 threads -- in two forms on the SAME operand
// bytes per k:
//   A  exact   fc1_tile_regs<1, 16, false>   (features [k/4][16][4] fp32, weights fc1_lane16: one dwordx4 per lane per 16 k each)
//   B  fp16x2  the plan-2 operands (features: the trunk's featb pair images; weights: the fc1_fast image) fed to
//              v_mfma_f32_16x16x32_f16 straight from memory into a register ring: per 32 k two dwordx4 per lane per operand (the two
//              pieces), three products, TWO accumulators (even / odd chunks) so that the chain is 126 links instead of 252
// and prints the launch time of each (best of 20, HIP events, launches back to back and launch + synchronise).  The pass around the
// FC1 phase is the same in both forms (the plan-2 trunk epilogue costs +128 VALU per task: ethcnn_trunk_task.h), so the difference
// here is an UPPER bound of what a plan-2 single-launch pass could gain.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form -w \
//         -I../../include -I../../hevc-complexity-reduction_amd/csrc fc1_tile_f16_probe.hip -o fc1_tile_f16_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "ethcnn_fc1_regs.h"

using namespace ethcnn;
typedef _Float16 hf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void k_exact(const float* feat, const float* wlane, const float* bias, float* out, int M) {
    const int nb = blockIdx.x % 28, mt = blockIdx.x / 28;
    fc1_tile_regs<1, 16, false>(feat, wlane, bias, out, M, mt, nb);
}

template <int D>
__global__ __launch_bounds__(256, 2) void k_f16(const char* featb, const char* wf, const float* bias, float* out, int M, float unscale) {
    constexpr int NC = kNFeat / 32;  // 84 chunks of 32 k
    const int nb = blockIdx.x % 28, mt = blockIdx.x / 28;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = lane & 15, kg = lane >> 4;
    const int m0 = mt * 64 + wv * 16;
    const int grp = min(m0 >> 4, ((M + 15) >> 4) - 1);
    const int pair = grp >> 1, par = grp & 1, t = nb >> 1, hsel = nb & 1;
    // per-lane byte offsets inside a 32-k chunk's two featb / fc1_fast 16-k chunks: chunk 2 C + (kg >> 1), k half kg & 1
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(featb) + (size_t)pair * fast_pair_bytes(2), 0, fast_pair_bytes(2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wf), 0, kFastChunks * 14 * 2 * 1024, 0x00020000);
    const int va = (kg >> 1) * 2 * 1024 + ((kg & 1) * 32 + 16 * par + col) * 16;                   // + C * 4096 (+ piece * 1024)
    const int vb = ((kg >> 1) * 14 + t) * 2 * 1024 + ((kg & 1) * 32 + 16 * hsel + col) * 16;       // + C * 2 * 14 * 2048 (+ piece * 1024)
    u4 ra[D][2], rb[D][2];
#pragma unroll
    for (int u = 0; u < D; ++u)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ra[u][q] = __builtin_amdgcn_raw_buffer_load_b128(rA, va, u * 4096 + q * 1024, 0);
            rb[u][q] = __builtin_amdgcn_raw_buffer_load_b128(rB, vb, u * (2 * 14 * 2048) + q * 1024, 0);
        }
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NC; ++u) {
        const int slot = u % D;
        f32x4& a = acc[u & 1];
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hf8, ra[slot][0]), __builtin_bit_cast(hf8, rb[slot][0]), a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hf8, ra[slot][1]), __builtin_bit_cast(hf8, rb[slot][0]), a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hf8, ra[slot][0]), __builtin_bit_cast(hf8, rb[slot][1]), a, 0, 0, 0);
        if (u + D < NC) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                ra[slot][q] = __builtin_amdgcn_raw_buffer_load_b128(rA, va, (u + D) * 4096 + q * 1024, 0);
                rb[slot][q] = __builtin_amdgcn_raw_buffer_load_b128(rB, vb, (u + D) * (2 * 14 * 2048) + q * 1024, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(out, 0, M * kNVec * 4, 0x00020000);
    const int lane_out = ((m0 + 4 * kg) * kNVec + col) * 4;
    const float bv = bias[nb * 16 + col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float h = (acc[0][r] + acc[1][r]) * unscale + bv;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(0.2f * h, h)), rO, lane_out + r * kNVec * 4, nb * 16 * 4, 0);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int cap = 2048 + 64;
    float *feat, *wlane, *bias, *out;
    char *featb, *wf;
    CK(hipMalloc(&feat, (size_t)cap * kNFeat * 4));
    CK(hipMalloc(&wlane, (size_t)kNFeat * kNVec * 4));
    CK(hipMalloc(&bias, kNVec * 4));
    CK(hipMalloc(&out, (size_t)cap * kNVec * 4));
    CK(hipMalloc(&featb, (size_t)(cap / 32) * fast_pair_bytes(2)));
    CK(hipMalloc(&wf, (size_t)kFastChunks * 14 * 2 * 1024));
    CK(hipMemset(feat, 0x3c, (size_t)cap * kNFeat * 4));
    CK(hipMemset(wlane, 0x3c, (size_t)kNFeat * kNVec * 4));
    CK(hipMemset(bias, 0, kNVec * 4));
    CK(hipMemset(featb, 0x2c, (size_t)(cap / 32) * fast_pair_bytes(2)));  // halves 0x2c2c = 0.0652
    CK(hipMemset(wf, 0x2c, (size_t)kFastChunks * 14 * 2 * 1024));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("# FC1 phase of the single-launch pass alone: ceil(n / 64) x 28 blocks of the 64 x 16 tile; us per launch, best of 20\n");
    const int sizes[4] = {96, 510, 1020, 2040};
    for (int si = 0; si < 4; ++si) {
        const int n = sizes[si], blocks = ((n + 63) / 64) * 28;
        float best[3] = {1e9f, 1e9f, 1e9f}, b2b[3] = {0, 0, 0};
        for (int form = 0; form < 3; ++form) {
            auto launch = [&]() {
                if (form == 0) hipLaunchKernelGGL(k_exact, dim3(blocks), dim3(256), 0, 0, feat, wlane, bias, out, n);
                else if (form == 1) hipLaunchKernelGGL(k_f16<8>, dim3(blocks), dim3(256), 0, 0, featb, wf, bias, out, n, 1.0f / 65536.0f);
                else hipLaunchKernelGGL(k_f16<12>, dim3(blocks), dim3(256), 0, 0, featb, wf, bias, out, n, 1.0f / 65536.0f);
            };
            for (int it = 0; it < 25; ++it) {
                hipEventRecord(e0, 0);
                launch();
                hipEventRecord(e1, 0);
                CK(hipEventSynchronize(e1));
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (it >= 5) best[form] = std::min(best[form], ms);
            }
            hipEventRecord(e0, 0);
            for (int it = 0; it < 200; ++it) launch();
            hipEventRecord(e1, 0);
            CK(hipEventSynchronize(e1));
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            b2b[form] = ms / 200.0f;
        }
        printf("n = %4d CTUs (%4d blocks): exact fp32 %6.1f us (back to back %6.1f) | fp16x2 16x16x32, ring 8: %6.1f (%6.1f) | ring 12: %6.1f (%6.1f)\n", n, blocks,
               best[0] * 1e3, b2b[0] * 1e3, best[1] * 1e3, b2b[1] * 1e3, best[2] * 1e3, b2b[2] * 1e3);
    }
    return 0;
}
