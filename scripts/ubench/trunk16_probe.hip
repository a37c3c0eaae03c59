// Development probe for plan 3's trunk (ethcnn_trunk_fast.hip compiled with -DTRUNK16_STAMPS): one wave's way through k1_trunk_f16_foldall
// in shader-clock stamps -- where a slab iteration spends its cycles (LDS fill, barriers, the S task's phases, the M / L tasks).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form -w \
//         -I../../include -I../../hevc-complexity-reduction_amd/csrc -DTRUNK16_STAMPS trunk16_probe.hip -o trunk16_probe
// tags: 0 slab top | 1 LDS filled | 2 barrier 1 passed | 6 records read + barrier 2 passed | 3 conv1 / conv2 of the four positions done |
//       4 next slab's loads issued | 5 feature pieces split + stores issued | (next 0 = conv3 + its stores done) | 7 / 8 M task | 9 / 10 L task
#include "ethcnn_trunk_fast.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace ethcnn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int W = 3840, H = 2160, NF = argc > 1 ? atoi(argv[1]) : 50;
    const int cw = W / 64, chh = (H + 63) / 64, nctu = cw * chh, N = nctu * NF;
    uint8_t* luma;
    CK(hipMalloc(&luma, (size_t)W * H * NF));
    {
        std::vector<uint8_t> h((size_t)W * H);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)((i * 2654435761u) >> 24);
        for (int f = 0; f < NF; ++f) CK(hipMemcpy(luma + (size_t)f * W * H, h.data(), h.size(), hipMemcpyHostToDevice));
    }
    char* F;
    CK(hipMalloc(&F, (size_t)((N + 31) / 32) * kFastPairBytes));
    uint16_t* wimg;
    float* cfrag;
    int* flags;
    CK(hipMalloc(&wimg, 3 * kTrunk16Halves * 2));
    CK(hipMalloc(&cfrag, 3 * kTrunk16Consts * 4));
    CK(hipMalloc(&flags, 4096));
    {
        std::vector<uint16_t> hw(3 * kTrunk16Halves);
        for (size_t i = 0; i < hw.size(); ++i) hw[i] = (uint16_t)(0x2800 + (i * 40503u) % 0x0400) | (uint16_t)((i & 1) << 15);  // halves around +-0.03
        CK(hipMemcpy(wimg, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        std::vector<float> hc(3 * kTrunk16Consts, 0.01f);
        CK(hipMemcpy(cfrag, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    }
    Trunk16Scalars sc;
    for (int b = 0; b < 3; ++b) { sc.C1[b] = 1.0f; sc.U2[b] = 0.25f; sc.U3[b] = 0.25f; }
    const int groups = (N + 15) / 16, blocks = groups < 512 ? groups : 512;
    int sel[2] = {0, 300};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_t16_block), sel, sizeof sel));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        std::vector<unsigned long long> z(2 * 4 * 512, 0ull);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_t16_stamps), z.data(), z.size() * 8));
        hipEventRecord(e0);
        hipLaunchKernelGGL(k1_trunk_f16_foldall<true>, dim3(blocks), dim3(256), 0, 0, luma, W, H, (long)W, (long)W * H, cw, nctu, 0, 0, N, flags, 16, wimg, cfrag, sc, F);
        hipEventRecord(e1);
        CK(hipEventSynchronize(e1));
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("k1_trunk_f16_foldall<true>: %d CTUs, %d blocks, best of 6: %.1f us (stamped build)\n", N, blocks, best * 1e3);
    std::vector<unsigned long long> st(2 * 4 * 512);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_t16_stamps), st.size() * 8));
    for (int b = 0; b < 2; ++b)
        for (int w = 0; w < 4; ++w) {
            const unsigned long long* p = &st[(b * 4 + w) * 512];
            printf("block %d wave %d: tag:cycles since the previous stamp\n ", sel[b], w);
            unsigned long long prev = p[0] & 0x00ffffffffffffffull;
            int col = 0;
            for (int i = 0; i < 512 && p[i]; ++i) {
                const int tag = (int)(p[i] >> 56);
                const unsigned long long t = p[i] & 0x00ffffffffffffffull;
                if (i >= 150) break;  // ~ the first three groups
                if (tag == 0 && i) { printf("\n "); col = 0; }
                printf(" %d:%llu", tag, t - prev);
                prev = t;
                ++col;
            }
            printf("\n");
        }
    return 0;
}
