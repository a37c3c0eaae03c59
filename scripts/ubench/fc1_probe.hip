// Development probe for the FC1 bulk launch (ethcnn_dense.hip compiled with -DFC1_STAMPS): device-wide timeline of
// resident blocks -- how long the launch runs full (3 blocks per CU) and how long its ragged end is.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form -w \
//         -I../../include -I../../hevc-complexity-reduction_amd/csrc -DFC1_STAMPS fc1_probe.hip -o fc1_probe
#include "ethcnn_dense.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace ethcnn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 102000;
    const int cap = (n + 127) / 128 * 128 + 128;
    Workspace ws;
    DeviceWeights dw;
    float* out;
    CK(hipMalloc(&ws.feat, (size_t)cap * kNFeat * 4));
    CK(hipMalloc(&out, (size_t)cap * kNVec * 4));
    CK(hipMalloc(&dw.fc1_img112, (size_t)kNFeat * kNVec * 4));
    CK(hipMalloc(&dw.fc1_b, kNVec * 4));
    CK(hipMemset(ws.feat, 0x3c, (size_t)cap * kNFeat * 4));  // small positive floats
    CK(hipMemset(dw.fc1_img112, 0x3c, (size_t)kNFeat * kNVec * 4));
    CK(hipMemset(dw.fc1_b, 0, kNVec * 4));
    dw.fc1_img64 = dw.fc1_img32 = dw.fc1_img16 = dw.fc1_img112;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0, 0);
        launch_fc1(ws, dw, n, out, 0);
        hipEventRecord(e1, 0);
        CK(hipEventSynchronize(e1));
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    const int big_tiles = ((n / 128) * 4 / 256) * 256 / 4, row0 = big_tiles * 128;
    const int rem_tiles = (n - row0 + 63) / 64;
    const int rem_blocks = (rem_tiles + 7) / 8 * 8 * 4, main_blocks = (big_tiles + 7) / 8 * 8 * 4, nb = rem_blocks + main_blocks;
    std::vector<unsigned long long> st((size_t)(1 << 14) * 2);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_fc1_stamps), st.size() * 8));
    printf("n = %d rows: %d bulk blocks (128 x 112) + %d remainder blocks (64 x 112, first), FC1 stage %.1f us by HIP events (best of 5)\n", n, main_blocks, rem_blocks, best * 1e3);
    printf("ideal: %.1f us of fp32-MFMA time\n", (double)n * 2688 * 448 / 1024.0 * 32 / (1024.0 * 2390.0));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < nb; ++b) { if (!st[b * 2]) continue; t0 = std::min(t0, st[b * 2]); t1 = std::max(t1, st[b * 2 + 1]); }
    printf("first block entry -> last block exit %.1f us\n", (t1 - t0) / 100.0);
    const int K = 40;
    for (int k = 0; k < K; ++k) {
        const unsigned long long t = t0 + (t1 - t0) * (2 * k + 1) / (2 * K);
        int res[2] = {0, 0};
        for (int b = 0; b < nb; ++b)
            if (st[b * 2] && st[b * 2] <= t && t < st[b * 2 + 1]) res[b >= rem_blocks]++;
        if (k < 3 || k >= K - 12 || k % 8 == 0) printf("  t = %7.1f us: resident  remainder %4d  bulk %4d   (768 slots)\n", (t - t0) / 100.0, res[0], res[1]);
    }
    double dsum[2] = {0, 0}; int cnt[2] = {0, 0};
    for (int b = 0; b < nb; ++b) if (st[b * 2]) { dsum[b >= rem_blocks] += (st[b * 2 + 1] - st[b * 2]) / 100.0; cnt[b >= rem_blocks]++; }
    printf("mean block lifetime: remainder %.1f us (%d blocks), bulk %.1f us (%d blocks)\n", cnt[0] ? dsum[0] / cnt[0] : 0, cnt[0], cnt[1] ? dsum[1] / cnt[1] : 0, cnt[1]);
    return 0;
}
