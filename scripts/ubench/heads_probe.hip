// Development probe for k_heads (ethcnn_heads.hip compiled with -DHEADS_STAMPS): where does the heads stage's time go?
// Wave 0 of every block stamps s_memtime (shader clock; the counter is per XCC, so only differences inside a block are
// used) at entry / first DMA issued / first W2 chunk landed / K loop done / exit, and s_memrealtime (100 MHz, one counter
// for the device) at entry and exit.  Output: mean / max phase lengths per head, and the device-wide timeline of resident
// blocks per head -- how long the launch runs full, and what its tail is made of.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form -w \
//         -I../../include -I../../hevc-complexity-reduction_amd/csrc -DHEADS_STAMPS heads_probe.hip -o heads_probe
//   ./heads_probe [CTUs = 102000]
#include "ethcnn_heads.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace ethcnn { int chunks_per_frame(int nctu) { return (nctu + kSubBatch - 1) / kSubBatch; } }
using namespace ethcnn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 102000;
    float *h1, *probs, *wbuf;
    int* flags;
    CK(hipMalloc(&h1, (size_t)n * kNVec * 4));
    CK(hipMalloc(&probs, (size_t)n * kNOut * 4));
    CK(hipMalloc(&flags, 4 * 1024 * 4));
    CK(hipMemset(flags, 0, 4 * 1024 * 4));
    std::vector<float> hh((size_t)n * kNVec);
    unsigned s = 12345;
    for (auto& v : hh) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    CK(hipMemcpy(h1, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
    const int n1[3] = {64, 128, 256}, n2[3] = {48, 96, 192}, n3[3] = {1, 4, 16};
    size_t tot = 0;
    for (int h = 0; h < 3; ++h) tot += (size_t)(n1[h] + 1) * n2[h] + n2[h] + (size_t)(n2[h] + 1) * n3[h] + n3[h];
    std::vector<float> hw(tot);
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.1f; }
    CK(hipMalloc(&wbuf, tot * 4));
    CK(hipMemcpy(wbuf, hw.data(), tot * 4, hipMemcpyHostToDevice));
    Workspace ws;
    DeviceWeights dw;
    ws.h1 = h1;
    ws.flags = flags;
    size_t o = 0;
    for (int h = 0; h < 3; ++h) {
        dw.fc2_w[h] = wbuf + o; o += (size_t)(n1[h] + 1) * n2[h];
        dw.fc2_b[h] = wbuf + o; o += n2[h];
        dw.fc3_w[h] = wbuf + o; o += (size_t)(n2[h] + 1) * n3[h];
        dw.fc3_b[h] = wbuf + o; o += n3[h];
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0, 0);
        launch_heads(ws, dw, n, 0.6f, 2040, 0L, 0.5f, 0.5f, probs, 0);
        hipEventRecord(e1, 0);
        CK(hipEventSynchronize(e1));
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    const int gx = (n + 63) / 64, nb = 3 * gx;  // launch_heads: grid (tiles, 3), blockIdx.y = 0 / 1 / 2 -> head 16 / 32 / 64
    const char* name[3] = {"head 16", "head 32", "head 64"};
    const double mfma[3] = {816, 216, 60};  // MFMAs per wave: FC2 + FC3
    std::vector<unsigned long long> st((size_t)(1 << 16) * 8);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_heads_stamps), st.size() * 8));
    if (nb > (1 << 16)) { printf("too many blocks for the stamp table\n"); return 1; }
    printf("n = %d CTUs, grid (%d, 3), kernel %.1f us by HIP events (best of 6; stamps are the last launch's)\n", n, gx, best * 1e3);
    printf("ideal: %.1f us of fp32-MFMA time (1092 MFMAs x 32 clocks per 16 CTUs, 1024 SIMDs, 2.39 GHz)\n", (n / 16.0) * 1092 * 32 / 1024 / 2390.0);
    for (int y = 0; y < 3; ++y) {
        double sum[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0};
        for (int x = 0; x < gx; ++x) {
            const unsigned long long* p = &st[(size_t)(y * gx + x) * 8];
            const double d[4] = {double(p[1] - p[0]), double(p[2] - p[1]), double(p[3] - p[2]), double(p[4] - p[3])};
            for (int i = 0; i < 4; ++i) { sum[i] += d[i]; mx[i] = std::max(mx[i], d[i]); }
        }
        printf("%s (MFMA time of a wave %5.0f clocks): setup %5.0f (max %5.0f) | first chunk landed %5.0f (max %5.0f) | K loop %6.0f (max %6.0f) | FC2 epilogue + FC3 + sigmoid %5.0f (max %5.0f)\n",
               name[y], mfma[y] * 32, sum[0] / gx, mx[0], sum[1] / gx, mx[1], sum[2] / gx, mx[2], sum[3] / gx, mx[3]);
    }
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < nb; ++b) { t0 = std::min(t0, st[b * 8 + 6]); t1 = std::max(t1, st[b * 8 + 7]); }
    printf("device timeline: first block entry -> last block exit %.2f us\n", (t1 - t0) / 100.0);
    for (int k = 0; k < 16; ++k) {
        const unsigned long long t = t0 + (t1 - t0) * (2 * k + 1) / 32;
        int res[3] = {0, 0, 0};
        for (int b = 0; b < nb; ++b)
            if (st[b * 8 + 6] <= t && t < st[b * 8 + 7]) res[b / gx]++;
        printf("  t = %6.1f us: resident blocks  head 16 %4d  head 32 %4d  head 64 %4d   total %4d\n", (t - t0) / 100.0, res[0], res[1], res[2], res[0] + res[1] + res[2]);
    }
    for (int y = 0; y < 3; ++y) {
        double first_end = 1e18, last_end = 0, last_start = 0;
        for (int x = 0; x < gx; ++x) {
            const unsigned long long* p = &st[(size_t)(y * gx + x) * 8];
            first_end = std::min(first_end, (double)(p[7] - t0));
            last_end = std::max(last_end, (double)(p[7] - t0));
            last_start = std::max(last_start, (double)(p[6] - t0));
        }
        printf("  %s: first block done %.1f us, last block started %.1f us, last block done %.1f us\n", name[y], first_end / 100, last_start / 100, last_end / 100);
    }
    return 0;
}
