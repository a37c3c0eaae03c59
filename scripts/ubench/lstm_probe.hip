// Development probe for the two LSTM launches of an LDP frame (ethcnn_lstm.hip compiled with -DLSTM_STAMPS): device-wide timeline of
// their blocks: entry, operands staged (cell) / fc2 exchanged (heads), chain done (cell) / outputs stored (heads), exit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form -w \
//         -I../../include -I../../hevc-complexity-reduction_amd/csrc -DLSTM_STAMPS lstm_probe.hip -o lstm_probe
#include "ethcnn_lstm.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace ethcnn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 1920, H = argc > 2 ? atoi(argv[2]) : 1080;
    const int one = argc > 3 ? atoi(argv[3]) : 0;  // 1: the one-launch frame kernel (k_lstm_frame)
    const int n = ((W + 63) / 64) * ((H + 63) / 64);
    float *blob, *vec, *s0, *s1, *probs;
    int* gate;
    const size_t bf = kLstmBlobFloats + kLstmPackFloats;
    CK(hipMalloc(&blob, bf * 4));
    CK(hipMemset(blob, 0x3c, bf * 4));  // small positive floats everywhere
    CK(hipMalloc(&vec, (size_t)(n + 64) * kNVec * 4));
    CK(hipMemset(vec, 0x3c, (size_t)(n + 64) * kNVec * 4));
    CK(hipMalloc(&s0, (size_t)(n + 64) * 2 * kNVec * 4));
    CK(hipMalloc(&s1, (size_t)(n + 64) * 2 * kNVec * 4));
    CK(hipMemset(s0, 0x3c, (size_t)(n + 64) * 2 * kNVec * 4));
    CK(hipMalloc(&probs, (size_t)(n + 64) * kNOut * 4));
    const int gw = lstm_frame_words(n);
    CK(hipMalloc(&gate, gw * 4));
    CK(hipMemset(gate, 0, gw * 4));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 20; ++it) {
        hipEventRecord(e0, 0);
        launch_lstm(vec, s0, s1, blob, n, 32, 3, 0.5f, 0.5f, nullptr, probs, gate, nullptr, 0u, one, it + 1, 0);
        hipEventRecord(e1, 0);
        CK(hipEventSynchronize(e1));
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> st((size_t)2 * (1 << 11) * 4);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_lstm_stamps), st.size() * 8));
    const int groups = (n + 15) / 16, cg = groups >= 12 ? 2 : 1, gx = (groups + cg - 1) / cg;
    printf("%dx%d: %d CTUs; %s: cell<%d> %d x 28 blocks, heads %d x 3 blocks; %.1f us by HIP events (best of 20)\n", W, H, n,
           one ? "ONE launch (k_lstm_frame)" : "two launches", cg, gx, groups, best * 1e3);
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < gx * 28; ++b) t0 = std::min(t0, st[b * 4]);
    struct Role { const char* name; int k, b0, b1; } roles[] = {
        {"cell 16 (256 units)", 0, 0, gx * 16}, {"cell 32 (128)", 0, gx * 16, gx * 24}, {"cell 64 (64)", 0, gx * 24, gx * 28},
        {"heads 16", 1, 0, groups}, {"heads 32", 1, groups, 2 * groups}, {"heads 64", 1, 2 * groups, 3 * groups}};
    if (one) {
        for (int i = 3; i < 6; ++i) { roles[i].b0 += gx * 28; roles[i].b1 += gx * 28; }  // heads blocks follow the cell blocks in ONE grid
    }
    printf("%-20s %6s | %-21s | %-21s | %-21s | %-21s\n", "role", "blocks", "entry  min/avg/max", "staged / exchanged", "chain / stored", "exit   min/avg/max");
    for (const Role& r : roles) {
        if (r.b1 > (1 << 11)) { printf("%s: more blocks than stamp slots\n", r.name); continue; }
        printf("%-20s %6d", r.name, r.b1 - r.b0);
        for (int s = 0; s < 4; ++s) {
            double mn = 1e18, mx = 0, sum = 0;
            for (int b = r.b0; b < r.b1; ++b) {
                const double v = (double)(long long)(st[((size_t)r.k * (1 << 11) + b) * 4 + s] - t0) / 100.0;
                mn = std::min(mn, v); mx = std::max(mx, v); sum += v;
            }
            printf(" | %6.1f %6.1f %6.1f ", mn, sum / (r.b1 - r.b0), mx);
        }
        printf("\n");
    }
    return 0;
}
