// Development probe for the single-launch small pass (ethcnn_small.hip compiled with -DSMALL_STAMPS): device-wide timeline of
// its three block roles (trunk + CTU load, FC1, heads): when blocks enter, are woken, finish computing and leave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form -w \
//         -I../../include -I../../hevc-complexity-reduction_amd/csrc -DSMALL_STAMPS -DETHCNN_EXPERIMENTS small_probe.hip -o small_probe
//   small_probe W H resi pull     pull = 1: the PULL form (picture in page-locked host memory; ETHCNN_PULL_BLOCKS=k: k pull blocks)
#include "ethcnn_small.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace ethcnn { int chunks_per_frame(int nctu) { return (nctu + kSubBatch - 1) / kSubBatch; } }
using namespace ethcnn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 1920, H = argc > 2 ? atoi(argv[2]) : 1080;
    const bool resi = argc > 3 && atoi(argv[3]) != 0;
    const bool pull = argc > 4 && atoi(argv[4]) != 0;
    FrameGeom g{W, H, W, (long)W * H, (W + 63) / 64, (H + 63) / 64, ((W + 63) / 64) * ((H + 63) / 64)};
    const int n = g.nctu, cap = (n + 63) / 64 * 64 + 64;
    Workspace ws;
    DeviceWeights dw;
    uint8_t* luma;
    float *probs, *arena;
    int* sync;
    if (pull) {
        CK(hipHostMalloc((void**)&luma, (size_t)W * H, hipHostMallocDefault));
        memset(luma, 0x55, (size_t)W * H);
        CK(hipMalloc(&ws.xs, (size_t)cap * 4096));
        CK(hipMalloc(&ws.xm, (size_t)cap * 2048));
        CK(hipMalloc(&ws.xl, (size_t)cap * 512));
    } else {
        CK(hipMalloc(&luma, (size_t)W * H));
        CK(hipMemset(luma, 0x55, (size_t)W * H));
    }
    CK(hipMalloc(&ws.feat, (size_t)cap * kNFeat * 4));
    CK(hipMalloc(&ws.h1, (size_t)cap * kNVec * 4));
    CK(hipMalloc(&probs, (size_t)cap * kNOut * 4));
    const size_t wfloats = (size_t)kNFeat * kNVec;
    CK(hipMalloc(&arena, (wfloats + 65536) * 4));
    CK(hipMemset(arena, 0x3c, (wfloats + 65536) * 4));  // small positive floats everywhere
    dw.trunk_w = dw.trunk_b = arena;
    dw.fc1_img112 = dw.fc1_img64 = dw.fc1_img32 = dw.fc1_img16 = arena;
    dw.fc1_b = dw.fc1_lane16 = arena;
    for (int h = 0; h < 3; ++h) dw.fc2_w[h] = dw.fc2_b[h] = dw.fc3_w[h] = dw.fc3_b[h] = dw.fc2_lane[h] = arena;
    const int words = small_pass_sync_words(n, 1);
    CK(hipMalloc(&sync, (size_t)words * 4));
    CK(hipMemset(sync, 0, (size_t)words * 4));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    {
        std::vector<unsigned long long> z((size_t)(1 << 13) * 4, 0ull);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_small_stamps), z.data(), z.size() * 8));
    }
    float best = 1e9;
    for (int it = 0; it < 20; ++it) {
        hipEventRecord(e0, 0);
        launch_small_pass(luma, g, 0, n, resi, ws, dw, ws.h1, 0.6f, 0.5f, 0.5f, probs, 1, sync, it + 1, nullptr, 0u, 0, pull);  // (epoch = it + 1: a fresh claim tag per launch)
        hipEventRecord(e1, 0);
        CK(hipEventSynchronize(e1));
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    const int ngroups = (n + 15) / 16, ntiles = (n + 63) / 64;
    const int nT = ngroups * 4 + ngroups + ngroups;
    const char* she = getenv("ETHCNN_SMALL_SHAPE");
    const int shape = she ? atoi(she) : ((n <= 1536 || pull) ? 0 : (n <= 2304 ? 1 : 2));  // (launch_small_pass's rule)
    const int nsplit = shape == 0 ? 28 : (shape == 1 ? 14 : 7);
    const char* pe = getenv("ETHCNN_PULL_BLOCKS");
    const int nP = pull ? std::min(ngroups, pe && atoi(pe) > 0 ? atoi(pe) : (ngroups <= 32 ? 32 : 16)) : 0;
    const int nF = ntiles * nsplit, nH = resi ? 0 : ngroups * 3, nb = nP + nT + nF + nH;
    std::vector<unsigned long long> st((size_t)(1 << 13) * 4);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_small_stamps), st.size() * 8));
    printf("%dx%d: %d CTUs, %s%s; blocks: trunk %d (S %d, M %d, L %d) + FC1 %d (64 x %d) + heads %d = %d; launch %.1f us by HIP events (best of 20)\n",
           W, H, n, resi ? "LDP front-end" : "All-Intra", pull ? " PULL form" : "", nT, ngroups * 4, ngroups, ngroups, nF, 448 / nsplit, nH, nb, best * 1e3);
    if (nb > (1 << 13)) { printf("(more blocks than stamp slots)\n"); return 0; }
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < nb; ++b) { t0 = std::min(t0, st[b * 4]); t1 = std::max(t1, st[b * 4 + 3]); }
    printf("first block entry -> last block exit %.1f us\n", (t1 - t0) / 100.0);
    struct Role { const char* name; int b0, b1; };
    std::vector<Role> roles;
    if (pull) {  // trunk blocks are ordered group by group (6 per group); shown by quarter of the picture
        roles.push_back({"pull", 0, nP});
        for (int q = 0; q < 4; ++q) roles.push_back({q == 0 ? "trunk q1" : q == 1 ? "trunk q2" : q == 2 ? "trunk q3" : "trunk q4", nP + 6 * (ngroups * q / 4), nP + 6 * (ngroups * (q + 1) / 4)});
        for (int q = 0; q < 4; ++q) roles.push_back({q == 0 ? "FC1 q1" : q == 1 ? "FC1 q2" : q == 2 ? "FC1 q3" : "FC1 q4", nP + nT + nsplit * (ntiles * q / 4), nP + nT + nsplit * (ntiles * (q + 1) / 4)});
        for (int q = 0; q < 4; ++q) roles.push_back({q == 0 ? "heads q1" : q == 1 ? "heads q2" : q == 2 ? "heads q3" : "heads q4", nP + nT + nF + 3 * (ngroups * q / 4), nP + nT + nF + 3 * (ngroups * (q + 1) / 4)});
    } else {
        roles = {{"trunk S", 0, ngroups * 4}, {"trunk M", ngroups * 4, ngroups * 5}, {"trunk L", ngroups * 5, nT}, {"FC1", nT, nT + nF}, {"heads", nT + nF, nb}};
    }
    printf("%-8s %6s | %-21s | %-21s | %-21s | %-21s\n", "role", "blocks", "entry  min/avg/max", "woken  min/avg/max", "computed min/avg/max", "exit   min/avg/max");
    for (const Role& r : roles) {
        if (r.b1 <= r.b0) continue;
        printf("%-8s %6d", r.name, r.b1 - r.b0);
        for (int s = 0; s < 4; ++s) {
            double mn = 1e18, mx = 0, sum = 0;
            bool unset = false;  // (trunk blocks stamp entry and exit only)
            for (int b = r.b0; b < r.b1; ++b) {
                if (st[b * 4 + s] == 0) { unset = true; break; }
                const double v = (st[b * 4 + s] - t0) / 100.0; mn = std::min(mn, v); mx = std::max(mx, v); sum += v;
            }
            if (unset) printf(" | %6s %6s %6s ", "-", "-", "-");
            else printf(" | %6.1f %6.1f %6.1f ", mn, sum / (r.b1 - r.b0), mx);
        }
        printf("\n");
    }
    return 0;
}
