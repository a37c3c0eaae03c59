// Development probe: do the heads stage of pass i and the trunk of pass i+1 gain from running at the same time
// (two streams), or is it a wash?  Links against libethcnn.so's internal launchers (ethcnn::launch_*).
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I../../include -I../../hevc-complexity-reduction_amd/csrc overlap_probe.cpp \
//         -L../../hevc-complexity-reduction_amd/lib -lethcnn -Wl,-rpath,'$ORIGIN/../../hevc-complexity-reduction_amd/lib' -o overlap_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ethcnn_kernels.h"
using namespace ethcnn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 102000;
    const int cap = (n + 127) / 128 * 128 + 128;
    Workspace ws;
    DeviceWeights dw;
    float* probs;
    CK(hipMalloc(&ws.xs, (size_t)cap * 4096));
    CK(hipMalloc(&ws.xm, (size_t)cap * 2048));
    CK(hipMalloc(&ws.xl, (size_t)cap * 512));
    CK(hipMalloc(&ws.feat, (size_t)cap * kNFeat * 4));
    CK(hipMalloc(&ws.h1, (size_t)cap * kNVec * 4));
    CK(hipMalloc(&ws.flags, 4096 * 4));
    CK(hipMalloc(&probs, (size_t)cap * kNOut * 4));
    CK(hipMemset(ws.xs, 0x55, (size_t)cap * 4096));
    CK(hipMemset(ws.xm, 0x01, (size_t)cap * 2048));
    CK(hipMemset(ws.xl, 0x01, (size_t)cap * 512));
    CK(hipMemset(ws.h1, 0x3c, (size_t)cap * kNVec * 4));
    CK(hipMemset(ws.flags, 0, 4096 * 4));
    CK(hipMalloc(&dw.trunk_w, 3 * kTrunkWFrags * 64 * 4));
    CK(hipMalloc(&dw.trunk_b, 3 * kTrunkBFrags * 64 * 4));
    CK(hipMemset(dw.trunk_w, 0x3c, 3 * kTrunkWFrags * 64 * 4));
    CK(hipMemset(dw.trunk_b, 0, 3 * kTrunkBFrags * 64 * 4));
    const int n1[3] = {64, 128, 256}, n2[3] = {48, 96, 192}, n3[3] = {1, 4, 16};
    for (int h = 0; h < 3; ++h) {
        CK(hipMalloc(&dw.fc2_w[h], (size_t)(n1[h] + 1) * n2[h] * 4)); CK(hipMemset(dw.fc2_w[h], 0x3c, (size_t)(n1[h] + 1) * n2[h] * 4));
        CK(hipMalloc(&dw.fc2_b[h], n2[h] * 4)); CK(hipMemset(dw.fc2_b[h], 0, n2[h] * 4));
        CK(hipMalloc(&dw.fc3_w[h], (size_t)(n2[h] + 1) * n3[h] * 4)); CK(hipMemset(dw.fc3_w[h], 0x3c, (size_t)(n2[h] + 1) * n3[h] * 4));
        CK(hipMalloc(&dw.fc3_b[h], n3[h] * 4)); CK(hipMemset(dw.fc3_b[h], 0, n3[h] * 4));
    }
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    auto heads = [&](hipStream_t s) { launch_heads(ws, dw, n, 0.6f, 2040, 0L, 0.5f, 0.5f, probs, s); launch_gate(ws, n, 2040, 0L, 0.5f, probs, s); };
    auto trunk = [&](hipStream_t s) { launch_trunk(ws, dw, n, false, s); };
    auto timeit = [&](int mode) -> double {  // 0: heads only, 1: trunk only, 2: heads then trunk on one stream, 3: heads || trunk on two streams
        double best = 1e9;
        for (int it = 0; it < 8; ++it) {
            (void)hipDeviceSynchronize();
            const double t0 = now();
            if (mode == 0 || mode == 2) heads(sa);
            if (mode == 1 || mode == 2) trunk(sa);
            if (mode == 3) { heads(sa); trunk(sb); }
            (void)hipStreamSynchronize(sa);
            (void)hipStreamSynchronize(sb);
            best = std::min(best, now() - t0);
        }
        return best * 1e6;
    };
    for (int rep = 0; rep < 2; ++rep) {
        const double h = timeit(0), t = timeit(1), seq = timeit(2), par = timeit(3);
        printf("n = %d: heads+gate %.1f us | trunk %.1f us | same stream %.1f us | two streams %.1f us  (host-timed, launch + sync included)\n", n, h, t, seq, par);
    }
    return 0;
}
