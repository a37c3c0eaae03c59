// ethcnn_trunk.hip -- k1: 16x16 block-mean removal (zero_mean_norm_local, net_CNN.py:78-84) and the
// three non-overlapping convs (non_overlap_conv, :86-92,127-141) of the 21 units of every CTU,
// written as `h_conv_flat` (:143-150).  gfx950, v_mfma_f32_16x16x4_f32.
//
// One wave = one task = the SAME unit position of 16 consecutive CTUs (a group): 16 S tasks, 4 M
// tasks and 1 L task per group, 240 MFMAs each.  lane = col + 16 g: col = CTU within the group
// (MFMA column), g = MFMA k-group.  The trunk runs "transposed" (rows = output channels): MFMA
// D[row][col] lives in lane (col, g) as rows 4g..4g+3, which is exactly B[k = g][col] for the 4
// k-steps r = 0..3 of the next layer when that layer's K is enumerated as (patch, r, g) with
// ci = 4g + r.  So each layer's accumulator registers ARE the next layer's B operand: no LDS, no
// cross-lane traffic (one lane-half swap for conv3's channels 16..23).  conv1's A operands and the
// biases stay in VGPRs, the 80 conv2 / conv3 A fragments of the branch in 21 KB of LDS (one
// conflict-free ds_read_b32 per MFMA costs the matrix pipe ~0.6 clocks).
//
// Three waves per SIMD (156 VGPRs) overlap one wave's VALU epilogues with the others' MFMAs; the
// next task's pixel record is prefetched a task ahead.  (A deeper in-wave software pipeline -- conv1(q+1) before leaky(conv1(q)) --
// was measured: it needs > 256 VGPRs, drops to one wave per SIMD and is 20 % slower.)  The task loop sits at 98 % of its
// instruction-mix bound: 240 MFMAs + 375 VALU (each already the cheapest instruction that computes it exactly) + 98 LDS
// + 14 VMEM = ~9960 predicted against 10,165 measured clocks per task.
// Features are written as feat[group][k/4][16][4]: every store is a 256-byte run per k-group.
//
// Arithmetic contract (DESIGN.md "canonical order"; oracle/ethcnn_oracle.c mode 0 restates it):
// each accumulator is one fmaf chain in MFMA k order starting from the bias; leaky = max(0.2h, h);
// v = fma(float(pixel sum), c255 * 2^-p, -mean) with exact integer pixel sums.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ethcnn_kernels.h"
#include "ethcnn_trunk_task.h"

namespace ethcnn {

template <bool RESI, int FAST = 0>
__global__ __launch_bounds__(256) void k1_trunk(const uint4* __restrict__ XS, const uint4* __restrict__ XM,
                                                const uint4* __restrict__ XL, int N, int bS, int bM,
                                                const float* __restrict__ wfrag, const float* __restrict__ bfrag,
                                                float* __restrict__ F, float fscale) {
    __shared__ float wl[RESI ? kTrunkResiLds : kTrunkWFrags * 64];  // this block's branch: 84 A-operand fragments, 21 KB (+ resi: the table of preprocessed pixel sums)
    // wave-uniform on purpose (readfirstlane): task, group and unit indices then live in SGPRs, and every load / store
    // below is "SGPR base + one 32-bit VGPR lane offset" (saddr form) instead of a 64-bit per-lane address: a VMEM
    // instruction with VGPR addresses costs several times more matrix-pipe time (profiles/r02_fc1_variants.txt)
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const int groups = (N + 15) / 16;
    if (b < bS) Trunk<0, RESI, false, false, FAST>::run(XS, groups * 16, b * 4 + w, bS * 4, wfrag, bfrag, F, N, wl, nullptr, nullptr, 0, nullptr, fscale);
    else if (b < bS + bM) Trunk<1, RESI, false, false, FAST>::run(XM, groups * 4, (b - bS) * 4 + w, bM * 4, wfrag, bfrag, F, N, wl, nullptr, nullptr, 0, nullptr, fscale);
    else Trunk<2, RESI, false, false, FAST>::run(XL, groups, (b - bS - bM) * 4 + w, (int)(gridDim.x - bS - bM) * 4, wfrag, bfrag, F, N, wl, nullptr, nullptr, 0, nullptr, fscale);
}

// blocks per 256 of the S / M / L branches: tasks 16 : 4 : 1, weighted by their instruction-mix cost per task (M / L tasks
// carry ~80 more VALU for the 16-bit unpack: ~9700 vs ~9400 clocks) -> 193 : 50 : 13 (195 : 49 : 12 measured 0.6 % slower)
#ifndef TRUNK_SH_S
#define TRUNK_SH_S 193
#define TRUNK_SH_M 50
#define TRUNK_SH_L 13
#endif

// A/B form (experiments build, ETHCNN_TILE_FOLD=1): the CTU-load stage folded into the trunk for big passes as
// well -- S / M waves gather their records straight from the luma frames (Trunk<.., DIRECT>, as the single-launch small pass does), an
// L task (64 KB of pixels) is one block whose four waves gather together and whose wave 0 computes.  No k0_tile_slab, no 6,656 B per
// CTU of slab records written and read back, no side stream.  Measured, not kept: profiles/r04_tile_fold.txt.
__global__ __launch_bounds__(256) void k1_trunk_direct(DirectSrc src, int N, int bS, int bM, const float* __restrict__ wfrag,
                                                       const float* __restrict__ bfrag, float* __restrict__ F) {
    __shared__ float wl[kTrunkWFrags * 64 + 8 * 64 * 4];  // weight fragments + the L gather's exchange area
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const int groups = (N + 15) / 16;
    if (b < bS) Trunk<0, false, true>::run(nullptr, groups * 16, b * 4 + w, bS * 4, wfrag, bfrag, F, N, wl, &src);
    else if (b < bS + bM) Trunk<1, false, true>::run(nullptr, groups * 4, (b - bS) * 4 + w, bM * 4, wfrag, bfrag, F, N, wl, &src);
    else Trunk<2, false, true>::run(nullptr, groups, b - bS - bM, 1 << 30, wfrag, bfrag, F, N, wl, &src);  // one L task per block
}

void launch_trunk_direct(const uint8_t* d_luma, const FrameGeom& g, long ctu0, const Workspace& ws, const DeviceWeights& w, int n, hipStream_t s) {
    const int groups = (n + 15) / 16, tS = groups * 16, tM = groups * 4;
    auto blocks = [](int tasks, int budget) { int b = (tasks + 3) / 4; return b < budget ? b : budget; };
    const int bS = blocks(tS, TRUNK_SH_S * 3), bM = blocks(tM, TRUNK_SH_M * 3);
    DirectSrc src;
    src.luma = d_luma; src.width = g.width; src.height = g.height; src.pitch = g.pitch; src.frame_stride = g.frame_stride;
    src.cw = g.cw; src.nctu = g.nctu; src.ctu0 = ctu0; src.n_total = n;
    hipLaunchKernelGGL(k1_trunk_direct, dim3(bS + bM + groups), dim3(256), 0, s, src, n, bS, bM, w.trunk_w, w.trunk_b, ws.feat);
}

void launch_trunk(const Workspace& ws, const DeviceWeights& w, int n, bool resi, hipStream_t s, int fc1_plan) {
    // tasks per group: 16 S, 4 M, 1 L -- all 240 MFMAs.  768 blocks = 3 per CU (156 VGPRs, 21 KB LDS).
    const int groups = (n + 15) / 16, tS = groups * 16, tM = groups * 4, tL = groups;
    auto blocks = [](int tasks, int budget) { int b = (tasks + 3) / 4; return b < budget ? b : budget; };
    static int per_cu = 0;
    if (!per_cu) { const char* e = dev_env("ETHCNN_TRUNK_BLOCKS_PER_CU"); per_cu = e ? atoi(e) : 3; }  // development knob
    const int bS = blocks(tS, TRUNK_SH_S * per_cu), bM = blocks(tM, TRUNK_SH_M * per_cu), bL = blocks(tL, TRUNK_SH_L * per_cu);
    if (fc1_plan == 2)  // plan 2 (All-Intra passes only): the features leave as fp16 x 2 pieces of the scaled values in ws.featb
        hipLaunchKernelGGL((k1_trunk<false, 2>), dim3(bS + bM + bL), dim3(256), 0, s, ws.xs, ws.xm, ws.xl, n, bS, bM,
                           w.trunk_w, w.trunk_b, reinterpret_cast<float*>(ws.featb), w.fast_scale_a);
    else if (resi)
        hipLaunchKernelGGL(k1_trunk<true>, dim3(bS + bM + bL), dim3(256), 0, s, ws.xs, ws.xm, ws.xl, n, bS, bM,
                           w.trunk_w, w.trunk_b, ws.feat, 1.0f);
    else
        hipLaunchKernelGGL(k1_trunk<false>, dim3(bS + bM + bL), dim3(256), 0, s, ws.xs, ws.xm, ws.xl, n, bS, bM,
                           w.trunk_w, w.trunk_b, ws.feat, 1.0f);
}

}  // namespace ethcnn
