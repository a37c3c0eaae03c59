// row_arrival_probe.hip -- can a kernel that is ALREADY RUNNING consume a picture band by band while the copy engine is still
// delivering it?  (one 4K picture host -> host is 151 us of PCIe + ~100 us of kernels, serial.)
//   stream B: for each of NB bands: hipMemcpyAsync(band, pinned -> HBM) ; hipStreamWriteValue32(flag, seq * 64 + band + 1)
//   stream A: ONE kernel, launched BEFORE the copies: every block waits for the flag of its band (system-scope loads, one
//             thread), then reads its slice of the band and checks it word by word against the pattern of this iteration
// Questions: (1) stale data?  The consumer's L2 / L1 may still hold the PREVIOUS iteration's lines of the same buffer, and no
// kernel boundary stands between the DMA and the loads.  Tried with the staging buffer in ordinary hipMalloc memory (plain
// loads, and sc1 = agent-scope loads) and in hipDeviceMallocUncached memory.  (2) the time of copy || kernel against copy ->
// kernel.  (3) what 8-byte gather loads cost from uncached memory.
// build: hipcc --offload-arch=gfx950 -O2 row_arrival_probe.hip -o row_arrival_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int NB = 8;  // bands

__device__ __forceinline__ unsigned sys_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// mode 0: plain loads, 1: agent-scope (sc1) loads
template <int MODE>
__global__ __launch_bounds__(256) void k_consume(const uint2* buf, size_t words_per_band, const unsigned* flag, unsigned base, unsigned salt,
                                                 unsigned* bad, unsigned long long* waited) {
    const int band = blockIdx.x % NB, slot = blockIdx.x / NB, slots = gridDim.x / NB;
    __shared__ unsigned long long t_wait;
    if (threadIdx.x == 0) {
        unsigned long long t0, t1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
        while ((int)(sys_load(flag) - (base + band + 1)) < 0) __builtin_amdgcn_s_sleep(8);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
        t_wait = t1 - t0;
    }
    __syncthreads();
    const uint2* b = buf + (size_t)band * words_per_band;
    unsigned nbad = 0;
    for (size_t i = (size_t)slot * 256 + threadIdx.x; i < words_per_band; i += (size_t)slots * 256) {
        uint2 v;
        if (MODE == 0) v = b[i];
        else {
            unsigned long long q = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(b + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v.x = (unsigned)q; v.y = (unsigned)(q >> 32);
        }
        const unsigned w = (unsigned)(band * words_per_band + i);
        if (v.x != (w * 2654435761u ^ salt) || v.y != ((w + 77u) * 40503u ^ salt)) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
    if (threadIdx.x == 0 && slot == 0) waited[band] = t_wait;
}

// ---- part 2: the kernel PULLS the picture itself.  One block = one group of 16 CTUs of a CTU row: 64 rows x 1 KiB, read from
// page-locked host memory with 16-byte loads (all 16 of a thread in flight at once), stored to HBM.  No copy engine, no
// stream dependency, no stale lines (the stores and the consumers' loads are agent-scope traffic of one launch).
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_pull(const v4u* __restrict__ src, v4u* dst, int width16, int cw16, unsigned long long* stamps) {
    // block b: CTU row b / cw16, 16-CTU column b % cw16; thread t: row (t >> 6) + 4 j, 16-byte piece t & 63
    const int r0 = (blockIdx.x / cw16) * 64, c0 = (blockIdx.x % cw16) * 64;
    v4u v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = __builtin_nontemporal_load(src + (size_t)(r0 + (threadIdx.x >> 6) + 4 * j) * width16 + c0 + (threadIdx.x & 63));
#pragma unroll
    for (int j = 0; j < 16; ++j) dst[(size_t)(r0 + (threadIdx.x >> 6) + 4 * j) * width16 + c0 + (threadIdx.x & 63)] = v[j];
    if (stamps && threadIdx.x == 0) {
        unsigned long long t;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        stamps[blockIdx.x] = t;
    }
}

static int pull_test(int w, int h) {
    const size_t bytes = (size_t)w * h;
    v4u *hsrc = nullptr, *d = nullptr;
    unsigned long long* d_st = nullptr;
    CK(hipHostMalloc((void**)&hsrc, bytes, hipHostMallocDefault));
    CK(hipMalloc((void**)&d, bytes));
    const int cw16 = w / 1024, rows = h / 64, blocks = cw16 * rows;  // (whole 16-CTU columns and CTU rows only: a probe)
    CK(hipMalloc((void**)&d_st, 8 * blocks));
    std::memset(hsrc, 0x5a, bytes);
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double t_pull = 0, t_dma = 0;
    const int iters = 40;
    std::vector<unsigned long long> st(blocks);
    for (int it = 0; it < iters; ++it) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_pull, dim3(blocks), dim3(256), 0, s, hsrc, d, w / 16, cw16, d_st);
        CK(hipStreamSynchronize(s));
        auto t1 = std::chrono::steady_clock::now();
        CK(hipMemcpyAsync(d, hsrc, (size_t)rows * 64 * w, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        auto t2 = std::chrono::steady_clock::now();
        if (it >= 8) { t_pull += std::chrono::duration<double, std::micro>(t1 - t0).count(); t_dma += std::chrono::duration<double, std::micro>(t2 - t1).count(); }
    }
    CK(hipMemcpy(st.data(), d_st, 8 * blocks, hipMemcpyDeviceToHost));
    unsigned long long lo = ~0ull, hi = 0;
    for (auto v : st) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    const double moved = (double)blocks * 65536;
    printf("pull %dx%d: %d blocks x 64 KiB = %.2f MB  kernel pulls from page-locked memory: %.1f us host to host (%.1f GB/s)   hipMemcpyAsync of the same rows + sync: %.1f us (%.1f GB/s)\n"
           "    first block had its 64 KiB after .. last block: spread %.1f us; arrival of blocks 0, 1/4, 1/2, 3/4, last (us after the first): %.1f %.1f %.1f %.1f %.1f\n",
           w, h, blocks, moved / 1e6, t_pull / (iters - 8), moved / (t_pull / (iters - 8)) / 1e3, t_dma / (iters - 8), (double)rows * 64 * w / (t_dma / (iters - 8)) / 1e3,
           (hi - lo) / 100.0, (st[0] - lo) / 100.0, (st[blocks / 4] - lo) / 100.0, (st[blocks / 2] - lo) / 100.0, (st[3 * blocks / 4] - lo) / 100.0, (st[blocks - 1] - lo) / 100.0);
    return 0;
}

int main() {
    if (pull_test(1920, 1080) || pull_test(3840, 2160)) return 1;  // (1920 = 1 full 16-CTU column + a ragged one: the probe moves the full one only)
    if (pull_test(2048, 1088) || pull_test(4096, 2176)) return 1;
    const size_t bytes = 3840 * 2160;  // one 4K luma plane
    const size_t words = bytes / 8, wpb = words / NB;
    uint2* h = nullptr;
    CK(hipHostMalloc((void**)&h, bytes, hipHostMallocDefault));
    unsigned *d_bad, *d_flag_u, *d_flag_c;
    unsigned long long* d_wait;
    CK(hipMalloc((void**)&d_bad, 4));
    CK(hipMalloc((void**)&d_wait, 8 * NB));
    CK(hipMalloc((void**)&d_flag_c, 64));
    CK(hipExtMallocWithFlags((void**)&d_flag_u, 64, hipDeviceMallocUncached));
    CK(hipMemset(d_flag_c, 0, 64));
    CK(hipMemset(d_flag_u, 0, 64));
    uint2 *buf_c = nullptr, *buf_u = nullptr;
    CK(hipMalloc((void**)&buf_c, bytes));
    CK(hipExtMallocWithFlags((void**)&buf_u, bytes, hipDeviceMallocUncached));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    unsigned seq = 0;
    for (int cfg = 0; cfg < 4; ++cfg) {
        uint2* buf = cfg < 2 ? buf_c : buf_u;
        unsigned* flag = d_flag_u;  // the flag itself always lives in uncached memory
        const int mode = cfg & 1;
        unsigned total_bad = 0;
        double t_overlap = 0, t_serial = 0;
        unsigned long long waited[NB] = {0};
        const int iters = 60;
        for (int it = 0; it < iters; ++it) {
            const unsigned salt = 0x9e3779b9u * (unsigned)(it + 1 + 1000 * cfg);
            for (size_t w = 0; w < words; ++w) { h[w].x = ((unsigned)w * 2654435761u) ^ salt; h[w].y = (((unsigned)w + 77u) * 40503u) ^ salt; }
            CK(hipMemset(d_bad, 0, 4));
            CK(hipDeviceSynchronize());
            // ---- overlapped: kernel first, then the banded copy with a flag write behind every band
            ++seq;
            const unsigned base = seq * 64u;
            auto t0 = std::chrono::steady_clock::now();
            if (mode == 0) hipLaunchKernelGGL(k_consume<0>, dim3(NB * 32), dim3(256), 0, sa, buf, wpb, flag, base, salt, d_bad, d_wait);
            else hipLaunchKernelGGL(k_consume<1>, dim3(NB * 32), dim3(256), 0, sa, buf, wpb, flag, base, salt, d_bad, d_wait);
            for (int b = 0; b < NB; ++b) {
                CK(hipMemcpyAsync(buf + (size_t)b * wpb, h + (size_t)b * wpb, wpb * 8, hipMemcpyHostToDevice, sb));
                CK(hipStreamWriteValue32(sb, flag, base + b + 1, 0));
            }
            CK(hipStreamSynchronize(sa));
            auto t1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(sb));
            unsigned nb = 0;
            CK(hipMemcpy(&nb, d_bad, 4, hipMemcpyDeviceToHost));
            total_bad += nb;
            if (it == iters - 1) CK(hipMemcpy(waited, d_wait, sizeof waited, hipMemcpyDeviceToHost));
            // ---- serial reference: whole copy, then the kernel (flags already satisfied)
            CK(hipDeviceSynchronize());
            auto t2 = std::chrono::steady_clock::now();
            CK(hipMemcpyAsync(buf, h, bytes, hipMemcpyHostToDevice, sa));
            if (mode == 0) hipLaunchKernelGGL(k_consume<0>, dim3(NB * 32), dim3(256), 0, sa, buf, wpb, flag, base, salt, d_bad, d_wait);
            else hipLaunchKernelGGL(k_consume<1>, dim3(NB * 32), dim3(256), 0, sa, buf, wpb, flag, base, salt, d_bad, d_wait);
            CK(hipStreamSynchronize(sa));
            auto t3 = std::chrono::steady_clock::now();
            if (it >= 10) {
                t_overlap += std::chrono::duration<double, std::micro>(t1 - t0).count();
                t_serial += std::chrono::duration<double, std::micro>(t3 - t2).count();
            }
        }
        printf("%-22s %-18s: stale / wrong words over %d pictures: %u   kernel launched before the banded copy: %.1f us per picture; copy then kernel: %.1f us;  "
               "last picture, flag seen at (us after the block started): ",
               cfg < 2 ? "hipMalloc buffer" : "uncached buffer", mode == 0 ? "plain loads" : "agent-scope loads", iters, total_bad, t_overlap / (iters - 10),
               t_serial / (iters - 10));
        for (int b = 0; b < NB; ++b) printf("%.0f ", (double)waited[b] / 100.0);
        printf("\n");
    }
    return 0;
}
