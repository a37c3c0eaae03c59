// launch_rtt.hip -- host round trip of ONE kernel launch + completion on this box: what a single-launch call pays above its
// kernel.  Variants: hipStreamSynchronize with the default / spin / yield / blocking scheduling flags, and completion through
// a flag in page-locked host memory that the kernel's last thread stores with system scope while the host spins on it.
// build: hipcc --offload-arch=gfx950 -O2 launch_rtt.hip -o launch_rtt ; run: ./launch_rtt [kernel_us]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

__global__ void k_wait(unsigned long long ticks, volatile unsigned* flag, unsigned value) {
    unsigned long long t0, t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    do {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    } while (t - t0 < ticks);
    if (flag && threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store((unsigned*)flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    const double kernel_us = argc > 1 ? atof(argv[1]) : 40.0;
    const unsigned mode = argc > 2 ? (unsigned)atoi(argv[2]) : 0;  // 0 default, 1 spin, 2 yield, 3 blocking
    const unsigned flags[] = {hipDeviceScheduleAuto, hipDeviceScheduleSpin, hipDeviceScheduleYield, hipDeviceScheduleBlockingSync};
    if (hipSetDeviceFlags(flags[mode & 3]) != hipSuccess) printf("hipSetDeviceFlags failed\n");
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    volatile unsigned* flag = nullptr;
    hipHostMalloc((void**)&flag, 64, hipHostMallocMapped);
    *flag = 0;
    volatile unsigned* sig = nullptr;
    if (hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory) != hipSuccess) { printf("signal memory failed\n"); return 1; }
    *sig = 0;
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    const unsigned long long ticks = (unsigned long long)(kernel_us * 100.0);
    const int reps = 2000;
    for (int variant = 0; variant < 4; ++variant) {
        std::vector<double> v;
        for (int i = 0; i < reps + 200; ++i) {
            const double t0 = now_us();
            if (variant == 0) {
                hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, s, ticks, (volatile unsigned*)nullptr, 0u);
                hipStreamSynchronize(s);
            } else if (variant == 1) {
                hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, s, ticks, flag, (unsigned)(i + 1));
                while (*flag != (unsigned)(i + 1)) {}
            } else if (variant == 2) {
                hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, s, ticks, (volatile unsigned*)nullptr, 0u);
                if (hipStreamWriteValue32(s, (void*)sig, (unsigned)(i + 1), 0) != hipSuccess) { printf("hipStreamWriteValue32 failed\n"); return 1; }
                while (*sig != (unsigned)(i + 1)) {}
            } else {
                hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, s, ticks, (volatile unsigned*)nullptr, 0u);
                hipEventRecord(ev, s);
                while (hipEventQuery(ev) == hipErrorNotReady) {}
            }
            const double t1 = now_us();
            if (i >= 200) v.push_back(t1 - t0);
        }
        hipStreamSynchronize(s);
        std::sort(v.begin(), v.end());
        printf("sched mode %u  kernel %.0f us  %-34s round trip median %.1f us  p10 %.1f  p90 %.1f  -> overhead %.1f us\n", mode, kernel_us,
               variant == 0 ? "hipStreamSynchronize" : variant == 1 ? "host spins on a pinned-memory flag" : variant == 2 ? "hipStreamWriteValue32 + host spin" : "hipEventRecord + hipEventQuery spin", v[v.size() / 2], v[v.size() / 10],
               v[v.size() * 9 / 10], v[v.size() / 2] - kernel_us);
    }
    return 0;
}
