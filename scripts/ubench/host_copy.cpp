// Host-side microbenchmark for the staging path of ethcnn_predict_luma / ethcnn_predict_yuv_file (S2 / S3 scopes):
// how fast can the host fill a pinned buffer, with what, and how fast does the DMA engine drain it?
//   fill variants : memcpy | AVX2 non-temporal copy | pread from a tmpfs file | NT copy from an mmap of that file
//   destinations  : hipHostMalloc default | write-combined | plain malloc (reference: no pinning)
//   threads       : 4 .. 64
//   H2D           : hipMemcpyAsync from the pinned buffer, one stream and two streams (halves)
// Build: hipcc -O3 -mavx2 -pthread -o host_copy host_copy.cpp      Run on the GPU box.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void nt_copy(void* dst, const void* src, size_t n) {  // dst 32-byte aligned, n multiple of 128
    const __m256i* s = (const __m256i*)src;
    __m256i* d = (__m256i*)dst;
    for (size_t i = 0; i < n / 32; i += 4) {
        const __m256i a = _mm256_loadu_si256(s + i), b = _mm256_loadu_si256(s + i + 1), c = _mm256_loadu_si256(s + i + 2),
                      e = _mm256_loadu_si256(s + i + 3);
        _mm256_stream_si256(d + i, a);
        _mm256_stream_si256(d + i + 1, b);
        _mm256_stream_si256(d + i + 2, c);
        _mm256_stream_si256(d + i + 3, e);
    }
    _mm_sfence();
}

static double run_threads(int nt, size_t total, size_t unit, const std::function<void(size_t off, size_t n)>& fn) {
    std::vector<std::thread> th;
    const size_t units = total / unit;
    const double t0 = now();
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            for (size_t u = t; u < units; u += nt) fn(u * unit, unit);
        });
    for (auto& x : th) x.join();
    return now() - t0;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t total = (size_t)400 << 20, unit = 512 << 10;  // one C3 step of luma; the library's fill unit
    uint8_t *src = (uint8_t*)aligned_alloc(4096, total), *pin = nullptr, *wc = nullptr, *plain = (uint8_t*)aligned_alloc(4096, total), *dev = nullptr;
    memset(src, 7, total);
    memset(plain, 1, total);
    if (hipHostMalloc((void**)&pin, total, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void**)&wc, total, hipHostMallocWriteCombined) != hipSuccess ||
        hipMalloc((void**)&dev, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    memset(pin, 1, total);
    memset(wc, 1, total);
    const char* path = "/dev/shm/host_copy_probe.bin";
    int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600);
    for (size_t o = 0; o < total; o += (80 << 20)) if (write(fd, src + o, 80 << 20) != (80 << 20)) return 2;
    printf("%d hardware threads\n", (int)std::thread::hardware_concurrency());
    struct Dst { const char* name; uint8_t* p; } dsts[] = {{"pinned", pin}, {"pinned-WC", wc}, {"malloc", plain}};
    for (int nt : {4, 8, 16, 32, 64}) {
        for (auto& d : dsts) {
            double best[4] = {1e9, 1e9, 1e9, 1e9};
            for (int rep = 0; rep < 3; ++rep) {
                best[0] = std::min(best[0], run_threads(nt, total, unit, [&](size_t o, size_t n) { memcpy(d.p + o, src + o, n); }));
                best[1] = std::min(best[1], run_threads(nt, total, unit, [&](size_t o, size_t n) { nt_copy(d.p + o, src + o, n); }));
                best[2] = std::min(best[2], run_threads(nt, total, unit, [&](size_t o, size_t n) {
                                       size_t got = 0;
                                       while (got < n) { ssize_t r = pread(fd, d.p + o + got, n - got, (off_t)(o + got)); if (r <= 0) break; got += (size_t)r; }
                                   }));
            }
            printf("threads %2d -> %-9s  memcpy %6.1f GB/s   nt-copy %6.1f GB/s   pread(tmpfs) %6.1f GB/s\n", nt, d.name, total / best[0] / 1e9,
                   total / best[1] / 1e9, total / best[2] / 1e9);
        }
    }
    {   // mmap of the file: population cost, then NT copy out of it
        double t0 = now();
        uint8_t* m = (uint8_t*)mmap(nullptr, total, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
        const double t_map = now() - t0;
        if (m != MAP_FAILED) {
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) best = std::min(best, run_threads(32, total, unit, [&](size_t o, size_t n) { nt_copy(pin + o, m + o, n); }));
            printf("mmap(MAP_POPULATE) of the 400 MB tmpfs file %.2f ms (%.1f GB/s equivalent); nt-copy mmap -> pinned, 32 threads: %.1f GB/s\n", t_map * 1e3,
                   total / t_map / 1e9, total / best / 1e9);
            t0 = now();
            hipError_t e = hipHostRegister(m, total, hipHostRegisterDefault);
            const double t_reg = now() - t0;
            printf("hipHostRegister(mmap of the file): %s, %.2f ms (%.1f GB/s equivalent)\n", hipGetErrorString(e), t_reg * 1e3, total / t_reg / 1e9);
            if (e == hipSuccess) {
                hipStream_t s; hipStreamCreate(&s);
                t0 = now(); hipMemcpyAsync(dev, m, total, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
                printf("  H2D straight from the registered mapping: %.1f GB/s\n", total / (now() - t0) / 1e9);
                hipHostUnregister(m);
            }
            munmap(m, total);
        }
        t0 = now();
        hipError_t e = hipHostRegister(src, total, hipHostRegisterDefault);
        printf("hipHostRegister(malloc'd 400 MB): %s, %.2f ms\n", hipGetErrorString(e), (now() - t0) * 1e3);
        if (e == hipSuccess) hipHostUnregister(src);
    }
    hipStream_t s1, s2;
    hipStreamCreate(&s1);
    hipStreamCreate(&s2);
    for (auto& d : dsts) {
        if (d.p == plain) continue;
        double b1 = 1e9, b2 = 1e9, b3 = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            double t0 = now();
            hipMemcpyAsync(dev, d.p, total, hipMemcpyHostToDevice, s1);
            hipStreamSynchronize(s1);
            b1 = std::min(b1, now() - t0);
            t0 = now();
            hipMemcpyAsync(dev, d.p, total / 2, hipMemcpyHostToDevice, s1);
            hipMemcpyAsync(dev + total / 2, d.p + total / 2, total / 2, hipMemcpyHostToDevice, s2);
            hipStreamSynchronize(s1);
            hipStreamSynchronize(s2);
            b2 = std::min(b2, now() - t0);
            t0 = now();
            for (size_t o = 0; o < total; o += (8 << 20)) hipMemcpyAsync(dev + o, d.p + o, 8 << 20, hipMemcpyHostToDevice, s1);
            hipStreamSynchronize(s1);
            b3 = std::min(b3, now() - t0);
        }
        printf("H2D from %-9s: one 400 MB copy %.1f GB/s   two streams %.1f GB/s   50 x 8 MB on one stream %.1f GB/s\n", d.name, total / b1 / 1e9,
               total / b2 / 1e9, total / b3 / 1e9);
    }
    {   // fill and H2D at the same time (what the pipeline does): 32 threads nt-copy into one half while the other half is DMAed
        double t0 = now();
        for (int it = 0; it < 4; ++it) {
            uint8_t* a = pin + (it & 1) * (total / 2);
            hipMemcpyAsync(dev, pin + ((it + 1) & 1) * (total / 2), total / 2, hipMemcpyHostToDevice, s1);
            run_threads(32, total / 2, unit, [&](size_t o, size_t n) { nt_copy(a + o, src + o, n); });
            hipStreamSynchronize(s1);
        }
        printf("concurrent: nt-copy (32 threads) of one half + H2D of the other half: %.1f GB/s through the pipeline\n", 4.0 * (total / 2) / (now() - t0) / 1e9);
        t0 = now();
        for (int it = 0; it < 4; ++it) {
            uint8_t* a = pin + (it & 1) * (total / 2);
            hipMemcpyAsync(dev, pin + ((it + 1) & 1) * (total / 2), total / 2, hipMemcpyHostToDevice, s1);
            run_threads(32, total / 2, unit, [&](size_t o, size_t n) { memcpy(a + o, src + o, n); });
            hipStreamSynchronize(s1);
        }
        printf("concurrent: memcpy  (32 threads) of one half + H2D of the other half: %.1f GB/s through the pipeline\n", 4.0 * (total / 2) / (now() - t0) / 1e9);
    }
    close(fd);
    unlink(path);
    return 0;
}
