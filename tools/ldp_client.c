/* ldp_client.c -- the ENCODER's side of the Low-Delay-P file handshake, exactly as the reference's HM does it
 * (/root/reference/HM-16.5_Test_LDP/source/Lib/TLibEncoder/TEncGOP.cpp:1466-1506), without the encoder around it:
 *
 *     per inter picture:  [resi.yuv written by the pre-encode]  remove("pred_end.sig")
 *                         command.dat = "<POC> <w> <h> <qp> [end]"        (fopen "w+", fprintf, fclose)
 *                         fopen("pred_start.sig", "w+"); fclose
 *                         while ((f = fopen("pred_end.sig", "r")) == NULL) ;     busy wait
 *                         fclose + remove("pred_end.sig") until it succeeds
 *                         fread cu_depth.dat (nctu x 21 float32)
 *
 * It measures what the encoder stands and waits for -- HM's own "Predicting Time" line has a 1 ms resolution (clock() /
 * CLOCKS_PER_SEC printed with %.3f) and was the only encoder-side number so far -- against any daemon that serves the
 * protocol in the working directory (the Python daemon resi_to_cu_depth_LDP.py, the native tools/resi_to_cu_depth_ldp).
 *
 *   ldp_client <workdir> <width> <height> <qp> <frames> [--seed N] [--gap-us N] [--digest FILE] [--keep-resi] [--slow-us T]
 *
 * --slow-us T: every handshake longer than T microseconds is printed with its phases and CLOCK_MONOTONIC stamps ("slow POC n: ...",
 * on stderr) so that it can be laid beside the daemon's own record of the same frame (resi_to_cu_depth_ldp --trace-slow).
 *
 * Frames are seeded synthetic residual pictures (8-bit, Laplace-like around 128; a new one per frame unless --keep-resi).
 * Output: one line "ldp_client WxH frames N: handshake p50 .. p90 .. p99 .. max .. us (command.dat -> cu_depth.dat read);
 * incl. resi.yuv write p50 .. us".  --digest FILE appends one line per frame "<poc> <fnv1a64 of cu_depth.dat>" so that two
 * daemons can be compared frame by frame.  --gap-us: idle time between frames (an encoder spends tens of ms encoding). */
#define _POSIX_C_SOURCE 200809L
#include <errno.h>
#include <stdint.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}

static uint64_t rng_state;
static uint32_t rng(void) { /* splitmix64 */
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}

static void make_residual(uint8_t* luma, size_t n) { /* sum of two uniforms around 128, +-24: a residual-like histogram */
    for (size_t i = 0; i < n; ++i) {
        const uint32_t r = rng();
        const int v = 128 + (int)(r & 31) - 16 + (int)((r >> 8) & 15) - 8;
        luma[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

static int cmp_double(const void* a, const void* b) {
    const double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

int main(int argc, char** argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: ldp_client <workdir> <width> <height> <qp> <frames> [--seed N] [--gap-us N] [--digest FILE] [--keep-resi] [--slow-us US] [--poll-us US]\n");
        return 2;
    }
    const char* dir = argv[1];
    const int w = atoi(argv[2]), h = atoi(argv[3]), qp = atoi(argv[4]), frames = atoi(argv[5]);
    long gap_us = 0;
    double slow_us = 0.0;
    const char* digest = NULL;
    int keep_resi = 0;
    long poll_us = -1; /* < 0: HM's own wait, a tight loop of failing fopen("pred_end.sig") calls (TEncGOP.cpp:1483); >= 0 (DIAGNOSTIC, not
                          what the unchanged encoder does): sleep that long between two attempts -- profiles/r06_ldp_tail.txt uses it to show
                          what the busy loop itself does to file operations in a tmpfs directory */
    rng_state = 12345;
    for (int i = 6; i < argc; ++i) {
        if (!strcmp(argv[i], "--seed") && i + 1 < argc) rng_state = strtoull(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "--gap-us") && i + 1 < argc) gap_us = atol(argv[++i]);
        else if (!strcmp(argv[i], "--digest") && i + 1 < argc) digest = argv[++i];
        else if (!strcmp(argv[i], "--keep-resi")) keep_resi = 1;
        else if (!strcmp(argv[i], "--slow-us") && i + 1 < argc) slow_us = atof(argv[++i]);
        else if (!strcmp(argv[i], "--poll-us") && i + 1 < argc) poll_us = atol(argv[++i]);
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (w <= 0 || h <= 0 || frames <= 0) { fprintf(stderr, "bad geometry / frame count\n"); return 2; }
    if (chdir(dir) != 0) { fprintf(stderr, "chdir %s: %s\n", dir, strerror(errno)); return 1; }
    const size_t luma_bytes = (size_t)w * h, frame_bytes = luma_bytes * 3 / 2;
    const size_t nctu = (size_t)((w + 63) / 64) * ((h + 63) / 64);
    uint8_t* yuv = (uint8_t*)malloc(frame_bytes);
    float* depth = (float*)malloc(nctu * 21 * sizeof(float));
    double* t_hand = (double*)malloc(sizeof(double) * frames);
    double* t_full = (double*)malloc(sizeof(double) * frames);
    if (!yuv || !depth || !t_hand || !t_full) { fprintf(stderr, "out of memory\n"); return 1; }
    memset(yuv + luma_bytes, 128, frame_bytes - luma_bytes);
    make_residual(yuv, luma_bytes);
    FILE* fd = digest ? fopen(digest, "w") : NULL;
    if (digest && !fd) { fprintf(stderr, "cannot open %s\n", digest); return 1; }

    for (int f = 0; f < frames; ++f) {
        const int poc = f + 1; /* POC 0 is the intra picture: never predicted (TEncGOP.cpp:1455) */
        if (!keep_resi && f > 0) make_residual(yuv, luma_bytes);
        const double t0 = now_us();
        { /* the pre-encode's product (TEncSlice: resi.yuv, one 4:2:0 picture) */
            FILE* fy = fopen("resi.yuv", "wb");
            if (!fy || fwrite(yuv, 1, frame_bytes, fy) != frame_bytes) { fprintf(stderr, "cannot write resi.yuv\n"); return 1; }
            fclose(fy);
        }
        const double t1 = now_us();
        remove("pred_end.sig");
        FILE* fp = fopen("command.dat", "w+");
        if (!fp) { fprintf(stderr, "cannot write command.dat\n"); return 1; }
        fprintf(fp, "%d %d %d %d [end]", poc, w, h, qp);
        fclose(fp);
        const double t_cmd = now_us();
        FILE* fs = fopen("pred_start.sig", "w+");
        if (!fs) { fprintf(stderr, "cannot create pred_start.sig\n"); return 1; }
        fclose(fs);
        FILE* fe;
        const double t_wait0 = now_us();
        unsigned spins = 0;
        while ((fe = fopen("pred_end.sig", "r")) == NULL) {
            if ((++spins & 0xfff) == 0 && now_us() - t_wait0 > 30e6) { fprintf(stderr, "no answer from the daemon for 30 s (POC %d)\n", poc); return 3; }
            if (poll_us >= 0) {
                struct timespec ts = {0, poll_us * 1000};
                if (poll_us > 0) nanosleep(&ts, NULL); else sched_yield();
            }
        }
        const double t_seen = now_us();
        int rr = -1;
        while (rr < 0) {
            if (fe) { fclose(fe); fe = NULL; }
            rr = remove("pred_end.sig");
        }
        FILE* fc = fopen("cu_depth.dat", "rb");
        if (!fc) { fprintf(stderr, "cu_depth.dat missing after the ending signal (POC %d)\n", poc); return 3; }
        const size_t got = fread(depth, sizeof(float), nctu * 21, fc);
        fclose(fc);
        const double t2 = now_us();
        if (got != nctu * 21) { fprintf(stderr, "cu_depth.dat holds %zu floats, expected %zu (POC %d)\n", got, nctu * 21, poc); return 3; }
        t_hand[f] = t2 - t1;
        t_full[f] = t2 - t0;
        if (slow_us > 0.0 && f >= 5 && t2 - t1 > slow_us)
            fprintf(stderr, "slow POC %d: handshake %.0f us = remove + command.dat %.0f | create pred_start.sig %.0f | wait for pred_end.sig %.0f | "
                            "remove it + read cu_depth.dat %.0f ; monotonic us: pred_start.sig created %.0f, pred_end.sig seen %.0f\n",
                    poc, t2 - t1, t_cmd - t1, t_wait0 - t_cmd, t_seen - t_wait0, t2 - t_seen, t_wait0, t_seen);
        if (fd) {
            uint64_t hsh = 1469598103934665603ull;
            const uint8_t* b = (const uint8_t*)depth;
            for (size_t i = 0; i < nctu * 21 * sizeof(float); ++i) hsh = (hsh ^ b[i]) * 1099511628211ull;
            fprintf(fd, "%d %016llx\n", poc, (unsigned long long)hsh);
        }
        if (gap_us > 0) {
            struct timespec ts = {gap_us / 1000000, (gap_us % 1000000) * 1000};
            nanosleep(&ts, NULL);
        }
    }
    if (fd) fclose(fd);
    /* the first frames load the LSTM model and size the buffers: they are reported apart, not inside the percentiles */
    const int skip = frames > 20 ? 5 : 0, n = frames - skip;
    const double first = t_hand[0];
    qsort(t_hand + skip, (size_t)n, sizeof(double), cmp_double);
    qsort(t_full + skip, (size_t)n, sizeof(double), cmp_double);
    const double* a = t_hand + skip;
    printf("ldp_client %dx%d qp %d, %d frames (gap %ld us): handshake p50 %.1f  p90 %.1f  p99 %.1f  max %.1f us (command.dat -> cu_depth.dat read); "
           "incl. the resi.yuv write p50 %.1f us; first frame %.0f us\n",
           w, h, qp, frames, gap_us, a[n / 2], a[(int)(n * 0.9)], a[(int)(n * 0.99)], a[n - 1], (t_full + skip)[n / 2], first);
    free(yuv); free(depth); free(t_hand); free(t_full);
    return 0;
}
