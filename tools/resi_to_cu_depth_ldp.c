/*
 * resi_to_cu_depth_ldp -- native (C, no Python) form of the Low-Delay-P predictor daemon: same files, same handshake as the
 * reference's /root/reference/HM-16.5_Test_LDP/bin/resi_to_cu_depth_LDP.py:148-190, so the unchanged HM-LDP encoder
 * (TEncGOP.cpp:1463-1503) drives it; same rules as this repository's Python daemon
 * (hevc-complexity-reduction_amd/resi_to_cu_depth_LDP.py), byte-identical cu_depth.dat / state.dat per frame:
 *
 *     HM:     pre-encode -> resi.yuv ; command.dat = "<POC> <w> <h> <qp> [end]" ; create pred_start.sig
 *     daemon: read command.dat, remove pred_start.sig, (re)load the LSTM model when the QP band changed, predict
 *             (ethcnn_ldp_step: resi_cnn + one ETH-LSTM step + heads + gates, recurrent state resident in HBM),
 *             write cu_depth.dat, create pred_end.sig, THEN refresh state.dat (temp + rename) and its sidecar
 *     HM:     spins on pred_end.sig, removes it, freads cu_depth.dat
 *
 * Why it exists: measured from the encoder's side (tools/ldp_client.c) the Python daemon answers a
 * 1920x1080 frame in ~3x the 106 us the library call takes: interpreter time per frame, a 200 us poll sleep, numpy / ctypes
 * marshalling.  Here the loop is a dozen system calls around ethcnn_ldp_step:
 *   * wake-up by inotify on the working directory (IN_CREATE / IN_MOVED_TO / IN_CLOSE_WRITE), confirmed by access() -- no
 *     poll interval, no busy core while the encoder encodes; a short spin right after a frame catches back-to-back frames;
 *   * resi.yuv is read by eight threads straight into a page-locked buffer the kernels use in place (ethcnn_host_alloc) WHILE the
 *     frame's kernels are already queued and take the plane CTU row by CTU row as the threads report it (ethcnn_ldp_step_begin /
 *     ethcnn_rows_ready / ethcnn_ldp_step_end, see read_luma_start); the probabilities land in a second page-locked buffer and
 *     go to cu_depth.dat with one write().  (A library entry that read the file in 16 bands on its worker pool and DMA'd each band
 *     as it arrived -- kernels on device-resident pixels -- was built first and measured 276 us per 1080p call against 64 + 114
 *     for read-then-predict: sixteen small async copies from a pool cost more than the overlap wins.  Kernels that WAIT for the
 *     rows do not have that cost.)
 *   * state handling as in the Python daemon: resident in HBM while this daemon produced the previous frame of the same
 *     geometry and the state.dat it wrote is still the one on disk (inode / size / mtime), else read from state.dat; the
 *     sidecar state.dat.idx says "pending <i> <w> <h>" from before the ending signal until state.dat holds that frame.
 *     A stale sidecar ("pending" left by a daemon that died) makes the frame start from the file as it is after a warning
 *     ONLY when --accept-stale is given; otherwise the daemon reports the error, answers nothing and exits non-zero
 *     (the Python daemon does the same: a silently wrong recurrence is worse than a stopped encode).
 *
 *   resi_to_cu_depth_ldp [--max-frames N] [--idle-timeout SECONDS] [--quiet] [--accept-stale] [--trace] [--trace-slow US] [--spin] [--no-stream]   (cwd = HM-LDP's bin/)
 *
 * Environment as the Python daemon: ETHCNN_SYNTHETIC_SEED / ETHCNN_HEAD_GAIN (seeded weights when a trained blob is absent:
 * model_LDP_2000000_qp22~37.dat.data is not in the reference repository), ETHCNN_DEVICE.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/inotify.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "ethcnn.h"

#define MODEL_CNN_FILE "model_LDP_2000000_qp22~37.dat" /* resi_to_cu_depth_LDP.py:159 */
#define NVEC 448

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + (double)ts.tv_nsec * 1e-9;
}

static int file_exists(const char* path) { return access(path, F_OK) == 0; }

typedef struct { long ino, size, mt_s, mt_ns; int ok; } file_sig;
static file_sig sig_of(const char* path) {
    struct stat st;
    file_sig s = {0, 0, 0, 0, 0};
    if (stat(path, &st) == 0) {
        s.ino = (long)st.st_ino; s.size = (long)st.st_size; s.mt_s = (long)st.st_mtim.tv_sec; s.mt_ns = (long)st.st_mtim.tv_nsec; s.ok = 1;
    }
    return s;
}
static int sig_eq(file_sig a, file_sig b) { return a.ok && b.ok && a.ino == b.ino && a.size == b.size && a.mt_s == b.mt_s && a.mt_ns == b.mt_ns; }

static int write_atomic(const char* path, const void* data, size_t bytes) {
    char tmp[96];
    snprintf(tmp, sizeof tmp, "%s.tmp.%ld", path, (long)getpid());
    const int fd = open(tmp, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return -1;
    const char* p = (const char*)data;
    size_t done = 0;
    while (done < bytes) {
        const ssize_t r = write(fd, p + done, bytes - done);
        if (r <= 0) { close(fd); unlink(tmp); return -1; }
        done += (size_t)r;
    }
    if (close(fd) != 0 || rename(tmp, path) != 0) { unlink(tmp); return -1; }
    return 0;
}

static int read_exact(const char* path, void* dst, size_t bytes) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    char* p = (char*)dst;
    size_t got = 0;
    while (got < bytes) {
        const ssize_t r = read(fd, p + got, bytes - got);
        if (r <= 0) break;
        got += (size_t)r;
    }
    close(fd);
    return got == bytes ? 0 : -1;
}

/* resi.yuv's luma -> the page-locked buffer, STREAMED into the running prediction.  One read() of a 1920x1080 plane is a 2 MB
 * single-threaded copy out of the page cache: ~180 us of the ~320 us the encoder waited in round 3's first daemon; four threads (three
 * persistent helpers + the caller) brought it to ~50 us -- followed by the 115 us of ethcnn_ldp_step.  Now the frame's kernels are queued
 * FIRST (ethcnn_ldp_step_begin) and the threads copy the plane one CTU row (64 luma rows) at a time, drawing row numbers from a shared
 * counter and reporting each (ethcnn_rows_ready): the tile stage pulls a row over PCIe as soon as it is there, so the transfer and
 * the launch overheads run under the read: 1920x1080, handshake p50 249 -> 214 us with three helpers, and with seven the copy itself
 * drops under the 38 us the plane needs on the bus (profiles/r04_ldp_handshake.txt).  The helpers sleep on a condition variable between
 * frames.  Pictures under 512 KiB are read by the caller alone, then predicted (nothing to hide a launch under: 416x240 is 100 KB). */
#define NHELP 7
static struct {
    pthread_mutex_t mu;
    pthread_cond_t go, done;
    pthread_t th[NHELP];
    int started, gen, pending, fd, failed, quit;
    char* dst;
    int w, h, next; /* next: CTU row to copy (atomic) */
    ethcnn_ctx* ctx;
} g_rd = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, 0, 0, -1, 0, 0, NULL, 0, 0, 0, NULL};

static int pread_exact(int fd, char* dst, size_t bytes, off_t off) {
    size_t got = 0;
    while (got < bytes) {
        const ssize_t r = pread(fd, dst + got, bytes - got, off + (off_t)got);
        if (r <= 0) return -1;
        got += (size_t)r;
    }
    return 0;
}
/* copy CTU rows until none is left; every row is reported even when its read failed (the kernels waiting for it must drain) */
static int rd_rows(int fd, char* dst, int w, int h, ethcnn_ctx* ctx) {
    const int nrows = (h + 63) / 64;
    int bad = 0;
    for (;;) {
        const int cy = __atomic_fetch_add(&g_rd.next, 1, __ATOMIC_RELAXED);
        if (cy >= nrows) break;
        const size_t off = (size_t)cy * 64 * (size_t)w;
        const int rows = (cy * 64 + 64 <= h) ? 64 : h - cy * 64;
        if (pread_exact(fd, dst + off, (size_t)rows * (size_t)w, (off_t)off) != 0) bad = 1;
        if (ctx) ethcnn_rows_ready(ctx, cy, cy + 1);
    }
    return bad;
}
static void* rd_helper(void* arg) {
    (void)arg;
    int seen = 0;
    pthread_mutex_lock(&g_rd.mu);
    for (;;) {
        while (g_rd.gen == seen && !g_rd.quit) pthread_cond_wait(&g_rd.go, &g_rd.mu);
        if (g_rd.quit) break;
        seen = g_rd.gen;
        const int fd = g_rd.fd, w = g_rd.w, h = g_rd.h;
        char* dst = g_rd.dst;
        ethcnn_ctx* ctx = g_rd.ctx;
        pthread_mutex_unlock(&g_rd.mu);
        const int bad = rd_rows(fd, dst, w, h, ctx);
        pthread_mutex_lock(&g_rd.mu);
        if (bad) g_rd.failed = 1;
        if (--g_rd.pending == 0) pthread_cond_signal(&g_rd.done);
    }
    pthread_mutex_unlock(&g_rd.mu);
    return NULL;
}
/* start: open the file and (pictures of >= 512 KiB) wake the helpers; finish: the caller copies rows too, then waits for them.
 * ctx != NULL: every CTU row is reported to the prediction begun on it. */
static int read_luma_start(const char* path, void* dst, int w, int h, ethcnn_ctx* ctx) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    pthread_mutex_lock(&g_rd.mu);
    if (!g_rd.started && (size_t)w * (size_t)h >= (512u << 10)) {
        g_rd.started = 1;
        for (int i = 0; i < NHELP; ++i)
            if (pthread_create(&g_rd.th[i], NULL, rd_helper, NULL) != 0) { g_rd.started = -1; break; }
        /* (a partial set of helpers would deadlock the pending count: with -1 none is ever woken and the caller reads alone) */
    }
    g_rd.fd = fd; g_rd.dst = (char*)dst; g_rd.w = w; g_rd.h = h; g_rd.ctx = ctx; g_rd.failed = 0;
    __atomic_store_n(&g_rd.next, 0, __ATOMIC_RELAXED);
    g_rd.pending = 0;
    if (g_rd.started > 0 && (size_t)w * (size_t)h >= (512u << 10)) {
        g_rd.pending = NHELP;
        ++g_rd.gen;
        pthread_cond_broadcast(&g_rd.go);
    }
    pthread_mutex_unlock(&g_rd.mu);
    return 0;
}
static int read_luma_finish(void) {
    const int bad = rd_rows(g_rd.fd, g_rd.dst, g_rd.w, g_rd.h, g_rd.ctx);
    pthread_mutex_lock(&g_rd.mu);
    while (g_rd.pending > 0) pthread_cond_wait(&g_rd.done, &g_rd.mu);
    const int rc = (bad || g_rd.failed) ? -1 : 0;
    pthread_mutex_unlock(&g_rd.mu);
    close(g_rd.fd);
    g_rd.fd = -1;
    return rc;
}

/* "<i_frame> <w> <h> <qp> [end]" -> 0, or -1 while the line is incomplete (resi_to_cu_depth_LDP.py:56-72) */
static int get_command(int* i_frame, int* w, int* h, int* qp) {
    char buf[128], tail[16];
    FILE* f = fopen("command.dat", "r");
    if (!f) return -1;
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    tail[0] = 0;
    if (sscanf(buf, "%d %d %d %d %15s", i_frame, w, h, qp, tail) != 5 || strcmp(tail, "[end]") != 0) return -1;
    return 0;
}

static int load_cnn(ethcnn_ctx* ctx) {
    const char* seed = getenv("ETHCNN_SYNTHETIC_SEED");
    const char* gain = getenv("ETHCNN_HEAD_GAIN");
    if (file_exists(MODEL_CNN_FILE ".data-00000-of-00001") || !seed) return ethcnn_load_checkpoint(ctx, MODEL_CNN_FILE);
    return ethcnn_load_synthetic(ctx, (uint64_t)strtoull(seed, NULL, 10), gain ? atof(gain) : 1.0);
}
static int load_lstm(ethcnn_ctx* ctx, int qp, char* name, size_t cap) {
    const char* seed = getenv("ETHCNN_SYNTHETIC_SEED");
    const char* gain = getenv("ETHCNN_HEAD_GAIN");
    char data[128];
    if (ethcnn_lstm_model_name_for_qp(qp, name, cap) != ETHCNN_OK) return ETHCNN_ERR_ARG;
    snprintf(data, sizeof data, "%s.data-00000-of-00001", name);
    if (file_exists(data) || !seed) return ethcnn_load_lstm_checkpoint(ctx, name);
    snprintf(name, cap, "synthetic(seed=%s)", seed);
    return ethcnn_load_lstm_synthetic(ctx, (uint64_t)strtoull(seed, NULL, 10), gain ? atof(gain) : 1.0);
}

/* The sidecar is one short record, space-padded to a fixed 48 bytes and rewritten in place through a descriptor that stays open:
 * one pwrite instead of open / write / close / rename (~40 us on a disk file system, and it sits in front of the ending signal).
 * Readers split on white space (this file and the Python daemon alike). */
static int g_side_fd = -1;
static int sidecar(const char* text) {
    char rec[48];
    const size_t n = strlen(text);
    if (n >= sizeof rec) return -1;
    memset(rec, ' ', sizeof rec);
    memcpy(rec, text, n);
    rec[sizeof rec - 1] = '\n';
    if (g_side_fd < 0) g_side_fd = open("state.dat.idx", O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (g_side_fd < 0) return -1;
    return pwrite(g_side_fd, rec, sizeof rec, 0) == (ssize_t)sizeof rec ? 0 : -1;
}
/* cu_depth.dat: HM opens it only after it has seen pred_end.sig (TEncGOP.cpp:1487-1497), so it is written in place like the
 * reference's daemon does (resi_to_cu_depth_LDP.py:139-141) -- through a shared mapping that stays in place while the size stays the
 * same: one memcpy per frame into the file's own page-cache pages (what HM's fread then reads; a pwrite of the 43 KB of a 1920x1080
 * frame cost ~10 us in front of the ending signal).  The file is re-created when the geometry changes, or when somebody removed it;
 * where it cannot be mapped the daemon writes it with pwrite. */
static int g_depth_fd = -1;
static size_t g_depth_bytes = 0;
static void* g_depth_map = NULL;
/* (the checks, the open and the mapping: done while the GPU is still working on the frame) */
static int prepare_cu_depth(size_t bytes) {
    struct stat st;
    if (g_depth_fd >= 0 && (g_depth_bytes != bytes || stat("cu_depth.dat", &st) != 0 || fstat(g_depth_fd, &st) != 0 || st.st_nlink == 0)) {
        if (g_depth_map) munmap(g_depth_map, g_depth_bytes);
        g_depth_map = NULL;
        close(g_depth_fd);
        g_depth_fd = -1;
    }
    if (g_depth_fd < 0) {
        g_depth_fd = open("cu_depth.dat", O_RDWR | O_CREAT | O_TRUNC, 0644);
        g_depth_bytes = bytes;
        if (g_depth_fd >= 0 && bytes > 0 && ftruncate(g_depth_fd, (off_t)bytes) == 0) {
            void* m = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, g_depth_fd, 0);
            g_depth_map = m == MAP_FAILED ? NULL : m;
        }
    }
    return g_depth_fd < 0 ? -1 : 0;
}
static int write_cu_depth(const void* data, size_t bytes) {
    if (g_depth_fd < 0 || g_depth_bytes != bytes) return -1;
    if (g_depth_map) {
        memcpy(g_depth_map, data, bytes);
        return 0;
    }
    const char* p = (const char*)data;
    size_t done = 0;
    while (done < bytes) {
        const ssize_t r = pwrite(g_depth_fd, p + done, bytes - done, (off_t)done);
        if (r <= 0) return -1;
        done += (size_t)r;
    }
    return 0;
}

/* state.dat, refreshed behind the ending signal: rewritten IN PLACE through a shared mapping while its size stays the same (the
 * sidecar says "pending" around the write, which is what protects a restarted daemon from a torn file; HM never reads state.dat, and
 * the reference's daemon rewrites it in place too, resi_to_cu_depth_LDP.py:136-138).  Round 4 wrote a temp file and renamed it: on
 * tmpfs that allocates and frees 3.6 MB of fresh pages per 1920x1080 frame, next to the 3 MB the encoder's own resi.yuv rewrite
 * churns (profiles/r05_ldp_tail.txt).  First frame, a new geometry, a file somebody replaced: temp + rename as before. */
static int g_state_fd = -1;
static size_t g_state_bytes = 0;
static void* g_state_map = NULL;
static int write_state(const void* data, size_t bytes) {
    struct stat st, sp;
    if (g_state_fd >= 0 && (g_state_bytes != bytes || stat("state.dat", &sp) != 0 || fstat(g_state_fd, &st) != 0 || st.st_nlink == 0 ||
                            st.st_ino != sp.st_ino || (size_t)st.st_size != bytes)) {
        if (g_state_map) munmap(g_state_map, g_state_bytes);
        g_state_map = NULL;
        close(g_state_fd);
        g_state_fd = -1;
    }
    if (g_state_fd >= 0 && g_state_map) {
        memcpy(g_state_map, data, bytes);
        /* (the file's mtime must move: the resident-state check compares the signature recorded behind this write) */
        struct timespec now2[2] = {{0, UTIME_OMIT}, {0, UTIME_NOW}};
        (void)futimens(g_state_fd, now2);
        return 0;
    }
    if (write_atomic("state.dat", data, bytes) != 0) return -1;
    g_state_fd = open("state.dat", O_RDWR);
    g_state_bytes = bytes;
    if (g_state_fd >= 0 && bytes > 0) {
        void* m = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, g_state_fd, 0);
        g_state_map = m == MAP_FAILED ? NULL : m;
        if (!g_state_map) { close(g_state_fd); g_state_fd = -1; }
    }
    return 0;
}

/* --trace: per-stage times of up to TRACE_N frames; medians are printed (a mean is at the mercy of one 5 ms scheduling hiccup) */
#define TRACE_N 8192
static float g_tr[7][TRACE_N];
static int cmp_f(const void* a, const void* b) { const float x = *(const float*)a, y = *(const float*)b; return x < y ? -1 : x > y; }
static double median_us(float* v, int n) { if (n <= 0) return 0.0; qsort(v, (size_t)n, sizeof(float), cmp_f); return 1e6 * (double)v[n / 2]; }

int main(int argc, char** argv) {
    long max_frames = -1;
    double idle_timeout = -1.0;
    int quiet = 0, accept_stale = 0, trace = 0, no_stream_opt = 0, streamed_frames = 0, spin = 0;
    int sig_pending = 0; /* a frame has been accepted whose pred_start.sig is still there (it is removed off the critical path) */
    int n_trace = 0;
    double trace_slow = 0.0; /* --trace-slow US: frames that took this daemon longer than US microseconds are printed stage by stage */
    double bad_cmd_since = -1.0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--max-frames") && i + 1 < argc) max_frames = atol(argv[++i]);
        else if (!strcmp(argv[i], "--idle-timeout") && i + 1 < argc) idle_timeout = atof(argv[++i]);
        else if (!strcmp(argv[i], "--quiet")) quiet = 1;
        else if (!strcmp(argv[i], "--accept-stale")) accept_stale = 1;
        else if (!strcmp(argv[i], "--trace")) trace = 1;
        else if (!strcmp(argv[i], "--trace-slow") && i + 1 < argc) trace_slow = atof(argv[++i]) * 1e-6;
        else if (!strcmp(argv[i], "--spin")) spin = 1; /* busy-wait for pred_start.sig as the reference's daemon does (a core at 100 %) */
        else if (!strcmp(argv[i], "--no-stream")) no_stream_opt = 1; /* read resi.yuv first, then predict (A/B runs) */
        else { fprintf(stderr, "usage: resi_to_cu_depth_ldp [--max-frames N] [--idle-timeout S] [--quiet] [--accept-stale] [--trace] [--trace-slow US] [--spin] [--no-stream]\n"); return 2; }
    }
    ethcnn_ctx* ctx = NULL;
    ethcnn_options opt;
    memset(&opt, 0, sizeof opt);
    const char* dev = getenv("ETHCNN_DEVICE");
    opt.device = dev ? atoi(dev) : 0;
    if (ethcnn_create(&ctx, &opt) != ETHCNN_OK) { fprintf(stderr, "resi_to_cu_depth_ldp: create: %s\n", ethcnn_last_error(NULL)); return 1; }
    int rc = 1;
    uint8_t* luma = NULL;
    float* probs = NULL;
    float* state = NULL;
    size_t luma_cap = 0, probs_cap = 0, state_cap = 0;
    if (ethcnn_load_thresholds(ctx, "Thr_info.txt") != ETHCNN_OK || load_cnn(ctx) != ETHCNN_OK) {
        fprintf(stderr, "resi_to_cu_depth_ldp: %s\n", ethcnn_last_error(ctx));
        goto out;
    }
    {
        char name[160];
        ethcnn_device_name(ctx, name, sizeof name);
        if (!quiet) { printf("predictor initialized on %s.\n", name); fflush(stdout); }
    }
    { const int fd = open(".pred_end.sig.ethcnn", O_WRONLY | O_CREAT | O_TRUNC, 0644); if (fd >= 0) close(fd); } /* (see the ending signal below) */
    /* wake-up source: inotify on the working directory (falls back to a 50 us poll when it cannot be had) */
    int ifd = inotify_init1(IN_NONBLOCK);
    if (ifd >= 0 && inotify_add_watch(ifd, ".", IN_CREATE | IN_MOVED_TO | IN_CLOSE_WRITE) < 0) { close(ifd); ifd = -1; }

    long n_total = 0;
    int qp_seq = 0;
    int last_w = -1, last_h = -1, last_i = -1; /* geometry / frame of the state resident in HBM */
    file_sig state_sig = {0, 0, 0, 0, 0};      /* of the state.dat this daemon wrote for it */
    double idle_since = now_s(), spin_until = 0.0;
    while (max_frames < 0 || n_total < max_frames) {
        if (!file_exists("pred_start.sig")) {
            const double t = now_s();
            if (idle_timeout >= 0.0 && t - idle_since > idle_timeout) break;
            if (spin || t < spin_until) continue;             /* right behind a frame: the next one may follow at once */
            if (ifd >= 0) {
                struct pollfd pf = {ifd, POLLIN, 0};
                if (poll(&pf, 1, 100) > 0) {                  /* drain the events; the loop re-checks the file itself */
                    char ev[4096];
                    while (read(ifd, ev, sizeof ev) > 0) {}
                }
            } else {
                struct timespec ts = {0, 50000};
                nanosleep(&ts, NULL);
            }
            continue;
        }
        int i_frame, w, h, qp;
        const double ts0 = now_s();
        if (get_command(&i_frame, &w, &h, &qp) != 0 || i_frame < 0) {
            /* command.dat still being written -- or never going to be right: do not burn a core on it forever, and let --idle-timeout end
             * the daemon */
            if (bad_cmd_since < 0.0) bad_cmd_since = ts0;
            if (idle_timeout >= 0.0 && ts0 - bad_cmd_since > idle_timeout) break;
            if (ts0 - bad_cmd_since > 1e-3) { struct timespec ts = {0, 20000}; nanosleep(&ts, NULL); }
            continue;
        }
        bad_cmd_since = -1.0;
        const int qp_last = qp_seq;
        qp_seq = qp;
        sig_pending = 1;
        if (w <= 0 || h <= 0) { fprintf(stderr, "resi_to_cu_depth_ldp: bad geometry %dx%d in command.dat\n", w, h); goto out; }
        if (qp_seq != qp_last) {
            char name[96];
            if (load_lstm(ctx, qp_seq, name, sizeof name) != ETHCNN_OK) { fprintf(stderr, "resi_to_cu_depth_ldp: LSTM model: %s\n", ethcnn_last_error(ctx)); goto out; }
            if (!quiet) printf("Set QP = %d\nLSTM model loaded (%s).\n", qp_seq, name);
        }
        const size_t npx = (size_t)w * h, nctu = (size_t)((w + 63) / 64) * ((h + 63) / 64);
        if (npx > luma_cap || nctu * 21 > probs_cap) { /* page-locked buffers, grown only */
            if (luma) ethcnn_host_free(ctx, luma);
            if (probs) ethcnn_host_free(ctx, probs);
            luma = NULL; probs = NULL;
            luma_cap = npx > luma_cap ? npx : luma_cap;
            probs_cap = nctu * 21 > probs_cap ? nctu * 21 : probs_cap;
            void *a = NULL, *b = NULL;
            if (ethcnn_host_alloc(ctx, luma_cap, &a) != ETHCNN_OK || ethcnn_host_alloc(ctx, probs_cap * sizeof(float), &b) != ETHCNN_OK) {
                fprintf(stderr, "resi_to_cu_depth_ldp: %s\n", ethcnn_last_error(ctx));
                goto out;
            }
            luma = (uint8_t*)a; probs = (float*)b;
        }
        if (nctu * 896 > state_cap) {
            free(state);
            state_cap = nctu * 896;
            state = (float*)malloc(state_cap * sizeof(float));
            if (!state) { fprintf(stderr, "resi_to_cu_depth_ldp: out of memory\n"); goto out; }
        }
        const double ts1 = now_s();
        /* the state of frame i_frame - 1: resident when this daemon produced it for this geometry and its state.dat is untouched */
        const float* state_in = NULL;
        const int resident = i_frame > 1 && last_w == w && last_h == h && last_i == i_frame - 1 && sig_eq(state_sig, sig_of("state.dat"));
        if (i_frame > 1 && !resident) {
            char tag[96];
            FILE* fi = fopen("state.dat.idx", "r");
            if (fi) { /* our sidecar (the reference daemon writes none: its state.dat is taken as it is) */
                const size_t n = fread(tag, 1, sizeof tag - 1, fi);
                fclose(fi);
                tag[n] = 0;
                int ti, tw, th;
                if (strncmp(tag, "pending", 7) == 0) {
                    fprintf(stderr, "resi_to_cu_depth_ldp: state.dat is stale: %s(the daemon that served that frame stopped after its ending signal); "
                                    "delete state.dat.idx to accept state.dat as it is, or restart the encode\n", tag);
                    if (!accept_stale) goto out;
                } else if (sscanf(tag, "%d %d %d", &ti, &tw, &th) != 3 || tw != w || th != h) {
                    fprintf(stderr, "resi_to_cu_depth_ldp: state.dat belongs to another sequence (%s), frame %d is %dx%d\n", tag, i_frame, w, h);
                    goto out;
                }
            }
            if (read_exact("state.dat", state, nctu * 896 * sizeof(float)) != 0) {
                fprintf(stderr, "resi_to_cu_depth_ldp: state.dat: need %zu floats for frame %d\n", nctu * 896, i_frame);
                goto out;
            }
            state_in = state;
        }
        const double ts2 = now_s();
        /* helpers start copying resi.yuv; the frame's kernels are queued while they do and take the rows as they are reported */
        const int stream = !no_stream_opt && npx >= (512u << 10);
        const int no_stream = !stream;
        if (read_luma_start("resi.yuv", luma, w, h, no_stream ? NULL : ctx) != 0) { fprintf(stderr, "resi_to_cu_depth_ldp: cannot open resi.yuv\n"); goto out; }
        int step_rc = ETHCNN_OK;
        if (!no_stream) step_rc = ethcnn_ldp_step_begin(ctx, luma, w, h, w, qp_seq, i_frame, state_in, probs);
        const double ts2b = now_s();
        unlink("pred_start.sig"); /* (HM only ever creates it; removed off the critical path, while the helpers copy) */
        sig_pending = 0;
        const int read_rc = read_luma_finish(); /* (also when begin failed: the rows it may be waiting for are reported) */
        const double ts3 = now_s();
        /* while the GPU works: the sidecar says "pending" from here (it must, before the ending signal; a failure below ends the daemon
         * with the marker in place, which is what a restart has to see), cu_depth.dat is checked / re-created */
        char tagbuf[96];
        snprintf(tagbuf, sizeof tagbuf, "pending %d %d %d", i_frame, w, h);
        if (sidecar(tagbuf) != 0 || prepare_cu_depth(nctu * 21 * sizeof(float)) != 0) { fprintf(stderr, "resi_to_cu_depth_ldp: cannot write state.dat.idx / cu_depth.dat: %s\n", strerror(errno)); if (step_rc == ETHCNN_OK && !no_stream) (void)ethcnn_ldp_step_end(ctx); goto out; }
        if (step_rc == ETHCNN_OK) step_rc = no_stream ? ethcnn_ldp_step(ctx, luma, w, h, w, qp_seq, i_frame, state_in, probs) : ethcnn_ldp_step_end(ctx);
        if (read_rc != 0) { fprintf(stderr, "resi_to_cu_depth_ldp: resi.yuv: short read (%zu luma bytes wanted)\n", npx); goto out; }
        if (step_rc == ETHCNN_ERR_ROWS_TIMEOUT && !no_stream) { /* (only this one: any other error of begin / end is final -- ADVICE r05) */
            /* streamed frame: a row came more than ~1 s late (a cold or remote resi.yuv) and the kernels gave up.  The buffer is complete by
             * now and the library kept the previous frame's state resident: answer the frame the plain way instead of leaving HM spinning
             * on pred_end.sig */
            fprintf(stderr, "resi_to_cu_depth_ldp: frame %d: %s -- running it again on the complete picture\n", i_frame, ethcnn_last_error(ctx));
            step_rc = ethcnn_ldp_step(ctx, luma, w, h, w, qp_seq, i_frame, state_in, probs);
        }
        if (step_rc != ETHCNN_OK) {
            fprintf(stderr, "resi_to_cu_depth_ldp: frame %d: %s\n", i_frame, ethcnn_last_error(ctx));
            goto out;
        }
        const double ts4 = now_s();
        if (write_cu_depth(probs, nctu * 21 * sizeof(float)) != 0) { fprintf(stderr, "resi_to_cu_depth_ldp: cannot write cu_depth.dat: %s\n", strerror(errno)); goto out; }
        /* the ending signal: a second name for an empty file this daemon keeps (one link() instead of open + close; HM only ever
         * fopen()s and removes it), created the ordinary way where that fails */
        if (link(".pred_end.sig.ethcnn", "pred_end.sig") != 0) {
            const int fd = open("pred_end.sig", O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (fd < 0) { fprintf(stderr, "resi_to_cu_depth_ldp: cannot create pred_end.sig\n"); goto out; }
            close(fd);
        }
        const double ts5 = now_s();
        /* HM is encoding again from here; the state file is refreshed behind its back, as the protocol asks */
        if (ethcnn_ldp_get_state(ctx, state, nctu * 896) != ETHCNN_OK || write_state(state, nctu * 896 * sizeof(float)) != 0) {
            fprintf(stderr, "resi_to_cu_depth_ldp: cannot refresh state.dat: %s\n", ethcnn_last_error(ctx));
            goto out;
        }
        snprintf(tagbuf, sizeof tagbuf, "%d %d %d", i_frame, w, h);
        if (sidecar(tagbuf) != 0) goto out;
        last_w = w; last_h = h; last_i = i_frame;
        state_sig = sig_of("state.dat");
        ++n_total;
        streamed_frames += stream;
        if (trace_slow > 0.0 && n_total > 5 && ts5 - ts0 > trace_slow)
            fprintf(stderr, "slow frame %d: %.0f us from detection to the ending signal = command.dat + buffers %.0f | state %.0f | resi.yuv read %.0f "
                            "(begin %.0f) | sidecar + step %.0f | cu_depth.dat + pred_end.sig %.0f ; monotonic us: detected %.0f, ending signal %.0f\n",
                    i_frame, 1e6 * (ts5 - ts0), 1e6 * (ts1 - ts0), 1e6 * (ts2 - ts1), 1e6 * (ts3 - ts2), 1e6 * (ts2b - ts2), 1e6 * (ts4 - ts3),
                    1e6 * (ts5 - ts4), 1e6 * ts0, 1e6 * ts5);
        if (n_total > 5 && n_trace < TRACE_N) {
            const double ts6 = now_s();
            const double v[7] = {ts1 - ts0, ts2 - ts1, ts3 - ts2, ts2b - ts2, ts4 - ts3, ts5 - ts4, ts6 - ts5};
            for (int k = 0; k < 7; ++k) g_tr[k][n_trace] = (float)v[k];
            ++n_trace;
        }
        idle_since = now_s();
        spin_until = idle_since + 2e-3;
        if (!quiet) { printf("%ld frames predicted.\n", n_total); fflush(stdout); }
    }
    rc = 0;
    if (ifd >= 0) close(ifd);
    if (trace && n_trace > 0) {
        double m[7];
        for (int k = 0; k < 7; ++k) m[k] = median_us(g_tr[k], n_trace);
        fprintf(stderr, "trace (median us per frame, frames 6..%ld): command.dat %.1f | state %.1f | resi.yuv read%s %.1f (of which the caller spent %.1f queueing) | "
                        "%s %.1f | cu_depth.dat + pred_end.sig %.1f | behind the signal: state.dat refresh %.1f\n",
                n_total, m[0], m[1], !streamed_frames ? "" : " (kernels queued under it: ethcnn_ldp_step_begin)", m[2], m[3],
                !streamed_frames ? "sidecar + ethcnn_ldp_step" : "sidecar + ethcnn_ldp_step_end", m[4], m[5], m[6]);
    }
out:
    unlink(".pred_end.sig.ethcnn");
    if (sig_pending) unlink("pred_start.sig"); /* an error exit in between: the request counts as taken, as with the reference's daemon */
    free(state);
    ethcnn_destroy(ctx); /* frees the page-locked buffers with the context */
    return rc;
}
