/*
 * hm_inprocess_hook -- what an in-process HM integration links instead of spawning Python
 * (INTEGRATION.md section 4): one C function with the arguments of the
 * reference's command line,
 *
 *     system("python video_to_cu_depth.py <InputFile> <SourceWidth> <SourceHeight> <QP>")
 *     (/root/reference/HM-16.5_Test_AI/source/App/TAppEncoder/TAppEncCfg.cpp:2317-2321)
 *
 * becomes   assert(ethcnn_hm_predict(m_pchInputFile, m_iSourceWidth, m_iSourceHeight, m_iQP) == 0);
 *
 * Same files in the encoder's cwd (Thr_info.txt, model_2000000_qpXX~YY.dat.*), same output
 * (cu_depth.dat, read unchanged by TEncCu::compressCtu, TEncCu.cpp:237-261), 0 = success.
 * ETHCNN_SYNTHETIC_SEED / ETHCNN_HEAD_GAIN / ETHCNN_DEVICE as in the launchers.
 *
 * ethcnn_hm_predict_picture is the REAL in-process form (tools/hm_inprocess_patch.py applies it to a copy
 * of the encoder source): no predictor run before encoding, no second read of the YUV file, no
 * cu_depth.dat.  TEncCu::compressCtu, at the first CTU of every picture it encodes, hands over that
 * picture's own luma plane (TComPicYuv, 16-bit Pel samples with a stride) and receives the nCtu*21
 * probabilities it stores with TComPic::setCUDepth (TEncCu.cpp:256-259, TComPic.h:84-88).  Only the
 * pictures HM actually encodes are predicted, so FrameSkip / FramesToBeEncoded are honoured -- the
 * file-based reference predicts every frame from 0 and then reads cu_depth.dat from its start
 * whatever FrameSkip says (video_to_cu_depth.py:139-140).
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <pthread.h>

#include "ethcnn.h"
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

/* context + thresholds + weights for sequence QP `qp`, as the launcher sets them up; NULL on failure */
static ethcnn_ctx* open_predictor(int qp) {
    ethcnn_ctx* ctx = NULL;
    ethcnn_options opt;
    char model[64], data[96];
    const char* seed = getenv("ETHCNN_SYNTHETIC_SEED");
    const char* dev = getenv("ETHCNN_DEVICE");
    FILE* f;

    memset(&opt, 0, sizeof opt);
    opt.device = dev ? atoi(dev) : 0;
    if (ethcnn_create(&ctx, &opt) != ETHCNN_OK) {
        fprintf(stderr, "ethcnn (in-process): %s\n", ethcnn_last_error(NULL));
        return NULL;
    }
    if (ethcnn_load_thresholds(ctx, "Thr_info.txt") != ETHCNN_OK) goto fail;
    if (ethcnn_model_name_for_qp(qp, model, sizeof model) != ETHCNN_OK) goto fail;
    snprintf(data, sizeof data, "%s.data-00000-of-00001", model);
    f = fopen(data, "rb");
    if (f) fclose(f);
    if (f || !seed) {
        if (ethcnn_load_checkpoint(ctx, model) != ETHCNN_OK) goto fail;
    } else {
        const char* gain = getenv("ETHCNN_HEAD_GAIN");
        if (ethcnn_load_synthetic(ctx, (uint64_t)strtoull(seed, NULL, 10), gain ? atof(gain) : 1.0) != ETHCNN_OK) goto fail;
    }
    return ctx;
fail:
    fprintf(stderr, "ethcnn (in-process): %s\n", ethcnn_last_error(ctx));
    ethcnn_destroy(ctx);
    return NULL;
}

/* one row of HM's picture (Pel = 16-bit samples at the internal bit depth) -> 8 bits: v >> shift, clamped to [0, 255] */
static void convert_row(const short* src, unsigned char* dst, int width, int shift) {
    int x = 0;
#if defined(__SSE2__)
    /* sixteen samples per step: arithmetic shift, then the pack instruction's own saturation IS the clamp to [0, 255]
     * (the scalar loop below, which gcc -O2 leaves scalar, cost 880 us per 1920x1080 picture: nine predictions;
     * tests/test_abi_and_host.py checks the two forms equal for every sample value) */
    const __m128i vshift = _mm_cvtsi32_si128(shift);
    for (; x + 16 <= width; x += 16) {
        const __m128i a = _mm_sra_epi16(_mm_loadu_si128((const __m128i*)(src + x)), vshift);
        const __m128i b = _mm_sra_epi16(_mm_loadu_si128((const __m128i*)(src + x + 8)), vshift);
        _mm_storeu_si128((__m128i*)(dst + x), _mm_packus_epi16(a, b));
    }
#endif
    for (; x < width; ++x) {
        int v = src[x] >> shift;
        dst[x] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

/* ---- conversion helpers.  HM is single-threaded, and at 3840x2160 one thread needs 270 us for the plane (16.6 MB in, 8.3 MB out) while
 * the bus needs 155 us for it: pictures of two million samples and more are converted by the caller and three persistent helper
 * threads (ETHCNN_HM_THREADS, 1 = the caller alone), CTU rows drawn from a shared counter, each reported to the queued prediction
 * as it is finished.  The helpers sleep on a condition variable between pictures. */
#define HM_MAX_HELP 7
static struct {
    pthread_mutex_t mu;
    pthread_cond_t go, done;
    pthread_t th[HM_MAX_HELP];
    int nhelp, started, gen, pending, quit;
    const short* src;
    unsigned char* dst;
    int stride, width, height, shift, next, report;
    ethcnn_ctx* ctx;
} g_cv = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, -1, 0, 0, 0, 0, NULL, NULL, 0, 0, 0, 0, 0, 0, NULL};

static void convert_ctu_rows(void) { /* until no CTU row is left */
    const int nrows = (g_cv.height + 63) / 64;
    for (;;) {
        const int cy = __atomic_fetch_add(&g_cv.next, 1, __ATOMIC_RELAXED);
        int y;
        if (cy >= nrows) break;
        for (y = cy * 64; y < g_cv.height && y < cy * 64 + 64; ++y)
            convert_row(g_cv.src + (size_t)y * g_cv.stride, g_cv.dst + (size_t)y * g_cv.width, g_cv.width, g_cv.shift);
        if (g_cv.report) (void)ethcnn_rows_ready(g_cv.ctx, cy, cy + 1);
    }
}
static void* convert_helper(void* arg) {
    int seen = 0;
    (void)arg;
    pthread_mutex_lock(&g_cv.mu);
    for (;;) {
        while (g_cv.gen == seen && !g_cv.quit) pthread_cond_wait(&g_cv.go, &g_cv.mu);
        if (g_cv.quit) break;
        seen = g_cv.gen;
        pthread_mutex_unlock(&g_cv.mu);
        convert_ctu_rows();
        pthread_mutex_lock(&g_cv.mu);
        if (--g_cv.pending == 0) pthread_cond_signal(&g_cv.done);
    }
    pthread_mutex_unlock(&g_cv.mu);
    return NULL;
}
/* start: the helpers begin on the picture (big pictures only); finish: the caller converts rows too, then waits for them */
static void convert_start(const short* src, int stride, unsigned char* dst, int width, int height, int shift, ethcnn_ctx* ctx, int report) {
    pthread_mutex_lock(&g_cv.mu);
    if (g_cv.nhelp < 0) {
        const char* e = getenv("ETHCNN_HM_THREADS");
        const int n = e ? atoi(e) : 4;
        g_cv.nhelp = n < 1 ? 0 : (n - 1 > HM_MAX_HELP ? HM_MAX_HELP : n - 1);
    }
    g_cv.src = src; g_cv.stride = stride; g_cv.dst = dst; g_cv.width = width; g_cv.height = height; g_cv.shift = shift;
    g_cv.ctx = ctx; g_cv.report = report;
    __atomic_store_n(&g_cv.next, 0, __ATOMIC_RELAXED);
    g_cv.pending = 0;
    if (g_cv.nhelp > 0 && (size_t)width * (size_t)height >= 2000000u) {
        if (!g_cv.started) {
            int i;
            g_cv.started = 1;
            for (i = 0; i < g_cv.nhelp; ++i)
                if (pthread_create(&g_cv.th[i], NULL, convert_helper, NULL) != 0) { g_cv.nhelp = i; break; } /* (fewer helpers) */
        }
        if (g_cv.nhelp > 0) {
            g_cv.pending = g_cv.nhelp;
            ++g_cv.gen;
            pthread_cond_broadcast(&g_cv.go);
        }
    }
    pthread_mutex_unlock(&g_cv.mu);
}
static void convert_finish(void) {
    convert_ctu_rows();
    pthread_mutex_lock(&g_cv.mu);
    while (g_cv.pending > 0) pthread_cond_wait(&g_cv.done, &g_cv.mu);
    pthread_mutex_unlock(&g_cv.mu);
}
static void convert_shutdown(void) {
    int i, n;
    pthread_mutex_lock(&g_cv.mu);
    g_cv.quit = 1;
    n = g_cv.started ? g_cv.nhelp : 0;
    pthread_cond_broadcast(&g_cv.go);
    pthread_mutex_unlock(&g_cv.mu);
    for (i = 0; i < n; ++i) pthread_join(g_cv.th[i], NULL);
}

/* ---- per-picture entry (the real in-process hook) ------------------------------------------- */
static ethcnn_ctx* g_ctx = NULL;
static int g_qp = -1;
static unsigned char* g_luma8 = NULL; /* page-locked (ethcnn_host_alloc): DMA-ed to the GPU as it is, no staging copy */
static size_t g_luma8_cap = 0;
static long g_pictures = 0;

static void close_predictor(void) {
    convert_shutdown();
    if (g_ctx) {
        printf("ethcnn (in-process): %ld picture(s) predicted from the encoder's own luma buffers\n", g_pictures);
        ethcnn_destroy(g_ctx); /* frees the page-locked luma buffer with it */
    }
    g_ctx = NULL;
    g_luma8 = NULL;
    g_luma8_cap = 0;
}

/* luma: `height` rows of `width` 16-bit samples, `stride` samples apart, `bit_depth` bits each (HM's
 * Pel plane of the picture being encoded; the reference feeds 8-bit files, InternalBitDepth 8: deeper
 * internal formats are shifted back to the 8 bits the network was trained on).  probs: nCtu*21 floats,
 * the layout TEncCu::xCompressCU reads (TEncCu.cpp:419-463).  0 = success. */
int ethcnn_hm_predict_picture(const short* luma, int stride, int width, int height, int bit_depth, int qp, float* probs) {
    const int shift = bit_depth > 8 ? bit_depth - 8 : 0;
    const size_t need = (size_t)width * (size_t)height;
    if (!luma || !probs || width <= 0 || height <= 0 || stride < width) return 1;
    if (!g_ctx || qp != g_qp) {
        if (g_ctx) {
            ethcnn_destroy(g_ctx); /* its page-locked buffers go with it */
            g_luma8 = NULL;
            g_luma8_cap = 0;
        } else {
            atexit(close_predictor);
        }
        g_ctx = open_predictor(qp);
        g_qp = qp;
        if (!g_ctx) return 1;
    }
    if (need > g_luma8_cap) {
        void* p = NULL;
        if (g_luma8) (void)ethcnn_host_free(g_ctx, g_luma8);
        g_luma8 = NULL;
        g_luma8_cap = 0;
        if (ethcnn_host_alloc(g_ctx, need, &p) != ETHCNN_OK) return 1;
        g_luma8 = (unsigned char*)p;
        g_luma8_cap = need;
    }
    /* STREAMED: the picture's pass is queued first and takes the plane CTU row by CTU row while this loop is still converting it
     * (a 1920x1080 conversion costs about what the prediction does; ETHCNN_HM_STREAM=0: convert, then ethcnn_predict_luma) */
    {
        static int stream = -1;
        int streamed;
        if (stream < 0) { const char* e = getenv("ETHCNN_HM_STREAM"); stream = !(e && atoi(e) == 0); }
        streamed = stream && ethcnn_predict_luma_begin(g_ctx, g_luma8, width, height, qp, probs) == ETHCNN_OK;  /* (geometries the
            streamed entry refuses -- more than 8191 CTUs -- take the plain call below.  Begin comes BEFORE the first row is reported:
            rows reported for a picture whose begin then fails would count for the next one) */
        convert_start(luma, stride, g_luma8, width, height, shift, g_ctx, streamed);
        convert_finish();
        if ((streamed ? ethcnn_predict_luma_end(g_ctx) : ethcnn_predict_luma(g_ctx, g_luma8, width, height, width, (ptrdiff_t)need, 1, qp, probs)) != ETHCNN_OK) {
            fprintf(stderr, "ethcnn (in-process): %s\n", ethcnn_last_error(g_ctx));
            return 1;
        }
    }
    {   /* verification aid: ETHCNN_HM_DUMP=<file> appends every picture's probabilities (tests compare them
         * with the oracle); the encoder itself never reads it */
        const char* dump = getenv("ETHCNN_HM_DUMP");
        if (dump) {
            const size_t n = (size_t)((width + 63) / 64) * (size_t)((height + 63) / 64) * 21;
            FILE* f = fopen(dump, g_pictures ? "ab" : "wb");
            if (f) {
                fwrite(probs, sizeof(float), n, f);
                fclose(f);
            }
        }
    }
    ++g_pictures;
    return 0;
}

/* ---- whole-sequence entry (file-based, same contract as the reference's command line) -------- */
int ethcnn_hm_predict(const char* yuv, int width, int height, int qp) {
    ethcnn_ctx* ctx = open_predictor(qp);
    int64_t nframes = 0;
    int rc = 1;
    if (!ctx) return 1;
    if (ethcnn_predict_yuv_file(ctx, yuv, width, height, qp, "cu_depth.dat", &nframes) != ETHCNN_OK) goto fail;
    printf("ethcnn (in-process, file-based): %lld frames predicted on %s\n", (long long)nframes, "the GPU");
    rc = 0;
fail:
    if (rc) fprintf(stderr, "ethcnn_hm_predict: %s\n", ethcnn_last_error(ctx));
    ethcnn_destroy(ctx);
    return rc;
}
