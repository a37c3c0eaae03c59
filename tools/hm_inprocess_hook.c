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

/* ---- per-picture entry (the real in-process hook) ------------------------------------------- */
static ethcnn_ctx* g_ctx = NULL;
static int g_qp = -1;
static unsigned char* g_luma8 = NULL; /* page-locked (ethcnn_host_alloc): DMA-ed to the GPU as it is, no staging copy */
static size_t g_luma8_cap = 0;
static long g_pictures = 0;

static void close_predictor(void) {
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
    int y;
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
            streamed entry refuses -- more than 8191 CTUs -- take the plain call below) */
        for (y = 0; y < height; ++y) {
            const short* src = luma + (size_t)y * stride;
            unsigned char* dst = g_luma8 + (size_t)y * width;
            convert_row(src, dst, width, shift);
            if (streamed && ((y & 63) == 63 || y == height - 1)) (void)ethcnn_rows_ready(g_ctx, y >> 6, (y >> 6) + 1);
        }
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
