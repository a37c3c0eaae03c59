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
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ethcnn.h"

int ethcnn_hm_predict(const char* yuv, int width, int height, int qp) {
    ethcnn_ctx* ctx = NULL;
    ethcnn_options opt;
    char model[64], data[96];
    int64_t nframes = 0;
    int rc = 1;
    const char* seed = getenv("ETHCNN_SYNTHETIC_SEED");
    const char* dev = getenv("ETHCNN_DEVICE");
    FILE* f;

    memset(&opt, 0, sizeof opt);
    opt.device = dev ? atoi(dev) : 0;
    if (ethcnn_create(&ctx, &opt) != ETHCNN_OK) {
        fprintf(stderr, "ethcnn_hm_predict: %s\n", ethcnn_last_error(NULL));
        return 1;
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
    if (ethcnn_predict_yuv_file(ctx, yuv, width, height, qp, "cu_depth.dat", &nframes) != ETHCNN_OK) goto fail;
    printf("ethcnn (in-process): %lld frames predicted on %s\n", (long long)nframes, "the GPU");
    rc = 0;
fail:
    if (rc) fprintf(stderr, "ethcnn_hm_predict: %s\n", ethcnn_last_error(ctx));
    ethcnn_destroy(ctx);
    return rc;
}
