/*
 * video_to_cu_depth -- native (C, no Python) form of the predictor step, same command line and
 * exit-status contract as the reference's driver:
 *
 *     video_to_cu_depth <yuv> <width> <height> <qp>          (cwd = HM's bin/ directory)
 *
 * replaces  system("python video_to_cu_depth.py <yuv> <w> <h> <qp>")
 * (/root/reference/HM-16.5_Test_AI/source/App/TAppEncoder/TAppEncCfg.cpp:2317-2321): reads
 * Thr_info.txt and model_2000000_qpXX~YY.dat.* from the cwd, writes cu_depth.dat, exits 0 on
 * success and 1 on any failure (HM asserts on the status).  It is also the smallest complete
 * consumer of include/ethcnn.h from plain C (built with gcc -std=c99 -pedantic), i.e. what an
 * in-process hook in HM would call.
 *
 * ETHCNN_SYNTHETIC_SEED=<n> [ETHCNN_HEAD_GAIN=<g>] opts into seeded synthetic weights when the
 * trained .data blob is absent (it is not in the reference repository); ETHCNN_DEVICE=<n> picks
 * the GPU.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ethcnn.h"

static int fail(const ethcnn_ctx* ctx, const char* what) {
    fprintf(stderr, "video_to_cu_depth: %s: %s\n", what, ethcnn_last_error(ctx));
    return 1;
}

static int file_exists(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return 0;
    fclose(f);
    return 1;
}

int main(int argc, char** argv) {
    ethcnn_ctx* ctx = NULL;
    ethcnn_options opt;
    char model[64], data[96];
    int64_t nframes = 0;
    int width, height, qp, rc;
    const char* seed = getenv("ETHCNN_SYNTHETIC_SEED");
    const char* dev = getenv("ETHCNN_DEVICE");

    if (argc != 5) {
        fprintf(stderr, "usage: %s <yuv> <width> <height> <qp>\n", argv[0]);
        return 1;
    }
    width = atoi(argv[2]);
    height = atoi(argv[3]);
    qp = atoi(argv[4]);
    memset(&opt, 0, sizeof opt);
    opt.device = dev ? atoi(dev) : 0;
    if (ethcnn_create(&ctx, &opt) != ETHCNN_OK) return fail(NULL, "create");
    if (ethcnn_load_thresholds(ctx, "Thr_info.txt") != ETHCNN_OK) { rc = fail(ctx, "Thr_info.txt"); goto out; }
    if (ethcnn_model_name_for_qp(qp, model, sizeof model) != ETHCNN_OK) { rc = fail(ctx, "model name"); goto out; }
    snprintf(data, sizeof data, "%s.data-00000-of-00001", model);
    if (file_exists(data) || !seed) {
        if (ethcnn_load_checkpoint(ctx, model) != ETHCNN_OK) { rc = fail(ctx, model); goto out; }
    } else {
        const char* gain = getenv("ETHCNN_HEAD_GAIN");
        if (ethcnn_load_synthetic(ctx, (uint64_t)strtoull(seed, NULL, 10), gain ? atof(gain) : 1.0) != ETHCNN_OK) {
            rc = fail(ctx, "synthetic weights");
            goto out;
        }
    }
    if (ethcnn_predict_yuv_file(ctx, argv[1], width, height, qp, "cu_depth.dat", &nframes) != ETHCNN_OK) {
        rc = fail(ctx, argv[1]);
        goto out;
    }
    printf("%lld frames predicted -> cu_depth.dat\n", (long long)nframes);
    rc = 0;
out:
    ethcnn_destroy(ctx);
    return rc;
}
