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
 * the GPU; ETHCNN_DEVICES=0,1,2,... shards the frames over several GPUs from this ONE process
 * (ethcnn_predict_yuv_file_sharded: a worker thread per listed device, no collective; a device
 * may be listed twice).  ETHCNN_FC1_PLAN=2|3: the opted-in plan is checked against the restored
 * checkpoint (ethcnn_check_fc1_plan); a refusal is printed and the run continues with the exact
 * plan -- the encoder asserts a zero exit status.  ETHCNN_TIMING=1 prints where the command's
 * wall time goes (stderr): the blocking system() of the caller sees all of it.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "ethcnn.h"

static int fail(const ethcnn_ctx* ctx, const char* what) {
    fprintf(stderr, "video_to_cu_depth: %s: %s\n", what, ethcnn_last_error(ctx));
    return 1;
}

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
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
    int width, height, qp, rc, ndev = 0, devices[64];
    const char* seed = getenv("ETHCNN_SYNTHETIC_SEED");
    const char* dev = getenv("ETHCNN_DEVICE");
    const char* devs = getenv("ETHCNN_DEVICES");
    const char* timing = getenv("ETHCNN_TIMING");
    const double t0 = now_ms();
    double t_create, t_weights, t_guard, t_predict = t0, runtime_ms = 0.0, create_ms = 0.0;

    if (argc != 5) {
        fprintf(stderr, "usage: %s <yuv> <width> <height> <qp>\n", argv[0]);
        return 1;
    }
    width = atoi(argv[2]);
    height = atoi(argv[3]);
    qp = atoi(argv[4]);
    if (devs && *devs) { /* "0,1,2,3": worker k on the k-th entry */
        const char* q = devs;
        while (*q && ndev < 64) {
            char* end;
            const long d = strtol(q, &end, 10);
            if (end == q || d < 0) { fprintf(stderr, "video_to_cu_depth: bad ETHCNN_DEVICES '%s'\n", devs); return 1; }
            devices[ndev++] = (int)d;
            q = (*end == ',') ? end + 1 : end;
            if (*end != ',' && *end != '\0') { fprintf(stderr, "video_to_cu_depth: bad ETHCNN_DEVICES '%s'\n", devs); return 1; }
        }
    }
    if (ndev == 0) devices[ndev++] = dev ? atoi(dev) : 0;
    memset(&opt, 0, sizeof opt);
    opt.device = devices[0];
    if (ethcnn_create(&ctx, &opt) != ETHCNN_OK) return fail(NULL, "create");
    t_create = now_ms();
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
    t_weights = now_ms();
    if (ethcnn_get_fc1_plan(ctx) != 0) {
        const int g = ethcnn_check_fc1_plan(ctx, ethcnn_get_fc1_plan(ctx), NULL, NULL);
        if (g == ETHCNN_ERR_PLAN_REFUSED) {
            fprintf(stderr, "video_to_cu_depth: %s\nvideo_to_cu_depth: continuing with the exact plan (0)\n", ethcnn_last_error(ctx));
            (void)ethcnn_set_fc1_plan(ctx, 0);
        } else if (g != ETHCNN_OK) { rc = fail(ctx, "plan check"); goto out; }
    }
    t_guard = now_ms();
    if (ethcnn_predict_yuv_file_sharded(ctx, devices, ndev, argv[1], width, height, qp, "cu_depth.dat", &nframes) != ETHCNN_OK) {
        rc = fail(ctx, argv[1]);
        goto out;
    }
    t_predict = now_ms();
    printf("%lld frames predicted -> cu_depth.dat\n", (long long)nframes);
    if (timing && atoi(timing) != 0) {
        (void)ethcnn_get_startup_times(ctx, &runtime_ms, &create_ms);
        if (getenv("ETHCNN_T0_MS")) fprintf(stderr, "video_to_cu_depth timing: spawn -> main %.1f ms\n", t0 - atof(getenv("ETHCNN_T0_MS")));
        fprintf(stderr, "video_to_cu_depth timing (ms since main): create %.1f (HIP runtime init %.1f, context %.1f) | thresholds + weights %.1f | plan guard %.1f | "
                        "predict (%d worker%s) %.1f | total in main %.1f\n",
                t_create - t0, runtime_ms, create_ms - runtime_ms, t_weights - t_create, t_guard - t_weights, ndev, ndev == 1 ? "" : "s",
                t_predict - t_guard, t_predict - t0);
    }
    rc = 0;
    {   /* cu_depth.dat is complete and renamed into place: nothing is left that an orderly teardown would save.  ethcnn_destroy (12-40 ms)
         * and the HIP runtime's exit handlers (~45 ms) are a quarter of this command's wall time on the reference's own 768x512 case, and
         * the caller blocks on it (TAppEncCfg.cpp:2317-2321); the driver reclaims the process's GPU resources either way.
         * ETHCNN_FAST_EXIT=0 keeps the orderly path (sanitizer / leak-check runs). */
        const char* fe = getenv("ETHCNN_FAST_EXIT");
        if (!(fe && atoi(fe) == 0)) {
            fflush(stdout);
            fflush(stderr);
            _exit(0);
        }
    }
out:
    ethcnn_destroy(ctx);
    if (timing && atoi(timing) != 0) fprintf(stderr, "video_to_cu_depth timing: destroy %.1f ms\n", now_ms() - t_predict);
    return rc;
}
