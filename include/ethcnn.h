/*
 * ethcnn.h -- C ABI of libethcnn.so: MI355X-native ETH-CNN CU-partition predictor.
 *
 * Drop-in boundary for the reference's predictor step (all paths relative to
 * /root/reference):
 *
 *   HM-16.5_Test_AI/source/App/TAppEncoder/TAppEncCfg.cpp:2317-2321
 *       system("python video_to_cu_depth.py <yuv> <w> <h> <qp>"), assert(status == 0)
 *   HM-16.5_Test_AI/bin/video_to_cu_depth.py  (driver: read YUV, tile, sub-batch, write)
 *   HM-16.5_Test_AI/bin/net_CNN.py            (ETH-CNN graph + batch gates)
 *   HM-16.5_Test_AI/source/Lib/TLibEncoder/TEncCu.cpp:237-261  (consumer of cu_depth.dat)
 *
 * The reference's boundary is a process boundary with files; there is no FFI in it.  The
 * functions below are what a binding for this path binds: one entry per reference
 * function on the path (cited per function).  Conventions: plain C types only, caller
 * owns every buffer, 0 = success, negative = error (ethcnn_last_error() has the text),
 * no exceptions cross the ABI, a context is not thread-safe (one per thread / per GPU).
 *
 * There is NO CPU fallback: every compute entry point runs hand-written HIP kernels on a
 * gfx950 device and fails with ETHCNN_ERR_DEVICE if none is usable.
 */
#ifndef ETHCNN_H
#define ETHCNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ethcnn_ctx ethcnn_ctx;

enum {
    ETHCNN_OK = 0,
    ETHCNN_ERR_ARG = -1,       /* bad argument */
    ETHCNN_ERR_IO = -2,        /* file open/read/write failed */
    ETHCNN_ERR_FORMAT = -3,    /* malformed checkpoint / Thr_info.txt / YUV size */
    ETHCNN_ERR_DEVICE = -4,    /* no usable gfx950 device or a HIP call failed */
    ETHCNN_ERR_NOWEIGHTS = -5, /* predict called before any ethcnn_load_* */
    ETHCNN_ERR_NOMEM = -6,
    ETHCNN_ERR_PLAN_REFUSED = -8, /* an opted-in 16-bit plan (ethcnn_set_fc1_plan 2 / 3) failed its load-time accuracy guard for the loaded
                                     weights; nothing was computed.  Plan 0 (and usually plan 2 when 3 was refused) still works. */
    ETHCNN_ERR_ROWS_TIMEOUT = -7 /* streamed input only (ethcnn_predict_luma_end / ethcnn_ldp_step_end): the kernels waited ~1 s for a CTU row
                                    that ethcnn_rows_ready never reported and gave up.  Nothing is wrong with the device: the same picture can
                                    be run again the plain way once its buffer is complete.  Every OTHER error of those calls is final. */
};

/* Geometry constants of the path (net_CNN.py:8-36). */
#define ETHCNN_CTU 64
#define ETHCNN_NOUT 21          /* 1 + 4 + 16 split probabilities per CTU            */
#define ETHCNN_NFEAT 2688       /* NUM_CONVLAYER_FLAT_FILTERS                        */
#define ETHCNN_NVEC 448         /* 64 + 128 + 256 (FC1 outputs; LDP VECTOR_LENGTH)   */
#define ETHCNN_SUB_BATCH 1024   /* video_to_cu_depth.py:64 sub_batch_size (gate scope)*/
#define ETHCNN_BLOB_FLOATS 1288210 /* 5,152,840-byte TF-V2 .data payload, fp32        */

typedef struct ethcnn_options {
    int device;            /* HIP device ordinal (default 0)                                 */
    int max_ctus_per_pass; /* workspace size in CTUs; 0 = default = maximum (131072: what the
                              kernels' 32-bit offsets cover; larger values are clamped). Frames
                              are never split across passes unless a single frame exceeds it. */
    int host_threads;      /* staging-fill threads of the host / file entry points; 0 = automatic:
                              ethcnn_host_thread_budget(local workers, usable CPUs)            */
    int reserved[5];
} ethcnn_options;

/* ---- lifecycle (replaces tf.Session()/Saver construction, video_to_cu_depth.py:22-29) */
int ethcnn_create(ethcnn_ctx** out, const ethcnn_options* opt /* may be NULL */);
void ethcnn_destroy(ethcnn_ctx* ctx);
const char* ethcnn_last_error(const ethcnn_ctx* ctx); /* ctx may be NULL: create() errors */
const char* ethcnn_version(void);

/* ---- weights (replaces saver.restore(sess, 'model_2000000_qpXX~YY.dat'),
 *      video_to_cu_depth.py:126-133).  The "blob" is the payload of a TF-V2
 *      .data-00000-of-00001 file: 36 fp32 tensors back to back in key order. */
int ethcnn_load_checkpoint(ethcnn_ctx* ctx, const char* prefix); /* prefix.index + prefix.data-00000-of-00001, crc32c-checked */
int ethcnn_load_blob(ethcnn_ctx* ctx, const float* blob, size_t nfloats);
int ethcnn_load_synthetic(ethcnn_ctx* ctx, uint64_t seed, double head_gain); /* trained blobs are absent from the reference */
int ethcnn_get_blob(const ethcnn_ctx* ctx, float* blob_out, size_t nfloats);
/* QP band -> checkpoint prefix (qp<25, <30, <35, else): video_to_cu_depth.py:126-133. */
int ethcnn_model_name_for_qp(int qp, char* out, size_t cap);

/* ---- thresholds (replaces net_CNN.get_thresholds('Thr_info.txt'), net_CNN.py:38-47:
 *      tokens [1] and [3] of the first line split on single spaces). */
int ethcnn_load_thresholds(ethcnn_ctx* ctx, const char* thr_info_path);
int ethcnn_parse_thresholds(const char* thr_info_path, float* thr_l1_lower, float* thr_l2_lower); /* no context / device needed */
int ethcnn_set_thresholds(ethcnn_ctx* ctx, float thr_l1_lower, float thr_l2_lower);
int ethcnn_get_thresholds(const ethcnn_ctx* ctx, float* thr_l1_lower, float* thr_l2_lower);

/* ---- prediction.  Replaces get_prob() (video_to_cu_depth.py:75-118) =
 *      get_Y_for_one_frame (:46-59) + tiling loop (:88-106) + get_y_conv_on_large_data
 *      (:61-73) + net_CNN.net (net_CNN.py:103-195).
 *      luma: 8-bit planes, `pitch` bytes between rows, `frame_stride` bytes between frames
 *      (w*h*3/2 when pointing into a 4:2:0 file image).  probs: float32
 *      [nframes][ceil(h/64)*ceil(w/64)][21], CTUs in raster order, row = [p64,p32[4],p16[16]]
 *      -- the cu_depth.dat layout TEncCu::compressCtu freads (TEncCu.cpp:237-261). */
int ethcnn_predict_luma_device(ethcnn_ctx* ctx, const uint8_t* d_luma, int width, int height,
                               ptrdiff_t pitch, ptrdiff_t frame_stride, int nframes, int qp,
                               float* d_probs); /* both pointers in HBM; asynchronous on the ctx stream */
int ethcnn_predict_luma(ethcnn_ctx* ctx, const uint8_t* luma, int width, int height,
                        ptrdiff_t pitch, ptrdiff_t frame_stride, int nframes, int qp,
                        float* probs); /* host pointers; synchronous */
/* ONE picture with STREAMED INPUT (the in-process encoder hook, INTEGRATION.md: TEncGOP hands over a 16-bit picture whose
 * conversion to 8 bits takes as long as the prediction -- the prediction runs under it):
 *   ethcnn_predict_luma_begin   queues the picture's pass on `luma` -- a tightly packed plane (pitch == width) in a buffer from
 *                               ethcnn_host_alloc that the caller has NOT filled yet -- and returns.  One picture of fewer than
 *                               8192 CTUs.  Until ethcnn_predict_luma_end no other call may be made on this context except
 *                               ethcnn_rows_ready.
 *   ethcnn_rows_ready           "luma rows [64 ctu_row_begin, 64 ctu_row_end) are in the buffer" (the last CTU row may be short).
 *                               (The call orders the caller's earlier stores, non-temporal ones included, before the report.)
 *                               Any thread, any order, each CTU row once; also BEFORE the begin of the same picture, but not before
 *                               the previous streamed call on this context has ended (a begin that FAILS consumes the picture: rows
 *                               reported for it are forgotten, and no further row of it may be reported).  Every CTU row
 *                               [0, ceil(height / 64)) must be reported: kernels that wait ~1 s for a row give up and the end call
 *                               fails with ETHCNN_ERR_ROWS_TIMEOUT (the GPU is not left hanging).  Shared with ethcnn_ldp_step_begin below.
 *   ethcnn_predict_luma_end     waits; probs (the pointer given to begin) are final when it returns ETHCNN_OK.
 * Results are bit-identical to ethcnn_predict_luma's. */
int ethcnn_predict_luma_begin(ethcnn_ctx* ctx, const uint8_t* luma, int width, int height, int qp, float* probs);
int ethcnn_rows_ready(ethcnn_ctx* ctx, int ctu_row_begin, int ctu_row_end);
int ethcnn_predict_luma_end(ethcnn_ctx* ctx);
/* Whole driver: all frames of the file from frame 0 (:135-140), writes `out_path`
 * (temp file + rename: never a partial cu_depth.dat).  *nframes_out may be NULL. */
int ethcnn_predict_yuv_file(ethcnn_ctx* ctx, const char* yuv_path, int width, int height, int qp,
                            const char* out_path, int64_t* nframes_out);

/* Multi-GPU sharding (no collective): one worker per GPU handles frames
 * [frame_begin, frame_end) and pwrites them at frame_begin * nctu * 84 into `out_path`,
 * which must already exist with its final size (frames * nctu * 84 bytes). */
int ethcnn_predict_yuv_shard(ethcnn_ctx* ctx, const char* yuv_path, int width, int height, int qp,
                             const char* out_path, int64_t frame_begin, int64_t frame_end);

/* The whole file over several GPUs from ONE process, a worker thread per listed device (thread-per-GPU form of the same split: the
 * reference's caller blocks in system(), TAppEncCfg.cpp:2317-2321, so what counts is the command's wall time -- N interpreters, N
 * checkpoint parses and N cold contexts cost more than 1/N of a short job saves).  `ctx` is worker 0 and must live on devices[0]; a
 * device may be listed more than once (several workers sharing one GPU: tests, and profiles/r06_cold_start.txt).  Workers 1.. are
 * contexts the library creates once and keeps with `ctx` (a copy of its weights, thresholds, plan; host threads: the node budget divided
 * by the worker count).  Frames [total k / n, total (k + 1) / n) to worker k, each pwriting at its offset into a temp file; one
 * rename.  Output byte-identical to ethcnn_predict_yuv_file.  ndevices == 1 is that call. */
int ethcnn_predict_yuv_file_sharded(ethcnn_ctx* ctx, const int* devices, int ndevices, const char* yuv_path, int width, int height, int qp,
                                    const char* out_path, int64_t* nframes_out);
/* the split both sharded forms use: worker k of n gets frames [floor(k F / n), floor((k + 1) F / n)).  Pure (no context, no device). */
int ethcnn_shard_range(int64_t nframes, int workers, int k, int64_t* frame_begin, int64_t* frame_end);

/* get_prob(yuv_name, ..., n_frames_start, n_frames_end, ...) (video_to_cu_depth.py:75-118): frames [frame_begin, frame_end) of the
 * file (the reference reads and discards the first n_frames_start frames, :86-87) -> an `out_path` that holds exactly those
 * frames, written to a temp file and renamed.  The reference's own call passes 0 and the frame count = ethcnn_predict_yuv_file. */
int ethcnn_predict_yuv_range(ethcnn_ctx* ctx, const char* yuv_path, int width, int height, int qp,
                             const char* out_path, int64_t frame_begin, int64_t frame_end);

/* ---- config #5 front-end: resi_cnn (HM-16.5_Test_LDP/bin/net_CNN_LSTM_one_step.py:151-199)
 *      fed as in resi_to_cu_depth_LDP.py:72-101.  One frame; vec = float32 [nctu][448]. */
int ethcnn_resi_vectors_device(ethcnn_ctx* ctx, const uint8_t* d_luma, int width, int height,
                               ptrdiff_t pitch, float* d_vec);
int ethcnn_resi_vectors(ethcnn_ctx* ctx, const uint8_t* luma, int width, int height,
                        ptrdiff_t pitch, float* vec);

/* ---- config #5 back-end ("next" row 1): one ETH-LSTM step + heads + gates per frame =
 *      what predict_cu_depth() fetches from sess.run (HM-16.5_Test_LDP/bin/
 *      resi_to_cu_depth_LDP.py:108-129; graph: net_CNN_LSTM_one_step.py:201-323).
 *      CNN weights: ethcnn_load_checkpoint("model_LDP_2000000_qp22~37.dat") (same 36-tensor
 *      table; only conv + FC1 are used).  LSTM weights: the 18-tensor bundle
 *      model_LDP_200000_qp{22,27,32,37}.dat, reloaded when the QP band changes (:166-179).
 *      state: float32 [nctu][2][448] = (c, h) per CTU, the layout of state.dat (:103-106,131-137);
 *      state_in NULL = zeros (i_frame <= 1).  efs = [qp/51*0.18, onehot4(i_frame % 4)].
 *      Gates are per 1024-CTU mini-batch of the frame (:118). */
#define ETHCNN_LSTM_BLOB_FLOATS 760078 /* 3,040,312-byte TF-V2 .data payload, fp32 */
int ethcnn_load_lstm_checkpoint(ethcnn_ctx* ctx, const char* prefix);
int ethcnn_load_lstm_blob(ethcnn_ctx* ctx, const float* blob, size_t nfloats);
int ethcnn_load_lstm_synthetic(ethcnn_ctx* ctx, uint64_t seed, double head_gain);
int ethcnn_get_lstm_blob(const ethcnn_ctx* ctx, float* blob_out, size_t nfloats);
int ethcnn_lstm_model_name_for_qp(int qp, char* out, size_t cap);
int ethcnn_lstm_step_device(ethcnn_ctx* ctx, const float* d_vec, const float* d_state_in /* may be NULL */,
                            int nctu, int qp, int i_frame, float* d_state_out, float* d_probs);
int ethcnn_ldp_predict_frame(ethcnn_ctx* ctx, const uint8_t* luma, int width, int height, ptrdiff_t pitch,
                             int qp, int i_frame, const float* state_in /* may be NULL */, float* state_out,
                             float* probs); /* host pointers; synchronous */
/* The same per-frame call with the recurrent (c, h) state RESIDENT in HBM between frames (the daemon's fast path:
 * state.dat is part of the reference's file protocol, but only the daemon itself ever reads it back --
 * resi_to_cu_depth_LDP.py:103-106 -- so the 2 x nctu x 3.5 KB PCIe round trip per frame is not needed):
 *   state_in != NULL          use (and upload) this state, as ethcnn_ldp_predict_frame does;
 *   state_in == NULL, i_frame <= 1   zeros (:109-110);
 *   state_in == NULL, i_frame > 1    the state the previous ethcnn_ldp_step left in HBM (error if there is none or the
 *                                     CTU count changed).
 * probs returns synchronously; ethcnn_ldp_get_state copies the new state out afterwards (off the encoder's critical
 * path: the daemon signals pred_end.sig first, then refreshes state.dat). */
int ethcnn_ldp_step(ethcnn_ctx* ctx, const uint8_t* luma, int width, int height, ptrdiff_t pitch, int qp, int i_frame,
                    const float* state_in /* may be NULL */, float* probs);
int ethcnn_ldp_get_state(ethcnn_ctx* ctx, float* state_out, size_t nfloats /* nctu * 896 */);
/* ethcnn_ldp_step with STREAMED INPUT: the frame's kernels are queued BEFORE its luma is in memory, and consume it CTU row by CTU row
 * while the caller is still filling the buffer -- the daemon's threads copying resi.yuv out of the page cache
 * (resi_to_cu_depth_LDP.py:116-118 reads the file, THEN predicts: 52 us + 115 us at 1920x1080; streamed, the transfer over PCIe
 * and the launch overheads run under the read).
 *   ethcnn_ldp_step_begin   arguments as ethcnn_ldp_step; luma MUST lie in a buffer from ethcnn_host_alloc (read in place).  Returns
 *                           once everything is queued.  Until ethcnn_ldp_step_end no other call may be made on this context
 *                           except ethcnn_rows_ready.
 *   ethcnn_rows_ready   "luma rows [64 ctu_row_begin, 64 ctu_row_end) are in the buffer" (the last CTU row may be short).  Any
 *                           thread, any order, each CTU row once; may be called BEFORE ethcnn_ldp_step_begin of the same frame, but
 *                           not before the previous streamed step has ended.  Every CTU row [0, ceil(height / 64)) must be
 *                           reported: kernels that wait ~1 s for a row give up, and ethcnn_ldp_step_end then fails with
 *                           ETHCNN_ERR_ROWS_TIMEOUT (the GPU is not left hanging; the step's INPUT state stays resident -- the state the
 *                           previous step left, or the copy of state_in --, so the frame can be run again with ethcnn_ldp_step and the
 *                           same arguments once its buffer is complete).
 *   ethcnn_ldp_step_end     waits; probs (the pointer given to begin) and the resident state are final when it returns ETHCNN_OK.
 * Results are bit-identical to ethcnn_ldp_step's. */
int ethcnn_ldp_step_begin(ethcnn_ctx* ctx, const uint8_t* luma, int width, int height, ptrdiff_t pitch, int qp, int i_frame,
                          const float* state_in /* may be NULL */, float* probs);
/* (ethcnn_rows_ready: declared with ethcnn_predict_luma_begin above) */
int ethcnn_ldp_step_end(ethcnn_ctx* ctx);
/* Pinned (page-locked) host memory: buffers a caller fills itself (file reads) and hands to the host entry points are
 * DMA-able directly, without the runtime's pageable staging copy.  ethcnn_ldp_step goes further: a luma / probs pointer that
 * lies inside such a buffer is read / written by the kernels IN PLACE (no copy launch at all); so does ethcnn_predict_luma for ONE
 * tightly packed picture (width % 16 == 0, up to 2304 CTUs): the launch pulls it over PCIe itself while it computes.  Buffers still allocated when
 * the context is destroyed are freed with it. */
int ethcnn_host_alloc(ethcnn_ctx* ctx, size_t bytes, void** out);
int ethcnn_host_free(ethcnn_ctx* ctx, void* p);

/* ---- host budget (multi-GPU): one process per GPU, and every process fills its pinned staging ring with a
 *      pool of host threads.  The node's CPU budget (cgroup quota, else logical CPUs / 2) is DIVIDED by the number of
 *      predictor processes on the node (env ETHCNN_LOCAL_WORKERS, else torchrun's LOCAL_WORLD_SIZE, else 1), capped at 16.
 *      ethcnn_host_thread_budget is pure (no context, no device): usable <= 0 probes the box. */
int ethcnn_host_thread_budget(int local_workers, int usable_cpus);
int ethcnn_host_threads(ethcnn_ctx* ctx); /* the pool size this context uses */

/* ---- device plumbing for callers without a HIP binding (ctypes, cgo, JNI ...) */
int ethcnn_device_alloc(ethcnn_ctx* ctx, size_t bytes, void** out);
int ethcnn_device_free(ethcnn_ctx* ctx, void* p);
int ethcnn_memcpy_h2d(ethcnn_ctx* ctx, void* dst, const void* src, size_t bytes);
int ethcnn_memcpy_d2h(ethcnn_ctx* ctx, void* dst, const void* src, size_t bytes);
/* Waits for everything enqueued on the context's main stream.  When the LAST thing enqueued is a single-launch pass or an LDP
 * step, the wait is a spin on a completion word in page-locked memory that the launch's last block stores (about 5 us sooner
 * than hipStreamSynchronize; bounded, with hipStreamSynchronize as the fallback, which also reports device faults); otherwise it
 * is hipStreamSynchronize.  Env ETHCNN_DONE_WORD=0: always the latter. */
int ethcnn_synchronize(ethcnn_ctx* ctx);
int ethcnn_device_name(const ethcnn_ctx* ctx, char* out, size_t cap);
/* Cold start (no reference counterpart; the reference quotes "1~10 s" of TensorFlow initialisation, README.md:124): how long
 * ethcnn_create took, and how much of it was its first HIP call (loading and initialising the runtime).  Milliseconds. */
int ethcnn_get_startup_times(const ethcnn_ctx* ctx, double* runtime_init_ms, double* create_ms);

/* ---- measurement: per-stage kernel time from HIP events recorded on the ctx stream
 *      (replaces the reference's only instrument, the 'Predicting Time' wall clock,
 *      video_to_cu_depth.py:142-145). */
enum {
    ETHCNN_STAGE_TILE = 0,  /* k0: CTU load + zero-pad tiling + integer pooling (HBM-bound) */
    ETHCNN_STAGE_TRUNK = 1, /* k1: mean removal + 3 conv stages x 21 units (MFMA)          */
    ETHCNN_STAGE_FC1 = 2,   /* k2: [N,2688]x[2688,448] (MFMA) -- dominant kernel           */
    ETHCNN_STAGE_HEADS = 3, /* k3: FC2 + FC3 + sigmoid of the three heads, fused (MFMA)    */
    ETHCNN_STAGE_GATE = 4,  /* k5: batch-level gates (zero fill)                           */
    ETHCNN_NSTAGES = 5
};
typedef struct ethcnn_stage_times {
    double ms[ETHCNN_NSTAGES];       /* accumulated kernel time per stage since reset */
    int64_t launches[ETHCNN_NSTAGES];
    int64_t ctus;                    /* CTUs processed since reset */
    int64_t timed[ETHCNN_NSTAGES];   /* launches whose time is in ms[] (level 1 samples every 3rd FC1 stage) */
    int64_t timed_ctus[ETHCNN_NSTAGES]; /* CTUs of those launches: rate = work(timed_ctus) / ms */
    int64_t timing_errors;           /* event create/record/elapsed failures: those launches are not in ms[]/timed[] */
} ethcnn_stage_times;
int ethcnn_set_profiling(ethcnn_ctx* ctx, int level); /* 0 off; 1 events around the dominant kernel (FC1) on every 3rd
                                                          pass (an event pair costs ~12 us of stream time); 2 around every launch */
int ethcnn_get_stage_times(ethcnn_ctx* ctx, ethcnn_stage_times* out); /* synchronizes */
/* Pass pipeline (default on): the CTU-load stage of pass i+1 runs on a side stream beside FC1 of pass i (consecutive passes
 * of one call, or consecutive asynchronous ethcnn_predict_luma_device calls).  Results do not depend on it.  Off = every
 * stage of every pass in order on one stream: what per-stage timings (profiling level 2) should be read against, because
 * with the pipeline on the stage intervals overlap.  Synchronizes.  Environment: ETHCNN_OVERLAP=0 starts contexts with it off. */
int ethcnn_set_pass_pipeline(ethcnn_ctx* ctx, int on);
int ethcnn_reset_stage_times(ethcnn_ctx* ctx);
/* Arithmetic plan of the big (multi-launch) All-Intra passes.  FC1 ([N,2688] x
 * [2688,448]) is 78 % of the path's arithmetic and the exact-fp32 MFMA runs at 1/16 of the 16-bit matrix rate.
 *   0 (default)  exact fp32 on v_mfma_f32_16x16x4_f32: every result of the library is bit-identical to oracle/ethcnn_oracle.c;
 *   2 ("fast", FC1 as fp16 x 2)  passes that take the multi-launch path (more than 2304 CTUs, or rows that are not 16-byte
 *                aligned) run FC1 on the 16-bit matrix pipe: both operands as TWO fp16 pieces of the power-of-two scaled fp32 value
 *                (h0 + h1 represents it to 2^-24 relative; the feature scale comes from a bound derived from the conv weights at
 *                load, not from observation, so no piece can overflow), THREE products, fp32 accumulation;
 *   3 ("fast", everything as fp16 x 2)  plan 2, and the trunk's three conv layers (ethcnn_trunk_fast.hip: conv1 on the exact integer
 *                pixel sums -- its weight pieces carry the pixel scale and its accumulators start at the mean-removal constant, so no
 *                arithmetic per output beyond the leaky-ReLU --, conv2 / conv3 with fp16 x 2 splits; the CTU-load
 *                stage is folded into the trunk: one pass over the luma frames, no pixel records in HBM) and the heads' FC2 / FC3
 *                (ethcnn_heads_fast.hip, scaled residual pieces) on the 16-bit pipe as well.  The fastest plan.
 *   (1 was round 4's bf16 x 3 form of FC1; removed in round 5 -- slower than plan 2 and no more accurate -- and now an argument error.)
 * Measured against float64 these sums are as accurate as plan 0's fmaf chains (profiles/r04_bf16x3_probe.txt: the error of a 2688-
 * term fp32 sum is set by the roundings of its accumulation), but the ORDER of the fp32 additions differs, so probabilities agree
 * with plan 0 / the oracle to about 1e-6..1e-5, not bit for bit (tests: <= 1e-4, the north star's tolerance; thresholded decisions may
 * differ on knife edges only).  The batch gates are computed exactly in every plan.  The single-launch small pass and the LDP path
 * always use plan 0 -- which includes ONE picture handed to ethcnn_predict_luma (the host entry's latency path: a picture with 16-byte
 * aligned rows runs as single-launch passes, whole or in 1024-CTU pieces): the plans show in ethcnn_predict_luma_device, the file /
 * multi-frame host entries and the streamed entries of pictures above 2304 CTUs (tests/test_gpu_fast_plan.py).  Env ETHCNN_FC1_PLAN=2|3 starts contexts in that plan.  Takes effect with the next pass enqueued. */
int ethcnn_set_fc1_plan(ethcnn_ctx* ctx, int plan);
int ethcnn_get_fc1_plan(const ethcnn_ctx* ctx);
/* Load-time accuracy guard of plans 2 / 3.  The pieces of a split value are exact to 2^-24 relative only while the value stays within
 * about 2^12 of the GUARANTEED bound its scale was derived from; below that an absolute floor takes over.  Seeded and well-conditioned
 * trained weights are far inside; a checkpoint with a few outlier weights (a bound 10^3..10^9 above typical values) is not.  At the
 * first pass under a plan (and here, on request) the library pushes every such floor of the plan through the |weights| of the layers
 * behind it -- all errors aligned, activations always at their floor -- down to the probabilities: a rigorous upper bound, computed from
 * the weights alone (video_to_cu_depth.py:126-133: any of the four checkpoints a run may restore must be safe).  At or below 2.5e-5 (a
 * quarter of the 1e-4 contract) the plan is accepted on the bound alone.  Above it the worst case decides nothing (it is pessimistic by
 * the looseness of the very bounds it guards), so the plan is MEASURED once: a seeded calibration picture (640 CTUs: flat, low-contrast,
 * gradient, noise, edge and texture tiles) through the exact plan and through the plan at QP 22 and QP 37, gates open; max |dp| <= 2.5e-5 accepts, anything
 * else REFUSES: the pass returns ETHCNN_ERR_PLAN_REFUSED with the numbers in ethcnn_last_error, nothing is computed, the context stays
 * usable (ethcnn_set_fc1_plan(ctx, 0 or 2)).  Never a silent loss of accuracy, never a silent fallback.  ~3 ms, once per weight load.
 *   ethcnn_check_fc1_plan   ETHCNN_OK / ETHCNN_ERR_PLAN_REFUSED for the loaded weights; *apriori_bound, *measured (may be NULL; measured
 *                           < 0: the bound sufficed and nothing was run).
 * The launchers (video_to_cu_depth.py, tools/video_to_cu_depth.c) call it when ETHCNN_FC1_PLAN is set and continue with plan 0 after
 * printing the refusal. */
int ethcnn_check_fc1_plan(ethcnn_ctx* ctx, int plan, double* apriori_bound /* may be NULL */, double* measured /* may be NULL */);
/* the a-priori bound for a blob in host memory: no context, no device (pure host arithmetic).  ETHCNN_OK: the bound alone accepts */
int ethcnn_fast_plan_bound(const float* blob, size_t nfloats, int plan, double* prob_err_bound, double* feature_bound /* may be NULL */);
/* Single-launch small pass (default on): a pass of <= 2304 CTUs (up to one 3840x2160 picture) whose rows are 16-byte aligned (width, pitch, frame stride and
 * base pointer multiples of 16) -- one picture from the in-process encoder hook, every Low-Delay-P frame, the reference's own
 * 768x512 case -- runs CTU load + trunk -> FC1 -> heads -> gates as ONE kernel launch (a dataflow inside one grid, per-group /
 * per-tile completion counters; csrc/ethcnn_small.hip) instead of five dependent launches; the LDP front-end
 * (ethcnn_resi_vectors*) likewise as one launch instead of three.  Time is booked under ETHCNN_STAGE_FC1.  Off, or for other
 * geometries: the tile / trunk / FC1 / heads / gate launches.  Results do not depend on it.  Env: ETHCNN_SMALL=0. */
int ethcnn_set_small_pass_launch(ethcnn_ctx* ctx, int on);

/* Box calibration (no reference counterpart): what THIS GPU sustains in exact-fp32 MFMAs (v_mfma_f32_16x16x4_f32, nothing
 * else issued) over about `seconds` (0 < seconds <= 5) of pure matrix work, in TFLOP/s.  bench.py prints it beside the
 * roofline: boxes of one pool differ by several per cent in sustained clock, the data-sheet peak (157.3) does not. */
int ethcnn_measure_mfma_rate(ethcnn_ctx* ctx, double seconds, double* tflops);

/* ---- parity-test introspection: intermediates of the LAST pass, copied to host. */
enum {
    ETHCNN_DBG_FEATURES = 0, /* [n][2688] h_conv_flat (net_CNN.py:150)          */
    ETHCNN_DBG_FC1 = 1,      /* [n][448]  h_fc1_64|32|16 after leaky-ReLU       */
    ETHCNN_DBG_FC2 = 2,      /* [n][336]  h_fc2_64|32|16 after leaky-ReLU       */
    ETHCNN_DBG_LOGITS = 3,   /* [n][21]   pre-sigmoid                           */
    ETHCNN_DBG_RAW_PROBS = 4 /* [n][21]   before the gates                      */
};
int ethcnn_debug_fetch(ethcnn_ctx* ctx, int which, float* host_out, size_t nfloats);
/* FC2 / LOGITS / RAW_PROBS are only stored when capture is on (off by default: 1.7 KB/CTU of HBM
 * writes nobody reads in production); FEATURES and FC1 are the path's own buffers, always there. */
int ethcnn_set_debug_capture(ethcnn_ctx* ctx, int on);

/* ---- checkpoint introspection (the TF-V2 bundle reader on its own; used by the loader
 *      above and by the known-answer tests against the reference's .index files). */
typedef struct ethcnn_ckpt_entry {
    char name[64];
    int dtype, rank;       /* dtype 1 = DT_FLOAT */
    int64_t shape[4];
    int shard;
    int64_t offset, size;  /* bytes, into <prefix>.data-<shard>-of-<n> */
    uint32_t crc32c;       /* masked, as stored */
} ethcnn_ckpt_entry;
int ethcnn_ckpt_read_index(const char* index_path, ethcnn_ckpt_entry* entries, int cap, int* n_out,
                           char* err, size_t errcap);
uint32_t ethcnn_crc32c_masked(const void* data, size_t nbytes);
/* <prefix>.index + <prefix>.data-00000-of-00001 -> blob (crc-checked); no context / device needed */
int ethcnn_ckpt_read_blob(const char* prefix, float* blob_out, size_t nfloats, char* err, size_t errcap);
int ethcnn_ckpt_read_lstm_blob(const char* prefix, float* blob_out, size_t nfloats, char* err, size_t errcap);

#ifdef __cplusplus
}
#endif
#endif /* ETHCNN_H */
