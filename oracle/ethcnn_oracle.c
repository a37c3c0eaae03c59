/*
 * ethcnn_oracle.c -- CPU restatement of the reference ETH-CNN CU-partition predictor.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped path (libethcnn.so) never
 * links, includes or calls anything in oracle/.
 *
 * PARITY STATUS: pinned to the reference's own SERIALIZED TensorFlow graphs, *unpinned against a
 * TensorFlow binary*.  The reference's arithmetic lives in TensorFlow 1.x (not vendored, not
 * installed here, no network), every trained ETH-CNN weight blob is absent from /root/reference
 * (.MISSING_LARGE_BLOBS) and the reference holds no golden vectors for this path.  What IS pinned:
 *   - THE REFERENCE'S OWN PYTHON FILES, EXECUTED in the build container (tests/ref_exec.py runs
 *     video_to_cu_depth.py as __main__ with its real argv and resi_to_cu_depth_LDP.py as the daemon it is,
 *     over tests/tf_shim.py, a stand-in for the ~30 TensorFlow calls they make): every line between
 *     `import tensorflow` and the output bytes -- zero pad, tiling loop, 1024 sub-batching, layer wiring,
 *     qp / efs columns, both tf.cond gates, QP-band restore, LSTM state slicing, the protocol -- is the
 *     reference's; the arithmetic inside each tf.* op is a stand-in, twice: numpy restatements
 *     (tests/golden/ref_exec_golden.npz; this file <= 2.4e-6) and PyTorch's own fp32 CPU kernels
 *     (ref_exec_golden_torch.npz; this file <= 1.2e-5 canonical, 4.3e-6 literal), identical gate patterns
 *     (tests/test_ref_exec.py).  What stays unpinned is TensorFlow's own op kernels, nothing else;
 *   - the MetaGraphDefs TF 1.4.1 wrote from the authors' graphs (the .meta files next to the
 *     checkpoints), executed node by node without TensorFlow by tests/meta_graph.py: golden
 *     vectors tests/golden/meta_exec_golden.npz (gen_meta_exec_golden.py); this file agrees to
 *     <= 1e-6 on features, ungated probabilities and the LDP 448-vectors (tests/test_meta_graph.py).
 *     Not covered by a saved graph: the threshold gates (net_CNN.py:175,187) and the LSTM cell;
 *   - constants / strides / shapes against the same .meta (tests/golden/meta_constants.json),
 *   - tensor names / shapes / offsets against the reference's .index files,
 *   - an independent PyTorch-CPU restatement and an independent numpy float64 restatement
 *     (tests/golden/gen_golden.py), both written from the reference .py, agree with this
 *     file to ~1e-6 on committed golden vectors.
 *
 * What follows the reference (paths relative to /root/reference/HM-16.5_Test_AI/bin):
 *   preprocess  x*1/255, qp*1/51 ........................ net_CNN.py:105-106
 *   aver_pool (k=4 -> L, k=2 -> M, identity -> S) ....... net_CNN.py:62-63,126,132,138
 *   zero_mean_norm_local (16x16 block mean removal) ..... net_CNN.py:78-84
 *   non_overlap_conv (k=stride, VALID, +bias, leaky) .... net_CNN.py:86-92,127-141
 *   flatten + concat [c3S c3M c3L c2S c2M c2L] .......... net_CNN.py:143-150
 *   full_connect x3 per head with qp as LAST column ..... net_CNN.py:94-101,156-185
 *   batch-level gates on the fed sub-batch .............. net_CNN.py:175,187
 *   frame read, zero pad, raster tiling ................. video_to_cu_depth.py:46-59,88-106
 *   <=1024-CTU sub-batches inside a frame ............... video_to_cu_depth.py:61-73
 *   output row [y64, y32[4], y16[16]] ................... video_to_cu_depth.py:72
 *   resi_cnn preprocess (x-128)/255.0*10, FC1 only ...... HM-16.5_Test_LDP/bin/net_CNN_LSTM_one_step.py:151-199
 *
 * TensorFlow leaves the floating-point summation order of conv2d / matmul / avg_pool
 * unspecified (Eigen contraction), so ANY fixed order is an equally valid restatement.
 * Two orders are implemented:
 *   mode 0 "canonical": the order libethcnn.so's kernels use (fp32 MFMA == k-ordered fmaf
 *          chain, see DESIGN.md "Canonical arithmetic"); pooling / block means use exact
 *          integer sums of the u8 pixels (one rounding instead of a float sum chain).
 *          The HIP path must match this mode BIT-FOR-BIT (logits, probabilities, decisions).
 *   mode 1 "literal":   TF-op-by-TF-op float arithmetic in plain raster / ascending-k
 *          order (x = u8*c255 first, float avg-pool sums, float mean sums).  Used to show
 *          the canonical order is a rounding-level (<=~1e-6) re-association only.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NFEAT 2688
#define NH1 448
#define NOUT 21

/* ---- checkpoint ("blob") layout: float offsets into the TF-V2 .data file, fp32 LE.
 * Same table as /root/reference/HM-16.5_Test_AI/bin/model_2000000_qp*.dat.index
 * (tests/test_ckpt.py checks it against the real .index files). Conv variables are
 * unnamed and numbered in creation order: L = Variable..Variable_5, M = _6.._11,
 * S = _12.._17 (net_CNN.py:126-141 builds L, M, S in that order). */
enum { BR_S = 0, BR_M = 1, BR_L = 2 };
static const int OFF_CW[3][3] = { /* [branch][layer] weight offset (floats) */
    {13504 / 4, 14592 / 4, 20832 / 4},  /* S: Variable_12, _14, _16 */
    {51904 / 4, 52992 / 4, 1088 / 4},   /* M: Variable_6, _8, _10  */
    {0 / 4, 33248 / 4, 39488 / 4}};     /* L: Variable, _2, _4     */
static const int OFF_CB[3][3] = {
    {14528 / 4, 20736 / 4, 33120 / 4},  /* S: _13, _15, _17 */
    {52928 / 4, 59136 / 4, 13376 / 4},  /* M: _7, _9, _11   */
    {1024 / 4, 39392 / 4, 51776 / 4}};  /* L: _1, _3, _5    */
/* heads in output order 64, 32, 16 */
static const int N1[3] = {64, 128, 256}, N2[3] = {48, 96, 192}, N3[3] = {1, 4, 16};
static const int OFF_FC1W[3] = {4189792 / 4, 2813280 / 4, 60256 / 4};
static const int OFF_FC1B[3] = {4189536 / 4, 2812768 / 4, 59232 / 4};
static const int OFF_FC2W[3] = {5126176 / 4, 5076448 / 4, 4878688 / 4};
static const int OFF_FC2B[3] = {5125984 / 4, 5076064 / 4, 4877920 / 4};
static const int OFF_FC3W[3] = {5152644 / 4, 5151088 / 4, 5138720 / 4};
static const int OFF_FC3B[3] = {5152640 / 4, 5151072 / 4, 5138656 / 4};
#define BLOB_FLOATS (5152840 / 4)

int oracle_blob_floats(void) { return BLOB_FLOATS; }

/* OpenMP team size for the calls below (bench.py's cpu_baseline reports 1 thread and all cores) */
#ifdef _OPENMP
#include <omp.h>
int oracle_set_threads(int n) { const int old = omp_get_max_threads(); if (n > 0) omp_set_num_threads(n); return old; }
#else
int oracle_set_threads(int n) { (void)n; return 1; }
#endif

static inline float c255(void) { return 1.0f / 255.0f; }  /* 0x3b808081, .meta 'scalar'   */
static inline float c51(void) { return 1.0f / 51.0f; }    /* 0x3ca0a0a1, .meta 'scalar_1' */

/* tf.nn.leaky_relu default alpha 0.2, emitted as Maximum(alpha*x, x) (net_CNN.py:69) */
static inline float lrelu(float h) { return fmaxf(0.2f * h, h); }

/* exp() with a fully specified operation sequence (only fmaf / mul / rint / exponent
 * arithmetic) so that the HIP kernel can reproduce it bit-for-bit.  ~1 ulp. */
static inline float oracle_expf(float x) {
    if (x > 80.0f) x = 80.0f;   /* keeps 2^n * p normal; sigmoid is saturated far earlier */
    if (x < -86.0f) x = -86.0f;
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);           /* ln2 hi (exact in 12 bits) */
    r = fmaf(n, -1.42860682030941723212e-6f, r);         /* ln2 lo */
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    union { float f; int32_t i; } u;
    u.f = p;
    u.i += ((int32_t)n) << 23;  /* p in [0.70,1.42], n in [-124,115]: stays normal */
    return u.f;
}
float oracle_expf_export(float x) { return oracle_expf(x); }

static inline float sigmoidf(float z) { return 1.0f / (1.0f + oracle_expf(-z)); }

/* ---- accumulation orders (index = position in the chain, value = flattened HWI index
 * k = (ky*kw + kx)*cin + ci of the reference's HWIO weight tensor). */
static void build_orders(int mode, int ord1[16], int ord2[64], int ord3[96]) {
    int t;
    if (mode == 1) { /* literal: ascending */
        for (t = 0; t < 16; ++t) ord1[t] = t;
        for (t = 0; t < 64; ++t) ord2[t] = t;
        for (t = 0; t < 96; ++t) ord3[t] = t;
        return;
    }
    /* canonical (matches csrc/ethcnn_kernels.hip k1_trunk MFMA operand mapping):
     * conv1: kx outer, ky inner. */
    t = 0;
    for (int kx = 0; kx < 4; ++kx)
        for (int ky = 0; ky < 4; ++ky) ord1[t++] = ky * 4 + kx;
    /* conv2: patch q1 (raster), then r, then g with ci = 4g + r */
    t = 0;
    for (int q1 = 0; q1 < 4; ++q1)
        for (int r = 0; r < 4; ++r)
            for (int g = 0; g < 4; ++g) ord2[t++] = q1 * 16 + 4 * g + r;
    /* conv3 phase A: channels 0..15 of the 4 positions; phase B: channels 16..23,
     * two positions per MFMA k-step (lane-half swap). */
    t = 0;
    for (int q2 = 0; q2 < 4; ++q2)
        for (int r = 0; r < 4; ++r)
            for (int g = 0; g < 4; ++g) ord3[t++] = q2 * 24 + 4 * g + r;
    for (int j = 0; j < 2; ++j)
        for (int r = 0; r < 4; ++r)
            for (int g = 0; g < 4; ++g) ord3[t++] = (2 * j + (g >> 1)) * 24 + 16 + 4 * (g & 1) + r;
}

/* One "unit": a 16x16 block at branch resolution.  v = mean-removed input.
 * a2out[q2][24] (q2 = 2*qy+qx), a3out[32]. */
static void unit_forward(const float v[256], const float* blob, int br, int mode, const int ord1[16],
                         const int ord2[64], const int ord3[96], float a2out[4][24],
                         float a3out[32]) {
    /* canonical: the chain starts at the bias (MFMA C operand); literal: conv2d from 0, then + b */
    const int binit = (mode == 0);
    const float* w1 = blob + OFF_CW[br][0];
    const float* b1 = blob + OFF_CB[br][0];
    const float* w2 = blob + OFF_CW[br][1];
    const float* b2 = blob + OFF_CB[br][1];
    const float* w3 = blob + OFF_CW[br][2];
    const float* b3 = blob + OFF_CB[br][2];
    float a1[16][16];
    for (int py = 0; py < 4; ++py)
        for (int px = 0; px < 4; ++px) {
            float acc[16];
            for (int co = 0; co < 16; ++co) acc[co] = binit ? b1[co] : 0.0f;
            for (int t = 0; t < 16; ++t) {
                const int k = ord1[t], ky = k >> 2, kx = k & 3;
                const float in = v[(4 * py + ky) * 16 + 4 * px + kx];
                for (int co = 0; co < 16; ++co) acc[co] = fmaf(w1[k * 16 + co], in, acc[co]);
            }
            for (int co = 0; co < 16; ++co) a1[py * 4 + px][co] = lrelu(binit ? acc[co] : acc[co] + b1[co]);
        }
    for (int q2 = 0; q2 < 4; ++q2) {
        const int qy = q2 >> 1, qx = q2 & 1;
        float acc[24];
        for (int co = 0; co < 24; ++co) acc[co] = binit ? b2[co] : 0.0f;
        for (int t = 0; t < 64; ++t) {
            const int k = ord2[t], q1 = k >> 4, ci = k & 15, ky = q1 >> 1, kx = q1 & 1;
            const float in = a1[(2 * qy + ky) * 4 + 2 * qx + kx][ci];
            for (int co = 0; co < 24; ++co) acc[co] = fmaf(w2[k * 24 + co], in, acc[co]);
        }
        for (int co = 0; co < 24; ++co) a2out[q2][co] = lrelu(binit ? acc[co] : acc[co] + b2[co]);
    }
    {
        float acc[32];
        for (int co = 0; co < 32; ++co) acc[co] = binit ? b3[co] : 0.0f;
        for (int t = 0; t < 96; ++t) {
            const int k = ord3[t], q2 = k / 24, ci = k % 24;
            const float in = a2out[q2][ci];
            for (int co = 0; co < 32; ++co) acc[co] = fmaf(w3[k * 32 + co], in, acc[co]);
        }
        for (int co = 0; co < 32; ++co) a3out[co] = lrelu(binit ? acc[co] : acc[co] + b3[co]);
    }
}

/* integer -> network input value.  AI: x*1/255 (net_CNN.py:105).  resi: (x-128)/255.0*10
 * (net_CNN_LSTM_one_step.py:153); `cnt` raw pixels were summed into s. */
static inline float px_value(int s, int cnt, int resi) {
    if (resi) return ((float)(s - 128 * cnt) / 255.0f) * 10.0f;
    return (float)s * c255();
}

/* Features of one CTU.  ctu = 64x64 u8 row-major.  F[2688] in the reference's concat order. */
static void ctu_features(const uint8_t* ctu, const float* blob, int mode, int resi,
                         const int ord1[16], const int ord2[64], const int ord3[96], float* F) {
    static const int OFF3[3] = {0, 512, 640}, OFF2[3] = {672, 2208, 2592};
    static const int NB[3] = {4, 2, 1}, POOL[3] = {1, 2, 4};
    for (int br = 0; br < 3; ++br) {
        const int nb = NB[br], pool = POOL[br];
        const float scale = 1.0f / (float)(pool * pool); /* exact power of two */
        for (int by = 0; by < nb; ++by)
            for (int bx = 0; bx < nb; ++bx) {
                float x[256], v[256], mean;
                int isum[256];
                if (mode == 0) {
                    int T = 0;
                    for (int yy = 0; yy < 16; ++yy)
                        for (int xx = 0; xx < 16; ++xx) {
                            int s = 0;
                            const int y0 = (by * 16 + yy) * pool, x0 = (bx * 16 + xx) * pool;
                            for (int dy = 0; dy < pool; ++dy)
                                for (int dx = 0; dx < pool; ++dx) s += ctu[(y0 + dy) * 64 + x0 + dx];
                            T += s;
                            isum[yy * 16 + xx] = s;
                            x[yy * 16 + xx] = px_value(s, pool * pool, resi) * scale;
                        }
                    mean = px_value(T, 256 * pool * pool, resi) * (scale * (1.0f / 256.0f));
                } else {
                    float msum = 0.0f;
                    for (int yy = 0; yy < 16; ++yy)
                        for (int xx = 0; xx < 16; ++xx) {
                            float s = 0.0f;
                            const int y0 = (by * 16 + yy) * pool, x0 = (bx * 16 + xx) * pool;
                            for (int dy = 0; dy < pool; ++dy)
                                for (int dx = 0; dx < pool; ++dx)
                                    s += px_value(ctu[(y0 + dy) * 64 + x0 + dx], 1, resi);
                            x[yy * 16 + xx] = s * scale; /* AvgPool: sum / count */
                            msum += x[yy * 16 + xx] * (1.0f / 256.0f); /* conv with const 1/256 kernel */
                        }
                    mean = msum;
                }
                if (mode == 0 && !resi) { /* canonical AI centring: one rounding, v = fma(sum, c255 * 2^-p, -mean) */
                    const float c = c255() * scale;
                    for (int i = 0; i < 256; ++i) v[i] = fmaf((float)isum[i], c, -mean);
                } else {
                    for (int i = 0; i < 256; ++i) v[i] = x[i] - mean;
                }
                float a2[4][24], a3[32];
                unit_forward(v, blob, br, mode, ord1, ord2, ord3, a2, a3);
                memcpy(F + OFF3[br] + (by * nb + bx) * 32, a3, 32 * sizeof(float));
                for (int q2 = 0; q2 < 4; ++q2) {
                    const int y = 2 * by + (q2 >> 1), xx = 2 * bx + (q2 & 1);
                    memcpy(F + OFF2[br] + (y * 2 * nb + xx) * 24, a2[q2], 24 * sizeof(float));
                }
            }
    }
}

/* position t of the accumulation chain -> k.  literal: ascending.  canonical (FC1, FC2): inside
 * every 16-wide chunk the kernel's MFMA steps e = 0..3 take k = 16c + 4g + e for g = 0..3
 * (a lane loads one float4 of activations per chunk; csrc/ethcnn_dense.hip). */
static inline int fc_k(int t, int mode) {
    if (mode == 1) return t;
    const int r = t & 15;
    return (t & ~15) + 4 * (r & 3) + (r >> 2);
}

/* FC1 of the three heads: H1[448] = lrelu(F.W1 + b1), order [64 | 128 | 256]. */
static void ctu_fc1(const float* F, const float* blob, int mode, float* H1) {
    int o = 0;
    for (int h = 0; h < 3; ++h) {
        const int n1 = N1[h];
        const float* W = blob + OFF_FC1W[h];
        const float* b = blob + OFF_FC1B[h];
        float acc[256];
        for (int f = 0; f < n1; ++f) acc[f] = 0.0f;
        for (int t = 0; t < NFEAT; ++t) {
            const int k = fc_k(t, mode);
            const float a = F[k];
            const float* w = W + (size_t)k * n1;
            for (int f = 0; f < n1; ++f) acc[f] = fmaf(a, w[f], acc[f]);
        }
        for (int f = 0; f < n1; ++f) H1[o + f] = lrelu(acc[f] + b[f]);
        o += n1;
    }
}

/* The same FC1 for a block of up to FC1_CB CTUs at once: every (CTU, output) accumulator is the SAME fmaf chain in the
 * SAME k order as ctu_fc1 (bit-identical results; tests/test_oracle_golden.py compares them) -- only the loop nest differs:
 * a 16-output tile of W1 is swept once per CTU block instead of once per CTU, with the accumulators in registers, so
 * the 4.8 MB of FC1 weights are streamed 1/8 as often.  Without this the all-core CPU baseline of bench.py is
 * weight-bandwidth-bound (22 k CTU/s on 256 threads vs 7 k on one). */
#define FC1_CB 8
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
static void block_fc1(const float* F /* [nb][NFEAT] */, int nb, const float* blob, int mode, float* H1 /* [nb][NH1] */) {
    int o = 0;
    for (int h = 0; h < 3; ++h) {
        const int n1 = N1[h];
        const float* W = blob + OFF_FC1W[h];
        const float* b = blob + OFF_FC1B[h];
        for (int f0 = 0; f0 < n1; f0 += 16) {
            __m256 acc[FC1_CB][2];
            for (int c = 0; c < FC1_CB; ++c) acc[c][0] = acc[c][1] = _mm256_setzero_ps();
            for (int t = 0; t < NFEAT; ++t) {
                const int k = fc_k(t, mode);
                const __m256 w0 = _mm256_loadu_ps(W + (size_t)k * n1 + f0), w1 = _mm256_loadu_ps(W + (size_t)k * n1 + f0 + 8);
                for (int c = 0; c < nb; ++c) {
                    const __m256 a = _mm256_broadcast_ss(F + (size_t)c * NFEAT + k);
                    acc[c][0] = _mm256_fmadd_ps(a, w0, acc[c][0]); /* one rounding: fmaf per lane */
                    acc[c][1] = _mm256_fmadd_ps(a, w1, acc[c][1]);
                }
            }
            for (int c = 0; c < nb; ++c) {
                float tmp[16];
                _mm256_storeu_ps(tmp, acc[c][0]);
                _mm256_storeu_ps(tmp + 8, acc[c][1]);
                for (int j = 0; j < 16; ++j) H1[(size_t)c * NH1 + o + f0 + j] = lrelu(tmp[j] + b[f0 + j]);
            }
        }
        o += n1;
    }
}
#else
static void block_fc1(const float* F, int nb, const float* blob, int mode, float* H1) {
    for (int c = 0; c < nb; ++c) ctu_fc1(F + (size_t)c * NFEAT, blob, mode, H1 + (size_t)c * NH1);
}
#endif

/* FC2 + FC3 + sigmoid: probs_raw[21] (before the batch gates), optional logits[21]. */
static void ctu_heads(const float* H1, const float* blob, float qn, int mode, float* probs, float* logits) {
    int o1 = 0, o3 = 0;
    for (int h = 0; h < 3; ++h) {
        const int n1 = N1[h], n2 = N2[h], n3 = N3[h];
        const float* W2 = blob + OFF_FC2W[h];
        const float* b2 = blob + OFF_FC2B[h];
        const float* W3 = blob + OFF_FC3W[h];
        const float* b3 = blob + OFF_FC3B[h];
        float acc[192], h2[192];
        for (int j = 0; j < n2; ++j) acc[j] = 0.0f;
        for (int t = 0; t < n1; ++t) {
            const int k = fc_k(t, mode);
            const float a = H1[o1 + k];
            for (int j = 0; j < n2; ++j) acc[j] = fmaf(a, W2[k * n2 + j], acc[j]);
        }
        for (int j = 0; j < n2; ++j) acc[j] = fmaf(qn, W2[n1 * n2 + j], acc[j]); /* qp = last column */
        for (int j = 0; j < n2; ++j) h2[j] = lrelu(acc[j] + b2[j]);
        float z[16];
        for (int j = 0; j < n3; ++j) z[j] = 0.0f;
        for (int t = 0; t < n2; ++t) {
            const int k = fc_k(t, mode);
            for (int j = 0; j < n3; ++j) z[j] = fmaf(h2[k], W3[k * n3 + j], z[j]);
        }
        for (int j = 0; j < n3; ++j) {
            z[j] = fmaf(qn, W3[n2 * n3 + j], z[j]) + b3[j];
            if (logits) logits[o3 + j] = z[j];
            probs[o3 + j] = sigmoidf(z[j]);
        }
        o1 += n1;
        o3 += n3;
    }
}

/* net_CNN.py:175,187 over one fed sub-batch of n rows [n][21]. */
static void apply_gates(float* probs, int n, float thr1, float thr2) {
    int any1 = 0, any2 = 0;
    for (int i = 0; i < n; ++i)
        if (probs[i * NOUT] > thr1) any1 = 1;
    if (!any1)
        for (int i = 0; i < n; ++i)
            for (int j = 1; j < 5; ++j) probs[i * NOUT + j] = 0.0f;
    for (int i = 0; i < n; ++i)
        for (int j = 1; j < 5; ++j)
            if (probs[i * NOUT + j] > thr2) any2 = 1; /* uses the GATED y32 */
    if (!any2)
        for (int i = 0; i < n; ++i)
            for (int j = 5; j < 21; ++j) probs[i * NOUT + j] = 0.0f;
}

/* ------------------------------------------------------------------ exported API --- */

/* ctus [n][64][64] u8 -> F [n][2688].  mode 0 canonical / 1 literal; resi 0/1. */
int oracle_features(const float* blob, const uint8_t* ctus, int n, int mode, int resi, float* F) {
    int o1[16], o2[64], o3[96];
    build_orders(mode, o1, o2, o3);
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < n; ++i)
        ctu_features(ctus + (size_t)i * 4096, blob, mode, resi, o1, o2, o3, F + (size_t)i * NFEAT);
    return 0;
}

int oracle_fc1(const float* blob, const float* F, int n, int mode, float* H1) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < n; ++i) ctu_fc1(F + (size_t)i * NFEAT, blob, mode, H1 + (size_t)i * NH1);
    return 0;
}

/* H1 [n][448] -> ungated probabilities [n][21] (+ optional logits). */
int oracle_heads(const float* blob, const float* H1, int n, int qp, int mode, float* probs, float* logits) {
    const float qn = (float)qp * c51();
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < n; ++i)
        ctu_heads(H1 + (size_t)i * NH1, blob, qn, mode, probs + (size_t)i * NOUT,
                  logits ? logits + (size_t)i * NOUT : (float*)0);
    return 0;
}

int oracle_gates(float* probs, int n, int chunk, float thr1, float thr2) {
    for (int s = 0; s < n; s += chunk) apply_gates(probs + (size_t)s * NOUT, (n - s < chunk) ? n - s : chunk, thr1, thr2);
    return 0;
}

/* video_to_cu_depth.py:46-59,88-106: zero-padded raster tiling of one luma plane. */
int oracle_tile_frame(const uint8_t* luma, int w, int h, long pitch, uint8_t* ctus) {
    const int cw = (w + 63) / 64, ch = (h + 63) / 64;
    for (int cy = 0; cy < ch; ++cy)
        for (int cx = 0; cx < cw; ++cx) {
            uint8_t* t = ctus + (size_t)(cy * cw + cx) * 4096;
            for (int y = 0; y < 64; ++y)
                for (int x = 0; x < 64; ++x) {
                    const int yy = cy * 64 + y, xx = cx * 64 + x;
                    t[y * 64 + x] = (yy < h && xx < w) ? luma[(size_t)yy * pitch + xx] : 0;
                }
        }
    return cw * ch;
}

/* The whole get_prob() body for frames resident in memory (video_to_cu_depth.py:75-118):
 * per frame: tile, run the net in <=1024-CTU sub-batches (gates per sub-batch), append.
 * probs [nframes][nctu][21].  Returns 0. */
int oracle_predict_frames(const float* blob, const uint8_t* luma, int w, int h, long pitch,
                          long frame_stride, int nframes, int qp, float thr1, float thr2, int mode,
                          float* probs) {
    const int cw = (w + 63) / 64, ch = (h + 63) / 64, nctu = cw * ch;
    int o1[16], o2[64], o3[96];
    build_orders(mode, o1, o2, o3);
    const float qn = (float)qp * c51();
    /* frames are independent: work on groups of frames so that all host cores stay busy
     * (the gate scope stays one frame's <=1024-CTU sub-batch) */
    int group = 16384 / nctu;
    if (group < 1) group = 1;
    if (group > nframes) group = nframes;
    uint8_t* ctus = (uint8_t*)malloc((size_t)group * nctu * 4096);
    if (!ctus) return -1;
    for (int f0 = 0; f0 < nframes; f0 += group) {
        const int nf = (nframes - f0 < group) ? nframes - f0 : group;
#pragma omp parallel for schedule(static)
        for (int f = 0; f < nf; ++f)
            oracle_tile_frame(luma + (size_t)(f0 + f) * frame_stride, w, h, pitch, ctus + (size_t)f * nctu * 4096);
        float* P = probs + (size_t)f0 * nctu * NOUT;
#pragma omp parallel for schedule(dynamic, 1)
        for (int i0 = 0; i0 < nf * nctu; i0 += FC1_CB) {
            float F[FC1_CB][NFEAT], H1[FC1_CB][NH1];
            const int nb = (nf * nctu - i0 < FC1_CB) ? nf * nctu - i0 : FC1_CB;
            for (int c = 0; c < nb; ++c) ctu_features(ctus + (size_t)(i0 + c) * 4096, blob, mode, 0, o1, o2, o3, F[c]);
            block_fc1(&F[0][0], nb, blob, mode, &H1[0][0]);
            for (int c = 0; c < nb; ++c) ctu_heads(H1[c], blob, qn, mode, P + (size_t)(i0 + c) * NOUT, (float*)0);
        }
        for (int f = 0; f < nf; ++f) oracle_gates(P + (size_t)f * nctu * NOUT, nctu, 1024, thr1, thr2);
    }
    free(ctus);
    return 0;
}

/* config #5 front-end: resi.yuv luma -> vector [nctu][448] (resi_cnn). */
int oracle_resi_vectors(const float* blob, const uint8_t* luma, int w, int h, long pitch, int mode,
                        float* vec) {
    const int cw = (w + 63) / 64, ch = (h + 63) / 64, nctu = cw * ch;
    int o1[16], o2[64], o3[96];
    build_orders(mode, o1, o2, o3);
    uint8_t* ctus = (uint8_t*)malloc((size_t)nctu * 4096);
    if (!ctus) return -1;
    oracle_tile_frame(luma, w, h, pitch, ctus);
#pragma omp parallel for schedule(dynamic, 1)
    for (int i0 = 0; i0 < nctu; i0 += FC1_CB) {
        float F[FC1_CB][NFEAT];
        const int nb = (nctu - i0 < FC1_CB) ? nctu - i0 : FC1_CB;
        for (int c = 0; c < nb; ++c) ctu_features(ctus + (size_t)(i0 + c) * 4096, blob, mode, 1, o1, o2, o3, F[c]);
        block_fc1(&F[0][0], nb, blob, mode, vec + (size_t)i0 * NH1);
    }
    free(ctus);
    return 0;
}

/* ======================================================================================
 * "next" row 1: ETH-LSTM one step + LDP heads, config #5 completed.
 * Follows /root/reference/HM-16.5_Test_LDP/bin/net_CNN_LSTM_one_step.py:201-323 (lstm(), net())
 * as fed by resi_to_cu_depth_LDP.py:114-129 (predict_cu_depth: efs = [qp, i_frame % 4],
 * 1024-CTU mini-batches).  tf.contrib.rnn.LSTMCell(n, forget_bias=1.0, cell_clip=5.0), no
 * peepholes / projection (TF 1.x rnn_cell_impl.LSTMCell.call):
 *     z = [x, h_prev] . kernel + bias ;  i, j, f, o = split(z, 4)
 *     c = sigmoid(f + forget_bias) * c_prev + sigmoid(i) * tanh(j) ;  c = clip(c, -5, 5)
 *     h = sigmoid(o) * tanh(c)
 * then  h2 = lrelu([h, efs] W2 + b2),  y = sigmoid([h2, efs] W3 + b3),  efs = [qp/51*0.18,
 * onehot4(i_frame % 4)], and the same batch gates as the AI net.  The LSTM checkpoints'
 * .data blobs ARE in the reference (model_LDP_200000_qp{22,27,32,37}.dat, 3,040,312 B): layout
 * below = their .index.  Parity is still unpinned (no TensorFlow to run the cell), see header.
 * Canonical order: every matmul chain in the fc_k order over its leading (multiple-of-16)
 * inputs, then the 5 efs columns in order, then + bias.
 */
#define LSTM_BLOB_FLOATS (3040312 / 4)
/* per level (64, 32, 16 -> hidden 64, 128, 256): float offsets of fc2_b, fc2_w, fc3_b, fc3_w, bias, kernel */
static const int LOFF[3][6] = {{723640, 723688, 727000, 727001, 727054, 727310},
                               {578784, 578880, 591648, 591652, 592056, 592568},
                               {0, 192, 50304, 50320, 53472, 54496}};
int oracle_lstm_blob_floats(void) { return LSTM_BLOB_FLOATS; }

static inline float tanhf_c(float x) {
    const float e = oracle_expf(2.0f * x);
    return (e - 1.0f) / (e + 1.0f);
}

static void ctu_lstm(const float* lb, const float* vec, const float* st_in, float qpn, int phase, int mode,
                     float* probs, float* st_out) {
    float efs[5] = {qpn, 0.0f, 0.0f, 0.0f, 0.0f};
    efs[1 + phase] = 1.0f; /* tf.one_hot(i_frame % 4, depth 4) */
    int o1 = 0, o3 = 0;
    for (int lv = 0; lv < 3; ++lv) {
        const int n = N1[lv], n2 = N2[lv], n3 = N3[lv];
        const float* b2 = lb + LOFF[lv][0];
        const float* W2 = lb + LOFF[lv][1];
        const float* b3 = lb + LOFF[lv][2];
        const float* W3 = lb + LOFF[lv][3];
        const float* bk = lb + LOFF[lv][4];
        const float* K = lb + LOFF[lv][5];
        const float* x = vec + o1;
        const float* cp = st_in + o1;            /* state[...,0,:] = c */
        const float* hp = st_in + NH1 + o1;      /* state[...,1,:] = h */
        float z[1024], c[256], h[256];
        for (int j = 0; j < 4 * n; ++j) z[j] = 0.0f;
        for (int t = 0; t < 2 * n; ++t) {
            const int k = fc_k(t, mode);
            const float in = (k < n) ? x[k] : hp[k - n];
            const float* w = K + (size_t)k * 4 * n;
            for (int j = 0; j < 4 * n; ++j) z[j] = fmaf(in, w[j], z[j]);
        }
        for (int j = 0; j < 4 * n; ++j) z[j] += bk[j];
        for (int u = 0; u < n; ++u) {
            const float gi = z[u], gj = z[n + u], gf = z[2 * n + u], go = z[3 * n + u];
            float cc = sigmoidf(gf + 1.0f) * cp[u] + sigmoidf(gi) * tanhf_c(gj);
            cc = fminf(fmaxf(cc, -5.0f), 5.0f);
            c[u] = cc;
            h[u] = sigmoidf(go) * tanhf_c(cc);
        }
        memcpy(st_out + o1, c, n * sizeof(float));
        memcpy(st_out + NH1 + o1, h, n * sizeof(float));
        float acc[192], h2[192], zz[16];
        for (int j = 0; j < n2; ++j) acc[j] = 0.0f;
        for (int t = 0; t < n; ++t) {
            const int k = fc_k(t, mode);
            for (int j = 0; j < n2; ++j) acc[j] = fmaf(h[k], W2[k * n2 + j], acc[j]);
        }
        for (int e = 0; e < 5; ++e)
            for (int j = 0; j < n2; ++j) acc[j] = fmaf(efs[e], W2[(n + e) * n2 + j], acc[j]);
        for (int j = 0; j < n2; ++j) h2[j] = lrelu(acc[j] + b2[j]);
        for (int j = 0; j < n3; ++j) zz[j] = 0.0f;
        for (int t = 0; t < n2; ++t) {
            const int k = fc_k(t, mode);
            for (int j = 0; j < n3; ++j) zz[j] = fmaf(h2[k], W3[k * n3 + j], zz[j]);
        }
        for (int e = 0; e < 5; ++e)
            for (int j = 0; j < n3; ++j) zz[j] = fmaf(efs[e], W3[(n2 + e) * n3 + j], zz[j]);
        for (int j = 0; j < n3; ++j) probs[o3 + j] = sigmoidf(zz[j] + b3[j]);
        o1 += n;
        o3 += n3;
    }
}

/* vec [n][448] (resi_cnn output), state_in/out [n][2][448] (c then h; state_in NULL = zeros, as for
 * i_frame <= 1, resi_to_cu_depth_LDP.py:103-112), probs [n][21] with the gates per <=1024 mini-batch. */
int oracle_lstm_step(const float* lstm_blob, const float* vec, const float* state_in, int n, int qp, int i_frame,
                     float thr1, float thr2, int mode, float* probs, float* state_out) {
    const float qpn = ((float)qp / 51.0f) * 0.18f; /* net():283 qp / 51.0 * 0.18 */
    const int phase = ((i_frame % 4) + 4) % 4;
    float* zeros = (float*)calloc((size_t)2 * NH1, sizeof(float));
    if (!zeros) return -1;
#pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < n; ++i)
        ctu_lstm(lstm_blob, vec + (size_t)i * NH1, state_in ? state_in + (size_t)i * 2 * NH1 : zeros, qpn, phase, mode,
                 probs + (size_t)i * NOUT, state_out + (size_t)i * 2 * NH1);
    free(zeros);
    return oracle_gates(probs, n, 1024, thr1, thr2);
}
