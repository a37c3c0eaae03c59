// ethcnn_weights.cpp -- checkpoint tensor table, synthetic weights, device packing.
//
// Tensor table == the key/shape/offset table of the reference's TF-V2 checkpoints
// (/root/reference/HM-16.5_Test_AI/bin/model_2000000_qp*.dat.index).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "ethcnn_spec.h"

namespace ethcnn {

const TensorDesc kTensors[kNumTensors] = {
    {"Variable", 4, {4, 4, 1, 16}, 0},
    {"Variable_1", 1, {16, 0, 0, 0}, 1024},
    {"Variable_10", 4, {2, 2, 24, 32}, 1088},
    {"Variable_11", 1, {32, 0, 0, 0}, 13376},
    {"Variable_12", 4, {4, 4, 1, 16}, 13504},
    {"Variable_13", 1, {16, 0, 0, 0}, 14528},
    {"Variable_14", 4, {2, 2, 16, 24}, 14592},
    {"Variable_15", 1, {24, 0, 0, 0}, 20736},
    {"Variable_16", 4, {2, 2, 24, 32}, 20832},
    {"Variable_17", 1, {32, 0, 0, 0}, 33120},
    {"Variable_2", 4, {2, 2, 16, 24}, 33248},
    {"Variable_3", 1, {24, 0, 0, 0}, 39392},
    {"Variable_4", 4, {2, 2, 24, 32}, 39488},
    {"Variable_5", 1, {32, 0, 0, 0}, 51776},
    {"Variable_6", 4, {4, 4, 1, 16}, 51904},
    {"Variable_7", 1, {16, 0, 0, 0}, 52928},
    {"Variable_8", 4, {2, 2, 16, 24}, 52992},
    {"Variable_9", 1, {24, 0, 0, 0}, 59136},
    {"h_fc1__16__b", 1, {256, 0, 0, 0}, 59232},
    {"h_fc1__16__w", 2, {2688, 256, 0, 0}, 60256},
    {"h_fc1__32__b", 1, {128, 0, 0, 0}, 2812768},
    {"h_fc1__32__w", 2, {2688, 128, 0, 0}, 2813280},
    {"h_fc1__64__b", 1, {64, 0, 0, 0}, 4189536},
    {"h_fc1__64__w", 2, {2688, 64, 0, 0}, 4189792},
    {"h_fc2__16__b", 1, {192, 0, 0, 0}, 4877920},
    {"h_fc2__16__w", 2, {257, 192, 0, 0}, 4878688},
    {"h_fc2__32__b", 1, {96, 0, 0, 0}, 5076064},
    {"h_fc2__32__w", 2, {129, 96, 0, 0}, 5076448},
    {"h_fc2__64__b", 1, {48, 0, 0, 0}, 5125984},
    {"h_fc2__64__w", 2, {65, 48, 0, 0}, 5126176},
    {"y_conv_flat__16__b", 1, {16, 0, 0, 0}, 5138656},
    {"y_conv_flat__16__w", 2, {193, 16, 0, 0}, 5138720},
    {"y_conv_flat__32__b", 1, {4, 0, 0, 0}, 5151072},
    {"y_conv_flat__32__w", 2, {97, 4, 0, 0}, 5151088},
    {"y_conv_flat__64__b", 1, {1, 0, 0, 0}, 5152640},
    {"y_conv_flat__64__w", 2, {49, 1, 0, 0}, 5152644},
};

const TensorDesc kLstmTensors[kNumLstmTensors] = {
    {"RNN16/fc2/full_connect_b", 1, {192, 0, 0, 0}, 0},
    {"RNN16/fc2/full_connect_w", 2, {261, 192, 0, 0}, 768},
    {"RNN16/fc3/full_connect_b", 1, {16, 0, 0, 0}, 201216},
    {"RNN16/fc3/full_connect_w", 2, {197, 16, 0, 0}, 201280},
    {"RNN16/multi_rnn_cell/cell_0/lstm_cell/bias", 1, {1024, 0, 0, 0}, 213888},
    {"RNN16/multi_rnn_cell/cell_0/lstm_cell/kernel", 2, {512, 1024, 0, 0}, 217984},
    {"RNN32/fc2/full_connect_b", 1, {96, 0, 0, 0}, 2315136},
    {"RNN32/fc2/full_connect_w", 2, {133, 96, 0, 0}, 2315520},
    {"RNN32/fc3/full_connect_b", 1, {4, 0, 0, 0}, 2366592},
    {"RNN32/fc3/full_connect_w", 2, {101, 4, 0, 0}, 2366608},
    {"RNN32/multi_rnn_cell/cell_0/lstm_cell/bias", 1, {512, 0, 0, 0}, 2368224},
    {"RNN32/multi_rnn_cell/cell_0/lstm_cell/kernel", 2, {256, 512, 0, 0}, 2370272},
    {"RNN64/fc2/full_connect_b", 1, {48, 0, 0, 0}, 2894560},
    {"RNN64/fc2/full_connect_w", 2, {69, 48, 0, 0}, 2894752},
    {"RNN64/fc3/full_connect_b", 1, {1, 0, 0, 0}, 2908000},
    {"RNN64/fc3/full_connect_w", 2, {53, 1, 0, 0}, 2908004},
    {"RNN64/multi_rnn_cell/cell_0/lstm_cell/bias", 1, {256, 0, 0, 0}, 2908216},
    {"RNN64/multi_rnn_cell/cell_0/lstm_cell/kernel", 2, {128, 256, 0, 0}, 2909240},
};

// ---------------------------------------------------------------- synthetic weights ---
// The trained blobs are absent from the reference (.MISSING_LARGE_BLOBS); every test and
// benchmark runs on this seeded generator (same function in oracle/ethcnn_np.py::synth_blob
// for the test side).
static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void synth_blob(uint64_t seed, double head_gain, float* blob) {
    for (int t = 0; t < kNumTensors; ++t) {
        const TensorDesc& d = kTensors[t];
        const size_t n = d.count();
        const uint64_t key = splitmix64(seed ^ (0xD6E8FEB86659FD93ull * (uint64_t)(t + 1)));
        double scale;
        if (d.rank == 1) {
            scale = 0.1;
        } else {
            double fan_in = 1.0;
            for (int i = 0; i + 1 < d.rank; ++i) fan_in *= (double)d.shape[i];
            scale = std::sqrt(3.0 / fan_in);
            if (std::strncmp(d.name, "h_fc2", 5) == 0 || std::strncmp(d.name, "y_conv", 6) == 0)
                scale = scale * head_gain;
        }
        float* out = blob + d.offset_bytes / 4;
        for (size_t i = 0; i < n; ++i) {
            const uint64_t h = splitmix64(key + (uint64_t)i);
            const double u = (double)(h >> 40);
            const double val = (u + 0.5) * (1.0 / 8388608.0) - 1.0;
            out[i] = (float)(val * scale);
        }
    }
}

// same generator for the LSTM table, tensor index t + 100 (oracle/ethcnn_lstm_np.py::synth_lstm_blob)
void synth_lstm_blob(uint64_t seed, double head_gain, float* blob) {
    for (int t = 0; t < kNumLstmTensors; ++t) {
        const TensorDesc& d = kLstmTensors[t];
        const size_t n = d.count();
        const uint64_t key = splitmix64(seed ^ (0xD6E8FEB86659FD93ull * (uint64_t)(t + 101)));
        double scale = 0.1;
        if (d.rank == 2) {
            scale = std::sqrt(3.0 / (double)d.shape[0]);
            if (std::strstr(d.name, "/fc") != nullptr) scale = scale * head_gain;
        }
        float* out = blob + d.offset_bytes / 4;
        for (size_t i = 0; i < n; ++i) {
            const uint64_t h = splitmix64(key + (uint64_t)i);
            const double u = (double)(h >> 40);
            const double val = (u + 0.5) * (1.0 / 8388608.0) - 1.0;
            out[i] = (float)(val * scale);
        }
    }
}

// ------------------------------------------------------------------- device packing ---
// MFMA v_mfma_f32_16x16x4_f32 operand layout: lane l holds A[i = l & 15][k = l >> 4].
// The trunk runs "transposed" (rows = output channels, columns = 16 units), so the
// weights are the A operand and the accumulator of one layer is directly the B operand
// of the next (see ethcnn_kernels.hip, k1_trunk).
void pack_trunk_fragments(const float* blob, float* w_out, float* b_out) {
    for (int br = 0; br < 3; ++br) {
        const float* W1 = blob + kOffConvW[br][0];  // [4][4][1][16]
        const float* W2 = blob + kOffConvW[br][1];  // [2][2][16][24]
        const float* W3 = blob + kOffConvW[br][2];  // [2][2][24][32]
        const float* B1 = blob + kOffConvB[br][0];
        const float* B2 = blob + kOffConvB[br][1];
        const float* B3 = blob + kOffConvB[br][2];
        float* w = w_out + (size_t)br * kTrunkWFrags * 64;
        float* b = b_out + (size_t)br * kTrunkBFrags * 64;
        for (int lane = 0; lane < 64; ++lane) {
            const int col = lane & 15, g = lane >> 4;
            for (int s = 0; s < 4; ++s) w[(0 + s) * 64 + lane] = W1[(g * 4 + s) * 16 + col];
            for (int t = 0; t < 2; ++t)
                for (int s2 = 0; s2 < 16; ++s2) {
                    const int q1 = s2 >> 2, r = s2 & 3, ci = 4 * g + r, co = 16 * t + col;
                    w[(4 + t * 16 + s2) * 64 + lane] = (co < 24) ? W2[(q1 * 16 + ci) * 24 + co] : 0.0f;
                }
            for (int t = 0; t < 2; ++t)
                for (int s = 0; s < 24; ++s) {
                    int q2, ci;
                    if (s < 16) {
                        q2 = s >> 2;
                        ci = 4 * g + (s & 3);
                    } else {
                        const int j = (s - 16) >> 2, r = (s - 16) & 3;
                        q2 = 2 * j + (g >> 1);
                        ci = 16 + 4 * (g & 1) + r;
                    }
                    w[(36 + t * 24 + s) * 64 + lane] = W3[(q2 * 24 + ci) * 32 + 16 * t + col];
                }
            for (int r = 0; r < 4; ++r) b[(0 + r) * 64 + lane] = B1[4 * g + r];
            for (int t = 0; t < 2; ++t)
                for (int r = 0; r < 4; ++r) {
                    const int co = 16 * t + 4 * g + r;
                    b[(4 + t * 4 + r) * 64 + lane] = (co < 24) ? B2[co] : 0.0f;
                    b[(12 + t * 4 + r) * 64 + lane] = B3[co];
                }
        }
    }
}

void pack_fc1(const float* blob, float* w_out, float* b_out) {
    for (int h = 0; h < 3; ++h) {
        const float* W = blob + kOffFc1W[h];
        const int n1 = kN1[h];
        for (int k = 0; k < kNFeat; ++k)
            std::memcpy(w_out + (size_t)k * kNVec + kO1[h], W + (size_t)k * n1, sizeof(float) * n1);
        std::memcpy(b_out + kO1[h], blob + kOffFc1B[h], sizeof(float) * n1);
    }
}

// W1 in the exact LDS image order of k_fc1: [column block][K chunk][LDS row][LDS column], so each
// LDS-DMA instruction of the kernel is a linear 1 KiB copy.  LDS position (p, c) of a chunk
// holds W1[chunk*bk + r][block*bn + cc] with the bank permutation of ethcnn_dense.hip:
//   bn % 32 == 0 : r = p,                 cc = c ^ (16 * ((p >> 2) & 1))   (column-group swap)
//   bn % 32 == 16: r = p ^ ((p >> 2) & 1), cc = c                           (row swap)
void pack_fc1_image(const float* w_cat, int bn, int bk, float* img) {
    const int nsplit = kNVec / bn, nk = kNFeat / bk;
    const bool colswz = (bn % 32 == 0);
    for (int nb = 0; nb < nsplit; ++nb)
        for (int kc = 0; kc < nk; ++kc)
            for (int p = 0; p < bk; ++p) {
                const int key = (p >> 2) & 1;
                const int r = colswz ? p : (p ^ key);
                float* dst = img + (((size_t)nb * nk + kc) * bk + p) * bn;
                const float* src = w_cat + (size_t)(kc * bk + r) * kNVec + nb * bn;
                for (int c = 0; c < bn; ++c) dst[c] = src[colswz ? (c ^ (key << 4)) : c];
            }
}

// W1 in MFMA B-operand order per 16-column tile (DeviceWeights::fc1_lane16): what lane (col, g) feeds the four MFMA steps of
// sub-chunk u is one float4
void pack_fc1_lane_image(const float* w_cat, float* img) {
    for (int t = 0; t < kNVec / 16; ++t)
        for (int u = 0; u < kNFeat / 16; ++u)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int col = lane & 15, g = lane >> 4;
                    *img++ = w_cat[(size_t)(16 * u + 4 * g + e) * kNVec + 16 * t + col];
                }
}

// ---- FC1 plan 2 (ethcnn_fc1_fast.hip): feature order and weight image of the fp16 x 2 form.
// The trunk wave of task T (ethcnn_trunk_task.h; lane = col + 16 g) ends up holding eight full quads of features in registers:
//   n = 0..3  conv2 position q2 = n, channels 4 g + e           k = OFF2 + slot(q2) * 24 + 4 g + e
//   n = 4, 5  conv2 channels 16..23 of positions 2 j, 2 j + 1   k = OFF2 + slot(2 j + (g >> 1)) * 24 + 16 + 4 (g & 1) + e   (j = n - 4)
//   n = 6, 7  conv3 channels 16 t + 4 g + e                     k = OFF3 + unit * 32 + 16 t + 4 g + e                       (t = n - 6)
// with slot(q2) = (2 by + (q2 >> 1)) * (2 NB) + 2 bx + (q2 & 1), unit = by * NB + bx.  Quads (2 p, 2 p + 1) form register pair p:
// its lanes g = 0, 1 / 2, 3 fill the two k halves of chunks 8 T + 2 p / 8 T + 2 p + 1.
int fast_feature_k(int chunk, int kh, int idx) {
    const int T = chunk >> 3, p = (chunk >> 1) & 3, g = 2 * (chunk & 1) + kh;
    const int n = 2 * p + (idx >> 2), e = idx & 3;
    int br, by, bx;
    if (T < 16) { br = 0; by = T >> 2; bx = T & 3; }
    else if (T < 20) { br = 1; by = ((T - 16) >> 1) & 1; bx = (T - 16) & 1; }
    else { br = 2; by = bx = 0; }
    const int nb = kNb[br];
    auto slot = [&](int q2) { return (2 * by + (q2 >> 1)) * (2 * nb) + 2 * bx + (q2 & 1); };
    if (n < 4) return kOff2[br] + slot(n) * 24 + 4 * g + e;
    if (n < 6) return kOff2[br] + slot(2 * (n - 4) + (g >> 1)) * 24 + 16 + 4 * (g & 1) + e;
    return kOff3[br] + (by * nb + bx) * 32 + 16 * (n - 6) + 4 * g + e;
}

void pack_fc1_fast_image(const float* w_cat, int plan, float scale_w, uint16_t* img) {
    const int np = fast_pieces(plan);
    for (int c = 0; c < kFastChunks; ++c)
        for (int t = 0; t < kNVec / 32; ++t) {
            uint16_t* rec = img + ((size_t)c * (kNVec / 32) + t) * np * 512;  // NP 1 KiB pieces
            for (int lane = 0; lane < 64; ++lane) {
                const int n = lane & 31, kh = lane >> 5;
                for (int idx = 0; idx < 8; ++idx) {
                    const float w = w_cat[(size_t)fast_feature_k(c, kh, idx) * kNVec + 32 * t + n];
                    uint16_t p[2];  // fp16 x 2, round to nearest even at both steps (ethcnn_spec.h::f16_rne)
                    const float ws = w * scale_w;  // exact: a power of two, no overflow / underflow by the choice of the scale
                    p[0] = f16_rne(ws);
                    p[1] = f16_rne(ws - f16_f32(p[0]));
                    for (int q = 0; q < np; ++q) rec[q * 512 + lane * 8 + idx] = p[q];
                }
            }
        }
}

// |input| <= 1 (All-Intra: v = x/255 - block mean, both in [0, 1]) -> per-channel bounds of conv1, conv2, conv3 outputs
// (leaky-ReLU never grows a magnitude); the features are the conv2 and conv3 outputs of the three branches
float fast_feature_bound(const float* blob) {
    double worst = 0.0;
    for (int br = 0; br < 3; ++br) {
        const float* W1 = blob + kOffConvW[br][0];  // [4][4][1][16]
        const float* W2 = blob + kOffConvW[br][1];  // [2][2][16][24]
        const float* W3 = blob + kOffConvW[br][2];  // [2][2][24][32]
        const float* B1 = blob + kOffConvB[br][0];
        const float* B2 = blob + kOffConvB[br][1];
        const float* B3 = blob + kOffConvB[br][2];
        double b1[16], b2[24], b3[32];
        for (int co = 0; co < 16; ++co) {
            double sacc = std::fabs((double)B1[co]);
            for (int t = 0; t < 16; ++t) sacc += std::fabs((double)W1[t * 16 + co]);
            b1[co] = sacc;
        }
        for (int co = 0; co < 24; ++co) {
            double sacc = std::fabs((double)B2[co]);
            for (int q = 0; q < 4; ++q)
                for (int ci = 0; ci < 16; ++ci) sacc += std::fabs((double)W2[(q * 16 + ci) * 24 + co]) * b1[ci];
            b2[co] = sacc;
            worst = std::max(worst, sacc);
        }
        for (int co = 0; co < 32; ++co) {
            double sacc = std::fabs((double)B3[co]);
            for (int q = 0; q < 4; ++q)
                for (int ci = 0; ci < 24; ++ci) sacc += std::fabs((double)W3[(q * 24 + ci) * 32 + co]) * b2[ci];
            b3[co] = sacc;
            worst = std::max(worst, sacc);
        }
        (void)b3;
    }
    return (float)(worst * 1.0001);
}

// ---- plan 3: trunk A operands as fp16 x 2 pieces (ethcnn_trunk_fast.hip)
static void f16x2(float x, uint16_t* hi, uint16_t* lo) {
    *hi = f16_rne(x);
    *lo = f16_rne(x - f16_f32(*hi));
}
static float pow2_scale(double maxabs) {  // the power of two that puts maxabs at <= 2^14 (fp16 overflows at 65504)
    return std::exp2f(14.0f - std::ceil(std::log2((float)maxabs)));
}
void pack_trunk_f16(const float* blob, float scale_a, uint16_t* w_out, float* c_out, Trunk16Scalars* sc) {
    for (int br = 0; br < 3; ++br) {
        const float* W1 = blob + kOffConvW[br][0];  // [4][4][1][16]
        const float* W2 = blob + kOffConvW[br][1];  // [2][2][16][24]
        const float* W3 = blob + kOffConvW[br][2];  // [2][2][24][32]
        const float* B1 = blob + kOffConvB[br][0];
        const float* B2 = blob + kOffConvB[br][1];
        const float* B3 = blob + kOffConvB[br][2];
        uint16_t* w = w_out + (size_t)br * kTrunk16Halves;
        float* c = c_out + (size_t)br * kTrunk16Consts;
        double m1 = 0, m2 = 0, m3 = 0, bound1 = 0;
        for (int i = 0; i < 16 * 16; ++i) m1 = std::max(m1, (double)std::fabs(W1[i]));
        for (int i = 0; i < 64 * 24; ++i) m2 = std::max(m2, (double)std::fabs(W2[i]));
        for (int i = 0; i < 96 * 32; ++i) m3 = std::max(m3, (double)std::fabs(W3[i]));
        double wsum[16];
        for (int co = 0; co < 16; ++co) {
            double sabs = std::fabs((double)B1[co]), ssum = 0;
            for (int t = 0; t < 16; ++t) { sabs += std::fabs((double)W1[t * 16 + co]); ssum += (double)W1[t * 16 + co]; }
            wsum[co] = ssum;
            bound1 = std::max(bound1, sabs);  // |input| <= 1
        }
        // conv1 weights are multiplied by pixel SUMS up to 255 (S), 1020 (M), 255 / 15 (L digits): exact products need no headroom,
        // only the pieces must stay finite.  They are the pieces of  w * (c255 / pool^2 * S1)  (one rounding of the weight): the
        // accumulator then holds conv1's output in the activation scale S1 without an affine step per value (ethcnn_trunk_fast.hip);
        // |w| c255 S1 <= bound1 S1 / 255 <= 2^14 / 255, so the x 16 copy of the L branch stays far below 65504
        const float s2w = pow2_scale(m2), s3w = pow2_scale(m3);
        const float S1 = pow2_scale(bound1 * 1.0001);
        const int pool = br == 0 ? 1 : (br == 1 ? 2 : 4);
        const float c255s = (1.0f / 255.0f) * (1.0f / (float)(pool * pool));
        const float k1 = c255s * S1;  // S1 a power of two: exact
        sc->C1[br] = k1;
        (void)m1;
        sc->U2[br] = scale_a / (S1 * s2w);
        sc->U3[br] = 1.0f / s3w;
        for (int lane = 0; lane < 64; ++lane) {
            const int row = lane & 15, kb = lane >> 4;
            for (int i = 0; i < 4; ++i) {  // conv1: k = 4 kb + i = (ky = kb, kx = i)
                const float v = W1[(kb * 4 + i) * 16 + row] * k1;
                uint16_t hi, lo, hi16, lo16;
                f16x2(v, &hi, &lo);
                f16x2(v * 16.0f, &hi16, &lo16);
                w[(0 * 64 + lane) * 4 + i] = hi;
                w[(1 * 64 + lane) * 4 + i] = lo;
                w[(2 * 64 + lane) * 4 + i] = hi16;
                w[(3 * 64 + lane) * 4 + i] = lo16;
            }
            for (int t = 0; t < 2; ++t)
                for (int st = 0; st < 2; ++st)
                    for (int i = 0; i < 8; ++i) {  // conv2: k step st covers patches 2 st, 2 st + 1; ci = 4 kb + (i & 3)
                        const int q1 = 2 * st + (i >> 2), ci = 4 * kb + (i & 3), co = 16 * t + row;
                        const float v = co < 24 ? W2[(q1 * 16 + ci) * 24 + co] * s2w : 0.0f;
                        uint16_t hi, lo;
                        f16x2(v, &hi, &lo);
                        w[kTrunk16Conv2At + (((t * 2 + st) * 2 + 0) * 64 + lane) * 8 + i] = hi;
                        w[kTrunk16Conv2At + (((t * 2 + st) * 2 + 1) * 64 + lane) * 8 + i] = lo;
                    }
            for (int t = 0; t < 2; ++t)
                for (int st = 0; st < 3; ++st)
                    for (int i = 0; i < 8; ++i) {
                        int q2, ci;
                        if (st < 2) { q2 = 2 * st + (i >> 2); ci = 4 * kb + (i & 3); }                          // channels 0..15 of two positions
                        else { q2 = 2 * (i >> 2) + (kb >> 1); ci = 16 + 4 * (kb & 1) + (i & 3); }                // channels 16..23, the packed quads
                        const float v = W3[(q2 * 24 + ci) * 32 + 16 * t + row] * s3w;
                        uint16_t hi, lo;
                        f16x2(v, &hi, &lo);
                        w[kTrunk16Conv3At + (((t * 3 + st) * 2 + 0) * 64 + lane) * 8 + i] = hi;
                        w[kTrunk16Conv3At + (((t * 3 + st) * 2 + 1) * 64 + lane) * 8 + i] = lo;
                    }
            // constants of C-layout rows 4 g + r (g = kb here: lane = col + 16 g)
            for (int r = 0; r < 4; ++r) {
                const int ch = 4 * kb + r;
                c[(0 + r) * 64 + lane] = (float)(-(double)S1 * wsum[ch]);
                c[(4 + r) * 64 + lane] = S1 * B1[ch];
                for (int t = 0; t < 2; ++t) {
                    const int co = 16 * t + ch;
                    c[(8 + t * 4 + r) * 64 + lane] = co < 24 ? scale_a * B2[co] : 0.0f;
                    c[(16 + t * 4 + r) * 64 + lane] = scale_a * B3[co];
                }
            }
        }
    }
}

// ---- plan 3: FC2 / FC3 A operands as fp16 x 2 pieces (ethcnn_heads_fast.hip; layout: ethcnn_spec.h)
static void f16x2r(float x, uint16_t* hi, uint16_t* lo) {  // scaled residual: x = hi + lo * 2^-11 (ethcnn_heads_fast.hip::split8r)
    *hi = f16_rne(x);
    *lo = f16_rne((x - f16_f32(*hi)) * 2048.0f);
}
// see ethcnn_spec.h.  floor of a two-piece value x (scaled units): nothing beyond 2^-24 |x| while the residual (<= 2^-12 |x|) is a normal
// fp16 number (>= 2^-14), else the subnormal spacing's half
static double floor_plain(double x) { return std::fabs(x) < 0.25 ? std::ldexp(1.0, -25) : 0.0; }
static double floor_resid(double x) { return std::fabs(x) < std::ldexp(1.0, -13) ? std::ldexp(1.0, -36) : 0.0; }  // residual x 2^11
FastGuard fast_plan_floor_bound(const float* blob, int plan, bool heads16) {
    FastGuard g{};
    const double FLa = std::ldexp(1.0, -25), FLr = std::ldexp(1.0, -36);
    const double Fb = fast_feature_bound(blob);
    g.feature_bound = Fb;
    const double Sa = std::exp2(14.0 - std::ceil(std::log2(Fb)));  // ethcnn_model.cpp::ensure_fast_weights
    const double da = FLa / Sa;                                     // floor of a stored feature piece pair, value units
    if (plan == 3) {  // the trunk of plan 3 (pack_trunk_f16): conv1 on exact pixel sums, conv2 / conv3 with split activations and weights
        for (int br = 0; br < 3; ++br) {
            const float* W1 = blob + kOffConvW[br][0];
            const float* W2 = blob + kOffConvW[br][1];
            const float* W3 = blob + kOffConvW[br][2];
            const float* B1 = blob + kOffConvB[br][0];
            const float* B2 = blob + kOffConvB[br][1];
            double b1[16], b2[24], e1[16], e2[24], m2 = 0, m3 = 0, bound1 = 0;
            for (int i = 0; i < 64 * 24; ++i) m2 = std::max(m2, (double)std::fabs(W2[i]));
            for (int i = 0; i < 96 * 32; ++i) m3 = std::max(m3, (double)std::fabs(W3[i]));
            for (int co = 0; co < 16; ++co) {
                b1[co] = std::fabs((double)B1[co]);
                for (int t = 0; t < 16; ++t) b1[co] += std::fabs((double)W1[t * 16 + co]);
                bound1 = std::max(bound1, b1[co]);
            }
            const double S1 = pow2_scale(bound1 * 1.0001), s2w = pow2_scale(m2), s3w = pow2_scale(m3);
            const int pool = br == 0 ? 1 : (br == 1 ? 2 : 4);
            const double k1 = (1.0 / 255.0) / (pool * pool) * S1, psum = 255.0 * pool * pool;  // weight pieces of w k1 times pixel sums <= psum
            for (int co = 0; co < 16; ++co) {
                double e = 0;
                for (int t = 0; t < 16; ++t) e += floor_plain((double)W1[t * 16 + co] * k1) * psum;
                e1[co] = e / S1;
            }
            const double d1 = FLa / S1;  // conv1 outputs as conv2's split B operand
            for (int co = 0; co < 24; ++co) {
                double e = 0, b = std::fabs((double)B2[co]);
                for (int q = 0; q < 4; ++q)
                    for (int ci = 0; ci < 16; ++ci) {
                        const double w = W2[(q * 16 + ci) * 24 + co];
                        e += std::fabs(w) * (e1[ci] + d1) + b1[ci] * floor_plain(w * s2w) / s2w;
                        b += std::fabs(w) * b1[ci];
                    }
                e2[co] = e;
                b2[co] = b;
                g.feat_err = std::max(g.feat_err, e);
            }
            for (int co = 0; co < 32; ++co) {
                double e = 0;
                for (int q = 0; q < 4; ++q)
                    for (int ci = 0; ci < 24; ++ci) {
                        const double w = W3[(q * 24 + ci) * 32 + co];
                        e += std::fabs(w) * (e2[ci] + da) + b2[ci] * floor_plain(w * s3w) / s3w;  // conv2 outputs = the stored feature pieces
                    }
                g.feat_err = std::max(g.feat_err, e);
            }
        }
    }
    // FC1 (plans 2 and 3): features as pieces at Sa, W1 as pieces at Sw (one global scale: max |W1| over the three heads)
    double wmax = 0;
    for (int h = 0; h < 3; ++h)
        for (size_t i = 0; i < (size_t)kNFeat * kN1[h]; ++i) wmax = std::max(wmax, (double)std::fabs(blob[kOffFc1W[h] + i]));
    g.w1_max = wmax;
    const double Sw = std::exp2(14.0 - std::ceil(std::log2(wmax)));
    for (int h = 0; h < 3; ++h) {
        const int n1 = kN1[h], n2 = kN2[h], n3 = kN3[h];
        const float* W1 = blob + kOffFc1W[h];
        const float* B1 = blob + kOffFc1B[h];
        const float* W2 = blob + kOffFc2W[h];
        const float* B2 = blob + kOffFc2B[h];
        const float* W3 = blob + kOffFc3W[h];
        std::vector<double> eh1(n1), bh1(n1), eh2(n2), bh2(n2);
        double worst1 = 0, worst2 = 0, m2 = 0, m3 = 0;
        for (int n = 0; n < n1; ++n) {
            double sabs = 0, fl = 0;
            for (int k = 0; k < kNFeat; ++k) {
                const double w = W1[(size_t)k * n1 + n];
                sabs += std::fabs(w);
                fl += floor_plain(w * Sw);
            }
            eh1[n] = sabs * (g.feat_err + da) + Fb * fl / Sw;
            bh1[n] = std::fabs((double)B1[n]) + Fb * sabs;
            worst1 = std::max(worst1, bh1[n]);
            g.h1_err = std::max(g.h1_err, eh1[n]);
        }
        for (int i = 0; i < n1 * n2; ++i) m2 = std::max(m2, (double)std::fabs(W2[i]));
        for (int i = 0; i < n2 * n3; ++i) m3 = std::max(m3, (double)std::fabs(W3[i]));
        for (int m = 0; m < n2; ++m) {
            bh2[m] = std::fabs((double)B2[m]) + std::fabs((double)W2[(size_t)n1 * n2 + m]);
            for (int n = 0; n < n1; ++n) bh2[m] += std::fabs((double)W2[(size_t)n * n2 + m]) * bh1[n];
            worst2 = std::max(worst2, bh2[m]);
        }
        const bool h16 = plan == 3 && heads16 && worst1 > 0 && worst2 > 0 && m2 > 0 && m3 > 0;
        const double S1 = h16 ? pow2_scale(worst1 * 1.0001) : 1, S2 = h16 ? pow2_scale(worst2 * 1.0001) : 1, sw2 = h16 ? pow2_scale(m2) : 1, sw3 = h16 ? pow2_scale(m3) : 1;
        for (int m = 0; m < n2; ++m) {
            double e = 0;
            for (int n = 0; n < n1; ++n) {
                const double w = W2[(size_t)n * n2 + m];
                e += std::fabs(w) * (eh1[n] + (h16 ? FLr / S1 : 0.0)) + (h16 ? bh1[n] * floor_resid(w * sw2) / sw2 : 0.0);
            }
            eh2[m] = e;
        }
        for (int o = 0; o < n3; ++o) {
            double e = 0;
            for (int m = 0; m < n2; ++m) {
                const double w = W3[(size_t)m * n3 + o];
                e += std::fabs(w) * (eh2[m] + (h16 ? FLr / S2 : 0.0)) + (h16 ? bh2[m] * floor_resid(w * sw3) / sw3 : 0.0);
            }
            g.prob_err = std::max(g.prob_err, 0.25 * e);
        }
    }
    if (!std::isfinite(g.prob_err)) g.prob_err = INFINITY;
    return g;
}

bool pack_heads_f16(const float* blob, float feature_bound, uint16_t* img, Heads16Scalars* sc) {
    for (int h = 0; h < 3; ++h) {
        const int n1 = kN1[h], n2 = kN2[h], n3 = kN3[h], nt = n2 / 16;
        const float* W1 = blob + kOffFc1W[h];  // [2688][n1]
        const float* B1 = blob + kOffFc1B[h];
        const float* W2 = blob + kOffFc2W[h];  // [n1 + 1][n2], last row = qp
        const float* B2 = blob + kOffFc2B[h];
        const float* W3 = blob + kOffFc3W[h];  // [n2 + 1][n3]
        // |h1[n]| <= |b1[n]| + feature_bound * sum_k |W1[k][n]|  (leaky-ReLU never grows a magnitude);
        // |h2[m]| <= |b2[m]| + |W2[n1][m]| (qp / 51 <= 1) + sum_n |W2[n][m]| * bound1[n]
        std::vector<double> b1(n1);
        double worst1 = 0, worst2 = 0, m2 = 0, m3 = 0;
        for (int n = 0; n < n1; ++n) {
            double acc = 0;
            for (int k = 0; k < kNFeat; ++k) acc += std::fabs((double)W1[(size_t)k * n1 + n]);
            b1[n] = std::fabs((double)B1[n]) + (double)feature_bound * acc;
            worst1 = std::max(worst1, b1[n]);
        }
        for (int m = 0; m < n2; ++m) {
            double acc = std::fabs((double)B2[m]) + std::fabs((double)W2[(size_t)n1 * n2 + m]);
            for (int n = 0; n < n1; ++n) acc += std::fabs((double)W2[(size_t)n * n2 + m]) * b1[n];
            worst2 = std::max(worst2, acc);
        }
        for (int i = 0; i < n1 * n2; ++i) m2 = std::max(m2, (double)std::fabs(W2[i]));
        for (int i = 0; i < n2 * n3; ++i) m3 = std::max(m3, (double)std::fabs(W3[i]));
        if (!(worst1 > 0) || !(worst2 > 0) || !(m2 > 0) || !(m3 > 0) || !std::isfinite(worst1) || !std::isfinite(worst2) || !std::isfinite(m2) ||
            !std::isfinite(m3))
            return false;
        const float S1 = pow2_scale(worst1 * 1.0001), S2 = pow2_scale(worst2 * 1.0001), sw2 = pow2_scale(m2), sw3 = pow2_scale(m3);
        sc->S1[h] = S1;
        sc->U2[h] = 1.0f / (S1 * sw2);
        sc->S2[h] = S2;
        sc->U3[h] = 1.0f / (S2 * sw3);
        //  a bound that is finite as a double but beyond FLT_MAX makes pow2_scale return 0 and its reciprocal inf: every scale
        // and every product of scales the kernel multiplies by must be a normal float, else the exact heads are kept
        for (float v : {S1, S2, sw2, sw3, sc->U2[h], sc->U3[h]})
            if (!std::isnormal(v)) return false;
        uint16_t* f2 = img + heads16_fc2_at(h);
        for (int c = 0; c < n1 / 32; ++c)
            for (int j = 0; j < nt; ++j)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 8; ++i) {
                        const int row = lane & 15, kg = lane >> 4;
                        uint16_t hi, lo;
                        f16x2r(W2[(size_t)(32 * c + 8 * kg + i) * n2 + 16 * j + row] * sw2, &hi, &lo);
                        f2[(((size_t)(c * nt + j) * 2 + 0) * 64 + lane) * 8 + i] = hi;
                        f2[(((size_t)(c * nt + j) * 2 + 1) * 64 + lane) * 8 + i] = lo;
                    }
        uint16_t* f3 = img + heads16_fc3_at(h);
        for (int p = 0; p < heads16_fc3_steps(h); ++p)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    const int row = lane & 15, kg = lane >> 4, tile = 2 * p + (i >> 2), k = 16 * tile + 4 * kg + (i & 3);
                    uint16_t hi, lo;
                    f16x2r((row < n3 && tile < nt) ? W3[(size_t)k * n3 + row] * sw3 : 0.0f, &hi, &lo);
                    f3[((size_t)(p * 2 + 0) * 64 + lane) * 8 + i] = hi;
                    f3[((size_t)(p * 2 + 1) * 64 + lane) * 8 + i] = lo;
                }
    }
    return true;
}

void pack_fc2_lane_image(const float* w2, int n1, int n2, float* img) {
    for (int j = 0; j < n2 / 16; ++j)
        for (int kc = 0; kc < n1 / 16; ++kc)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int col = lane & 15, g = lane >> 4;
                    *img++ = w2[(size_t)(16 * kc + 4 * g + e) * n2 + 16 * j + col];
                }
}

void pack_lstm_kernels(const float* blob, float* out) {
    for (int lv = 0; lv < 3; ++lv) {
        const int N = 64 << lv, NC = 2 * N / 16;
        const float* K = blob + kLstmKernelOff[lv];  // [2N][4N], gate order i, j, f, o
        float* o = out + kLstmPackOff[lv];
        for (int t = 0; t < N / 16; ++t)
            for (int q = 0; q < 4; ++q)
                for (int kc = 0; kc < NC; ++kc)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int col = lane & 15, g = lane >> 4;
                            *o++ = K[(size_t)(16 * kc + 4 * g + e) * (4 * N) + q * N + 16 * t + col];
                        }
        // fc2, rows 0 .. N-1 (the 5 efs rows stay where they are)
        const int N2 = 48 << lv;
        const float* W2 = blob + kLstmFc2Off[lv];
        float* o2 = out + kLstmPackFc2Off[lv];
        for (int j = 0; j < N2 / 16; ++j)
            for (int t = 0; t < N / 16; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const int col = lane & 15, g = lane >> 4;
                        *o2++ = W2[(size_t)(16 * t + 4 * g + r) * N2 + 16 * j + col];
                    }
    }
}

}  // namespace ethcnn
