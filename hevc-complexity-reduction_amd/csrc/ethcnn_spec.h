// ethcnn_spec.h -- shapes, checkpoint layout and packed-weight layouts shared by the host
// code and the HIP kernels of libethcnn.so.
//
// Network: /root/reference/HM-16.5_Test_AI/bin/net_CNN.py:103-195.
// Checkpoint tensor table: the .index files next to it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>

#include "../../include/ethcnn.h"

namespace ethcnn {

// Development knobs (A/B switches, forced-steal test modes; scripts/README.md lists them) are read from the environment ONLY by the
// experiments build of the library (make exp: -DETHCNN_EXPERIMENTS -> lib_exp/libethcnn.so, what the tests of those paths and the
// A/B scripts load through ETHCNN_LIB).  The shipped library ignores them: it reads ETHCNN_DEVICE(S), ETHCNN_FC1_PLAN,
// ETHCNN_HOST_THREADS, ETHCNN_LOCAL_WORKERS, ETHCNN_NUMA_BIND and nothing else.
#ifdef ETHCNN_EXPERIMENTS
inline const char* dev_env(const char* name) { return std::getenv(name); }
#else
inline const char* dev_env(const char*) { return nullptr; }
#endif

constexpr int kCtu = 64;
constexpr int kNOut = 21;
constexpr int kNFeat = 2688;
constexpr int kNVec = 448;   // FC1 outputs: 64 | 128 | 256
constexpr int kNFc2 = 336;   // FC2 outputs: 48 | 96 | 192
constexpr int kSubBatch = 1024;
constexpr size_t kBlobFloats = 1288210;
// Largest pass (CTUs) the kernels address: they use 32-bit BYTE offsets into the per-pass buffers (trunk: group image
// offset grp * kNFeat * 64; FC1: buffer resource of M * kNVec * 4 bytes; heads: 4 * (ctu * kNVec + ..); gate_chunk: u < 2^24).
// ethcnn_create clamps ethcnn_options.max_ctus_per_pass to this; a longer sequence simply takes more passes.
constexpr int kMaxCtusPerPass = 131072;
static_assert((long long)(kMaxCtusPerPass / 16) * kNFeat * 16 * 4 < (1ll << 31), "trunk: group image byte offset must fit int32");
static_assert((long long)kMaxCtusPerPass * kNVec * 4 < (1ll << 31), "FC1 / heads: h1 byte offsets must fit int32");
static_assert((long long)kMaxCtusPerPass * kNOut * 4 < (1ll << 31), "heads / gate: probability byte offsets must fit int32");
static_assert(2 * kMaxCtusPerPass < (1 << 24), "gate_chunk: r0 + ctu must be exact in float");

// ---- FC1 plan 2 ("fast": split operands on the 16-bit matrix pipe, ethcnn_fc1_fast.hip; plan 3 uses the same FC1).
//   every fp32 feature / weight, scaled by a power of two, as TWO fp16 pieces, a 2^s = a0 + a1 to 2^-24 relative
//   (two 11-bit significands, round to nearest even); three products (a0 w0, a1 w0, a0 w1)
//   (plan 1 of round 4 -- three bf16 pieces, six products -- was removed in round 5: slower and no more accurate)
// Features in v_mfma_f32_32x32x16_f16 A-operand order (NP = 2 pieces):
//   featb[pair of groups = 32 CTUs][chunk of 16 k: 168][piece: NP][1 KiB = [k half: 2][row: 32][8 x 16 bit]].
// Which feature sits in (chunk, k half, slot) is fast_feature_k below: the order in which the trunk's registers hold them.
constexpr int kFastChunks = kNFeat / 16;                     // 168 K chunks of 16
constexpr int fast_pieces(int /*plan*/) { return 2; }
constexpr int fast_pair_bytes(int plan) { return kFastChunks * fast_pieces(plan) * 1024; }  // 336 KiB per 32 CTUs
constexpr int kFastPairBytes = fast_pair_bytes(2);
static_assert((long long)(kMaxCtusPerPass / 32) * kFastPairBytes < (1ll << 31), "trunk / FC1 fast plans: pair image byte offset must fit int32");
// feature index held by slot `idx` (0..7) of k half `kh` of chunk `c`: chunk = 8 T + 2 p + (g >> 1), kh = g & 1 for trunk task T
// (unit position inside the group: 16 S, 4 M, 1 L), register pair p of the task and MFMA k-group g; slots 0..3 / 4..7 = the two quads of the pair
int fast_feature_k(int chunk, int kh, int idx);
// plan 2: a bound no feature of an All-Intra CTU can exceed (|input| <= 1 after mean removal; per-channel bounds pushed through
// the three conv layers of every branch) -- the feature scale 2^s is chosen from it so that no fp16 piece can overflow
float fast_feature_bound(const float* blob);
// ---- plan 3: the trunk's three conv layers on the 16-bit matrix pipe as well (fp16 x 2 splits; ethcnn_trunk_fast.hip).
// Per branch, A operands as fp16 pieces in MFMA order (halves): [conv1: 4 fragments x 64 lanes x 4 = piece 0, piece 1, 16 x piece 0,
// 16 x piece 1 (the last two for the L branch's high pixel-sum digits)][conv2: (t, s, piece) 8 fragments x 64 x 8][conv3: (t, s,
// piece) 12 fragments x 64 x 8]; per-lane constants (floats): [24 slots x 64 lanes]: conv1 -S1 Wsum (4), S1 b1 (4), conv2 sa b2
// (t, r: 8), conv3 sa b3 (t, r: 8); scalars per branch: C1 = c255 2^-p S1 -- the factor conv1's weight pieces CARRY (pieces of fl(w C1): the
// accumulator, started at S1 (b1 - mean Wsum), is conv1's output in the activation scale S1; the kernel does not multiply by it) --,
// U2 = sa / (S1 2^sw2), U3 = 2^-sw3.
constexpr int kTrunk16Halves = 4 * 64 * 4 + 8 * 64 * 8 + 12 * 64 * 8;  // 11,264 per branch
constexpr int kTrunk16Conv2At = 4 * 64 * 4, kTrunk16Conv3At = kTrunk16Conv2At + 8 * 64 * 8;
constexpr int kTrunk16Consts = 24 * 64;
struct Trunk16Scalars { float C1[3], U2[3], U3[3]; };
void pack_trunk_f16(const float* blob, float scale_a, uint16_t* w_out /*[3][kTrunk16Halves]*/, float* c_out /*[3][kTrunk16Consts]*/, Trunk16Scalars* sc);

// fp32 <-> IEEE binary16 on the host (round to nearest even, subnormals kept: what v_cvt_pk_f16_f32 / v_cvt_f32_f16 do on the device)
inline uint16_t f16_rne(float x) {
    uint32_t u;
    __builtin_memcpy(&u, &x, 4);
    const uint32_t sign = (u >> 16) & 0x8000u, e = (u >> 23) & 0xffu;
    uint32_t m = u & 0x7fffffu;
    if (e == 0xffu) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0u));
    const int E = (int)e - 127 + 15;
    if (E >= 31) return (uint16_t)(sign | 0x7c00u);
    if (E <= 0) {
        if (E < -10) return (uint16_t)sign;
        m |= 0x800000u;
        const int shift = 14 - E;  // 14 .. 24
        uint32_t half = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (half & 1u))) ++half;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)E << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) ++half;
    return (uint16_t)(sign | half);
}
inline float f16_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t u;
    if (e == 0) {
        const float f = (float)m * 5.9604644775390625e-08f;  // m 2^-24, exact
        __builtin_memcpy(&u, &f, 4);
        u |= sign;
    } else if (e == 31) {
        u = sign | 0x7f800000u | (m << 13);
    } else {
        u = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

// branches in feature order S, M, L (net_CNN.py:150 concat order)
enum Branch { kS = 0, kM = 1, kL = 2 };

struct TensorDesc {
    const char* name;
    int rank;
    int shape[4];
    size_t offset_bytes;  // into the TF-V2 .data payload
    size_t count() const {
        size_t n = 1;
        for (int i = 0; i < rank; ++i) n *= (size_t)shape[i];
        return n;
    }
};
constexpr int kNumTensors = 36;
extern const TensorDesc kTensors[kNumTensors];  // bundle (sorted-key) order
// ETH-LSTM checkpoints (HM-16.5_Test_LDP/bin/model_LDP_200000_qp*.dat.index): 18 tensors
constexpr int kNumLstmTensors = 18;
extern const TensorDesc kLstmTensors[kNumLstmTensors];
constexpr size_t kLstmBlobFloats = 760078;  // 3,040,312-byte .data payload
// LSTMCell kernels [2N][4N] re-laid for the gate-per-wave cell kernel (ethcnn_lstm.hip), appended to the blob on the device:
// level LV (N = 64 / 128 / 256): [tile N/16][gate 4][chunk 2N/16][lane 64][e 4] = K[16 chunk + 4 g + e][gate N + 16 tile + col],
// lane = col + 16 g -- one contiguous 1 KB run per (tile, gate, chunk), a dwordx4 per lane
constexpr int kLstmKernelOff[3] = {727310, 592568, 54496};    // float offsets of the three kernels inside the blob
constexpr int kLstmPackOff[3] = {0, 32768, 163840};           // float offsets inside the packed area
// ... followed by the fc2 matrices' first N rows in MFMA-operand order (the one-launch LDP frame kernel's heads, ethcnn_lstm.hip):
// level LV: [tile N2/16][chunk N/16][lane 64][r 4] = W2[16 chunk + 4 g + r][16 tile + col]
constexpr int kLstmFc2Off[3] = {723688, 578880, 192};         // float offsets of the fc2 matrices [N + 5][N2] inside the blob
constexpr int kLstmPackFc2Off[3] = {688128, 691200, 703488};  // 64 x 48, 128 x 96, 256 x 192 floats
constexpr size_t kLstmPackFloats = 752640;

// float offsets into the blob ------------------------------------------------------------
// conv variables are unnamed: L = Variable.._5, M = _6.._11, S = _12.._17 (creation order,
// net_CNN.py:126-141)
constexpr size_t kOffConvW[3][3] = {{13504 / 4, 14592 / 4, 20832 / 4},
                                    {51904 / 4, 52992 / 4, 1088 / 4},
                                    {0 / 4, 33248 / 4, 39488 / 4}};
constexpr size_t kOffConvB[3][3] = {{14528 / 4, 20736 / 4, 33120 / 4},
                                    {52928 / 4, 59136 / 4, 13376 / 4},
                                    {1024 / 4, 39392 / 4, 51776 / 4}};
// heads in output order 64, 32, 16
constexpr int kN1[3] = {64, 128, 256}, kN2[3] = {48, 96, 192}, kN3[3] = {1, 4, 16};
constexpr int kO1[3] = {0, 64, 192}, kO2[3] = {0, 48, 144}, kO3[3] = {0, 1, 5};
constexpr size_t kOffFc1W[3] = {4189792 / 4, 2813280 / 4, 60256 / 4};
constexpr size_t kOffFc1B[3] = {4189536 / 4, 2812768 / 4, 59232 / 4};
constexpr size_t kOffFc2W[3] = {5126176 / 4, 5076448 / 4, 4878688 / 4};
constexpr size_t kOffFc2B[3] = {5125984 / 4, 5076064 / 4, 4877920 / 4};
constexpr size_t kOffFc3W[3] = {5152644 / 4, 5151088 / 4, 5138720 / 4};
constexpr size_t kOffFc3B[3] = {5152640 / 4, 5151072 / 4, 5138656 / 4};

// ---- plan 3, round 5: the heads (FC2 + FC3) on the 16-bit matrix pipe as well (ethcnn_heads_fast.hip), operands as fp16 x 2 splits of
// power-of-two scaled values like FC1's and the trunk's.  Per head h (n1 = 64 / 128 / 256, n2 = 48 / 96 / 192, n3 = 1 / 4 / 16), A operands
// of v_mfma_f32_16x16x32_f16 ("transposed": rows = output features, columns = CTUs), lane = row + 16 kg holds eight k values:
//   FC2 image [n1 / 32 chunks][n2 / 16 tiles][2 pieces][64 lanes][8]:  W2[32 c + 8 kg + i][16 j + row] * sw2
//   FC3 image [ceil(n2 / 32) steps][2 pieces][64 lanes][8]:            W3[16 (2 p + (i >> 2)) + 4 kg + (i & 3)][row] * sw3  (0 for row >= n3
//                                                                       or a tile beyond n2): the k order in which the FC2 accumulators
//                                                                       of tiles 2 p, 2 p + 1 sit in a lane's registers
// The scales are powers of two from GUARANTEED bounds (fast_feature_bound pushed through |W1| and |W2|), so no piece can overflow.
constexpr int heads16_n1(int h) { return 64 << h; }
constexpr int heads16_n2(int h) { return 48 << h; }
constexpr int heads16_fc2_halves(int h) { return heads16_n1(h) * heads16_n2(h) * 2; }
constexpr int heads16_fc3_steps(int h) { return (heads16_n2(h) / 16 + 1) / 2; }
constexpr int heads16_fc3_halves(int h) { return heads16_fc3_steps(h) * 2 * 512; }
constexpr int heads16_fc2_at(int h) { return h == 0 ? 0 : heads16_fc2_at(h - 1) + heads16_fc2_halves(h - 1); }  // (usable in device code)
constexpr int heads16_fc3_at(int h) { return h == 0 ? heads16_fc2_at(3) : heads16_fc3_at(h - 1) + heads16_fc3_halves(h - 1); }
constexpr int kHeads16Halves = heads16_fc3_at(3);
struct Heads16Scalars { float S1[3], U2[3], S2[3], U3[3]; };  // h1 scale, 1 / (S1 sw2), h2 scale, 1 / (S2 sw3)
// false: a bound is not finite / zero (degenerate weights) -- the caller keeps the exact heads
bool pack_heads_f16(const float* blob, float feature_bound, uint16_t* img_out /*[kHeads16Halves]*/, Heads16Scalars* sc);

// ---- load-time accuracy guard of the 16-bit plans (ethcnn_weights.cpp).
// A value carried as two fp16 pieces of its power-of-two scaled form x is exact to 2^-24 |x| -- the class of fp32's own roundings -- as
// long as the residual piece is a NORMAL fp16 number; below that it is exact only to an ABSOLUTE floor (2^-25 in scaled units; 2^-36
// for the heads' 2^11-scaled residuals).  The scales come from guaranteed bounds, so the floor bites exactly when a bound is LOOSE
// (an outlier weight, heavy tails: DESIGN.md section 5 records the case that first showed it).  fast_plan_floor_bound pushes every
// floor of the plan -- activations at their worst (always at the floor), weights as they are -- through |W| of the layers behind it,
// all errors aligned, down to the probabilities (sigmoid' <= 1/4): a rigorous upper bound on what the plan's floors can move a
// probability by, computed from the weights alone.  Plans whose bound exceeds kFastGuardTol are refused at first use.
struct FastGuard {
    double prob_err;       // the bound, worst head
    double feat_err;       // plan 3: floor error of a feature (value units); plan 2: 0 (exact trunk)
    double h1_err;         // worst FC1 output
    double feature_bound;  // fast_feature_bound: the guaranteed |feature| bound the activation scale comes from
    double w1_max;         // max |W1|: the FC1 weight scale
};
constexpr double kFastGuardTol = 2.5e-5;  // a quarter of the north star's 1e-4: the rest is left to summation-order noise (measured ~7e-6)
FastGuard fast_plan_floor_bound(const float* blob, int plan, bool heads16);

// feature-vector map
constexpr int kOff3[3] = {0, 512, 640};       // conv3 S, M, L
constexpr int kOff2[3] = {672, 2208, 2592};   // conv2 S, M, L
constexpr int kNb[3] = {4, 2, 1};             // units per CTU side

// ---- device-side packed weights ---------------------------------------------------------
// Trunk (k1): per branch, MFMA A-operand fragments, one float per lane per k-step:
//   [0..3]    conv1  A1[s]       = W1[ky=g][kx=s][co=col]
//   [4..35]   conv2  A2[t][s2]   s2 = 4*q1 + r, ci = 4g + r, co = 16t + col (0 if co >= 24)
//   [36..83]  conv3  A3[t][s]    s<16: q2 = s>>2, r = s&3, ci = 4g+r
//                                s>=16: j = (s-16)>>2, r = (s-16)&3, q2 = 2j + (g>>1), ci = 16 + 4(g&1) + r
// with lane = col + 16 g.  Bias fragments (value for C-layout row 4g + r):
//   [0..3] conv1, [4..11] conv2 [t][r] (0 for co >= 24), [12..19] conv3 [t][r]
constexpr int kTrunkWFrags = 84, kTrunkBFrags = 20;
// LDS of a trunk block (floats): [84 x 64 weight fragments][8 x 64 x 4: gather exchange of the single-launch pass's L blocks]
// [resi only: the preprocessed values of the branch's 256 / 1021 / 4081 possible pixel sums, ethcnn_trunk_task.h]
constexpr int kTrunkResiTabAt = kTrunkWFrags * 64 + 8 * 64 * 4;
constexpr int kTrunkResiLds = kTrunkResiTabAt + 4096;

struct DeviceWeights {
    float* trunk_w = nullptr;  // [3][84][64]
    float* trunk_b = nullptr;  // [3][20][64]
    // FC1 weights ([2688][448], columns = 64 | 128 | 256) packed per column block in the exact
    // LDS image order of k_fc1 (ethcnn_dense.hip): [448/BN][2688/BK][BK][BN], bank-permuted
    float* fc1_img112 = nullptr;  // BN 112, BK 16
    float* fc1_img64 = nullptr;   // BN 64,  BK 32
    float* fc1_img32 = nullptr;   // BN 32 and BN 16: low-latency shapes for short row ranges (the image
    float* fc1_img16 = nullptr;   // depends on BN only: chunk rows are consecutive)
    // the same weights in MFMA-operand order per 16-column tile: [28 tiles][168 sub-chunks of 16 k][64 lanes][4]: lane (col, g)
    // holds W1[16 u + 4 g + e][16 t + col], e = 0..3 -- one dwordx4 load per lane per sub-chunk, no LDS (single-launch pass)
    float* fc1_lane16 = nullptr;
    // FC1 plan 2 (and 3): W1 as two fp16 pieces in the 32x32x16 MFMA's B-operand order, [168 chunks][14 column tiles of 32][2 pieces]
    // [1 KiB = [k half][32 columns][8]], k order = fast_feature_k (pack_fc1_fast_image)
    uint16_t* fc1_fast = nullptr;
    uint16_t* trunk16_w = nullptr;  // plan 3: [3][kTrunk16Halves]
    float* trunk16_c = nullptr;     //         [3][kTrunk16Consts]
    Trunk16Scalars trunk16_s{};
    uint16_t* heads16_w = nullptr;  // plan 3: FC2 / FC3 A operands as fp16 x 2 pieces, [kHeads16Halves] (null: exact heads)
    Heads16Scalars heads16_s{};
    float fast_scale_a = 1.0f, fast_scale_w = 1.0f;  // plan 2: powers of two applied to features / W1 before the fp16 split
    float* fc1_b = nullptr;    // [448]
    float* fc2_w[3] = {nullptr, nullptr, nullptr};  // [n1+1][n2] (last row = qp row)
    float* fc2_b[3] = {nullptr, nullptr, nullptr};
    // rows 0 .. n1-1 of fc2_w in MFMA-operand order: [n2/16 tiles][n1/16 chunks][64 lanes][4]: lane (col, g) holds
    // W2[16 kc + 4 g + e][16 j + col] (single-launch pass: one dwordx4 load per lane per tile and chunk, no LDS)
    float* fc2_lane[3] = {nullptr, nullptr, nullptr};
    float* fc3_w[3] = {nullptr, nullptr, nullptr};  // [n2+1][n3]
    float* fc3_b[3] = {nullptr, nullptr, nullptr};
};

// host-side packing (ethcnn_weights.cpp)
void pack_trunk_fragments(const float* blob, float* w_out /*[3][84][64]*/, float* b_out /*[3][20][64]*/);
void pack_fc1(const float* blob, float* w_out /*[2688][448]*/, float* b_out /*[448]*/);
void pack_fc1_image(const float* w_cat /*[2688][448]*/, int bn, int bk, float* img_out /*[2688*448]*/);
void pack_fc1_lane_image(const float* w_cat /*[2688][448]*/, float* img_out /*[2688*448]*/);
void pack_fc1_fast_image(const float* w_cat /*[2688][448]*/, int plan, float scale_w, uint16_t* img_out /*[2688*448*NP]*/);
void pack_fc2_lane_image(const float* w2 /*[n1+1][n2]*/, int n1, int n2, float* img_out /*[n1*n2]*/);
void synth_blob(uint64_t seed, double head_gain, float* blob_out /*[kBlobFloats]*/);
void synth_lstm_blob(uint64_t seed, double head_gain, float* blob_out /*[kLstmBlobFloats]*/);
void pack_lstm_kernels(const float* blob, float* out /*[kLstmPackFloats]*/);

// TF-V2 checkpoint bundle reader (tf_ckpt_v2.cpp).  Returns 0 or a negative ETHCNN_ERR_*;
// on error `err` holds the message.
using CkptEntry = ::ethcnn_ckpt_entry;
int ckpt_read_index(const char* index_path, CkptEntry* entries, int cap, int* n_out, char* err, size_t errcap);
int ckpt_load_blob(const char* prefix, float* blob_out /*[kBlobFloats]*/, char* err, size_t errcap);
int ckpt_load_table(const char* prefix, const TensorDesc* table, int ntensors, float* blob_out, char* err, size_t errcap);
uint32_t crc32c(const void* data, size_t n);
uint32_t crc32c_mask(uint32_t crc);

// Thr_info.txt / model name (ethcnn_io.cpp)
int parse_thr_info(const char* path, float* thr_l1_lower, float* thr_l2_lower, char* err, size_t errcap);

}  // namespace ethcnn
