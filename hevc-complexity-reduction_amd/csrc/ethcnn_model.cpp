// ethcnn_model.cpp -- weights of the context: upload, the 16-bit plans' images, the LSTM bundle; parity-test introspection
#include "ethcnn_ctx.h"

// -------------------------------------------------------------------- weights -------
static int upload_weights(ethcnn_ctx* c) {
    HIPCHK(c, hipSetDevice(c->device));
    // one arena: trunk_w | trunk_b | fc1 image (BN 112) | fc1_b | fc2 w,b x3 | fc3 w,b x3 | fc1 image (BN 64)
    // (each 64-float aligned)
    std::vector<size_t> sizes = {(size_t)3 * kTrunkWFrags * 64, (size_t)3 * kTrunkBFrags * 64,
                                 (size_t)kNFeat * kNVec, (size_t)kNVec};
    for (int h = 0; h < 3; ++h) { sizes.push_back((size_t)(kN1[h] + 1) * kN2[h]); sizes.push_back((size_t)kN2[h]); }
    for (int h = 0; h < 3; ++h) { sizes.push_back((size_t)(kN2[h] + 1) * kN3[h]); sizes.push_back((size_t)kN3[h]); }
    sizes.push_back((size_t)kNFeat * kNVec);  // [16] fc1 image BN 64
    sizes.push_back((size_t)kNFeat * kNVec);  // [17] fc1 image BN 32
    sizes.push_back((size_t)kNFeat * kNVec);  // [18] fc1 image BN 16
    sizes.push_back((size_t)kNFeat * kNVec);  // [19] fc1 in MFMA-operand (lane) order
    for (int h = 0; h < 3; ++h) sizes.push_back((size_t)kN1[h] * kN2[h]);  // [20..22] fc2 in MFMA-operand order
    std::vector<size_t> offs;
    size_t total = 0;
    for (size_t s : sizes) { offs.push_back(total); total += (s + 63) / 64 * 64; }
    std::vector<float> host(total, 0.0f);
    const float* blob = c->blob.data();
    pack_trunk_fragments(blob, host.data() + offs[0], host.data() + offs[1]);
    {
        std::vector<float> wcat((size_t)kNFeat * kNVec);
        pack_fc1(blob, wcat.data(), host.data() + offs[3]);
        pack_fc1_image(wcat.data(), 112, 16, host.data() + offs[2]);
        pack_fc1_image(wcat.data(), 64, 32, host.data() + offs[16]);
        pack_fc1_image(wcat.data(), 32, 32, host.data() + offs[17]);
        pack_fc1_image(wcat.data(), 16, 32, host.data() + offs[18]);
        pack_fc1_lane_image(wcat.data(), host.data() + offs[19]);
    }
    for (int h = 0; h < 3; ++h) {
        std::memcpy(host.data() + offs[4 + 2 * h], blob + kOffFc2W[h], sizes[4 + 2 * h] * 4);
        std::memcpy(host.data() + offs[5 + 2 * h], blob + kOffFc2B[h], sizes[5 + 2 * h] * 4);
        std::memcpy(host.data() + offs[10 + 2 * h], blob + kOffFc3W[h], sizes[10 + 2 * h] * 4);
        std::memcpy(host.data() + offs[11 + 2 * h], blob + kOffFc3B[h], sizes[11 + 2 * h] * 4);
        pack_fc2_lane_image(blob + kOffFc2W[h], kN1[h], kN2[h], host.data() + offs[20 + h]);
    }
    if (!c->dw_arena) HIPCHK(c, hipMalloc((void**)&c->dw_arena, total * 4));
    HIPCHK(c, hipDeviceSynchronize());  // no pass in flight (on any of the streams) may still read the old arena
    HIPCHK(c, hipMemcpy(c->dw_arena, host.data(), total * 4, hipMemcpyHostToDevice));
    DeviceWeights& d = c->dw;
    d.trunk_w = c->dw_arena + offs[0];
    d.trunk_b = c->dw_arena + offs[1];
    d.fc1_img112 = c->dw_arena + offs[2];
    d.fc1_img64 = c->dw_arena + offs[16];
    d.fc1_img32 = c->dw_arena + offs[17];
    d.fc1_img16 = c->dw_arena + offs[18];
    d.fc1_lane16 = c->dw_arena + offs[19];
    for (int h = 0; h < 3; ++h) d.fc2_lane[h] = c->dw_arena + offs[20 + h];
    d.fc1_b = c->dw_arena + offs[3];
    for (int h = 0; h < 3; ++h) {
        d.fc2_w[h] = c->dw_arena + offs[4 + 2 * h];
        d.fc2_b[h] = c->dw_arena + offs[5 + 2 * h];
        d.fc3_w[h] = c->dw_arena + offs[10 + 2 * h];
        d.fc3_b[h] = c->dw_arena + offs[11 + 2 * h];
    }
    d.fc1_fast = nullptr;  // the fast plans' images of W1 belong to the previous weights: repacked on the next such pass
    d.trunk16_w = nullptr;
    d.heads16_w = nullptr;
    d.trunk16_c = nullptr;
    c->guard_state[2] = c->guard_state[3] = 0;  // (the guard of the 16-bit plans is a function of the weights)
    ++c->weights_gen;
    c->have_weights = true;
    return ETHCNN_OK;
}

// Load-time accuracy guard of the 16-bit plans, two stages, cached per weight load (video_to_cu_depth.py:126-133 restores one of four
// checkpoints per run: any of them must either be safe under an opted-in plan or be refused -- never silently less accurate):
//   1. a-priori: ethcnn_spec.h::fast_plan_floor_bound, a RIGOROUS bound on what the plan's fp16 floors can move a probability by, from
//      the weights alone (every error aligned, every activation at its floor).  <= kFastGuardTol: accepted, nothing is run.
//   2. otherwise the worst case says nothing either way (it is pessimistic by the looseness of the very bounds it guards: plan 3 on
//      benign weights has a bound of 0.1 and a measured error of 7e-6), so the plan is MEASURED: a seeded calibration picture (flat,
//      low-contrast, gradient, noise, edge and texture macro tiles, 640 CTUs) through the exact plan and through the plan with the
//      loaded weights, gates open; max |dp| <= kFastGuardTol: accepted; else refused with both numbers.
static int build_fast_weights(ethcnn_ctx* c, int plan);
static int check_fast_plan(ethcnn_ctx* c, int plan) {
    if (plan != 2 && plan != 3) return set_err(c, ETHCNN_ERR_ARG, "plan must be 2 or 3, got %d", plan);
    if (!c->have_weights) return set_err(c, ETHCNN_ERR_NOWEIGHTS, "no weights loaded");
    if (c->guard_state[plan] == 0) {
        const FastGuard g = fast_plan_floor_bound(c->blob.data(), plan, /*heads16=*/true);
        c->guard_bound[plan] = g.prob_err;
        c->guard_info[plan] = g;
        c->guard_measured[plan] = -1.0;
        if (g.prob_err <= kFastGuardTol) {
            c->guard_state[plan] = 1;
        } else {
            int rc = build_fast_weights(c, plan);
            if (rc) return rc;
            double worst = 0.0;
            c->guard_state[plan] = 3;  // calibrating: the passes below are not guarded (they ARE the guard)
            rc = calibrate_fast_plan(c, plan, &worst);
            c->guard_state[plan] = 0;
            if (rc) return rc;
            c->guard_measured[plan] = worst;
            c->guard_state[plan] = (worst <= kFastGuardTol) ? 1 : 2;  // (a NaN compares false: refused)
        }
    }
    if (c->guard_state[plan] == 3) return ETHCNN_OK;
    if (c->guard_state[plan] != 1) {
        const FastGuard& g = c->guard_info[plan];
        return set_err(c, ETHCNN_ERR_PLAN_REFUSED,
                       "FC1 plan %d is refused for these weights: its fp16 x 2 pieces are exact to 2^-24 only while a value stays within ~2^12 of the "
                       "guaranteed bound its scale comes from, and these weights make the bounds too loose for that: on the calibration picture the plan "
                       "differs from the exact plan by %.3g on a probability (limit %.3g; a-priori worst case %.3g; guaranteed |feature| bound %.4g, "
                       "max |W1| %.4g): use plan %s",
                       plan, c->guard_measured[plan], kFastGuardTol, c->guard_bound[plan], g.feature_bound, g.w1_max, plan == 3 ? "2 or 0" : "0");
    }
    return ETHCNN_OK;
}
// the a-priori bound without a context or a device (host arithmetic on the blob only): what tests and tools print.  ETHCNN_OK: the
// bound alone accepts the plan; ETHCNN_ERR_PLAN_REFUSED: it does not (a context would go on to measure)
extern "C" int ethcnn_fast_plan_bound(const float* blob, size_t nfloats, int plan, double* prob_err_bound, double* feature_bound) {
    if (!blob || nfloats != kBlobFloats || (plan != 2 && plan != 3) || !prob_err_bound) return ETHCNN_ERR_ARG;
    const FastGuard g = fast_plan_floor_bound(blob, plan, /*heads16=*/true);
    *prob_err_bound = g.prob_err;
    if (feature_bound) *feature_bound = g.feature_bound;
    return g.prob_err <= kFastGuardTol ? ETHCNN_OK : ETHCNN_ERR_PLAN_REFUSED;
}
extern "C" int ethcnn_check_fc1_plan(ethcnn_ctx* c, int plan, double* apriori_bound, double* measured) {
    if (!c) return ETHCNN_ERR_ARG;
    if (apriori_bound) *apriori_bound = 0.0;
    if (measured) *measured = -1.0;
    if (plan == 0) return ETHCNN_OK;
    if (plan != 2 && plan != 3) return set_err(c, ETHCNN_ERR_ARG, "plan must be 0, 2 or 3, got %d", plan);
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = check_fast_plan(c, plan);
    if (rc == ETHCNN_OK || rc == ETHCNN_ERR_PLAN_REFUSED) {
        if (apriori_bound) *apriori_bound = c->guard_bound[plan];
        if (measured) *measured = c->guard_measured[plan];
    }
    return rc;
}

// plan 2: W1 as fp16 x 2 pieces in the MFMA's B-operand order (ethcnn_weights.cpp::pack_fc1_fast_image), once per weight load
int ensure_fast_weights(ethcnn_ctx* c, int plan) {
    const int rc = check_fast_plan(c, plan);  // (cached: one comparison per pass after the first)
    return rc ? rc : build_fast_weights(c, plan);
}
static int build_fast_weights(ethcnn_ctx* c, int plan) {
    if (plan == 3) {  // plan 3 = plan 2's FC1 + the trunk's convolutions as fp16 x 2 (ethcnn_trunk_fast.hip)
        int rc = build_fast_weights(c, 2);
        if (rc || c->dw.trunk16_w) return rc;
        const size_t wbytes = (size_t)3 * kTrunk16Halves * 2, cbytes = (size_t)3 * kTrunk16Consts * 4;
        std::vector<uint16_t> wimg((size_t)3 * kTrunk16Halves);
        std::vector<float> cimg((size_t)3 * kTrunk16Consts);
        pack_trunk_f16(c->blob.data(), c->dw.fast_scale_a, wimg.data(), cimg.data(), &c->dw.trunk16_s);
        if (!c->dw_trunk16) HIPCHK(c, hipMalloc((void**)&c->dw_trunk16, wbytes + cbytes));
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, hipMemcpy(c->dw_trunk16, wimg.data(), wbytes, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(reinterpret_cast<char*>(c->dw_trunk16) + wbytes, cimg.data(), cbytes, hipMemcpyHostToDevice));
        c->dw.trunk16_w = c->dw_trunk16;
        c->dw.trunk16_c = reinterpret_cast<float*>(reinterpret_cast<char*>(c->dw_trunk16) + wbytes);
        // ... and the heads' FC2 / FC3 (ethcnn_heads_fast.hip): scales from guaranteed bounds; degenerate weights (a zero / non-finite
        // bound) keep the exact heads
        std::vector<uint16_t> himg((size_t)kHeads16Halves);
        if (pack_heads_f16(c->blob.data(), fast_feature_bound(c->blob.data()), himg.data(), &c->dw.heads16_s)) {
            if (!c->dw_heads16) HIPCHK(c, hipMalloc((void**)&c->dw_heads16, (size_t)kHeads16Halves * 2));
            HIPCHK(c, hipMemcpy(c->dw_heads16, himg.data(), (size_t)kHeads16Halves * 2, hipMemcpyHostToDevice));
            c->dw.heads16_w = c->dw_heads16;
        }
        return 0;
    }
    if (c->dw.fc1_fast) return 0;
    const size_t n16 = (size_t)kNFeat * kNVec * fast_pieces(plan);
    std::vector<float> wcat((size_t)kNFeat * kNVec), b1(kNVec);
    pack_fc1(c->blob.data(), wcat.data(), b1.data());
    {   // the feature order of the plans is a table of the trunk's register order: it must be a permutation of 0 .. 2687
        std::vector<char> seen(kNFeat, 0);
        for (int ch = 0; ch < kFastChunks; ++ch)
            for (int s8 = 0; s8 < 16; ++s8) {
                const int k = fast_feature_k(ch, s8 >> 3, s8 & 7);
                if (k < 0 || k >= kNFeat || seen[k]) return set_err(c, ETHCNN_ERR_ARG, "internal: fast FC1 feature order is not a permutation (chunk %d)", ch);
                seen[k] = 1;
            }
    }
    float sw = 1.0f;
    {
        // powers of two that put the largest possible |feature| and the largest |weight| at <= 2^14 (fp16 overflows at 65504): the
        // feature bound is a guarantee derived from the conv weights (|input| <= 1), not an observation
        float wmax = 0.0f;
        for (float v : wcat) wmax = std::max(wmax, std::fabs(v));
        const float fmax = fast_feature_bound(c->blob.data());
        if (!(wmax > 0.0f) || !(fmax > 0.0f) || !std::isfinite(wmax) || !std::isfinite(fmax))
            return set_err(c, ETHCNN_ERR_ARG, "FC1 plan 2 needs finite, non-zero weights (max |W1| %g, feature bound %g)", (double)wmax, (double)fmax);
        sw = std::exp2f(14.0f - std::ceil(std::log2(wmax)));
        c->dw.fast_scale_w = sw;
        c->dw.fast_scale_a = std::exp2f(14.0f - std::ceil(std::log2(fmax)));
    }
    std::vector<uint16_t> img(n16);
    pack_fc1_fast_image(wcat.data(), plan, sw, img.data());
    if (!c->dw_fast) HIPCHK(c, hipMalloc((void**)&c->dw_fast, n16 * 2));
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(c->dw_fast, img.data(), n16 * 2, hipMemcpyHostToDevice));
    c->dw.fc1_fast = c->dw_fast;
    return 0;
}

extern "C" int ethcnn_load_blob(ethcnn_ctx* c, const float* blob, size_t nfloats) {
    if (!c || !blob) return ETHCNN_ERR_ARG;
    if (nfloats != kBlobFloats) return set_err(c, ETHCNN_ERR_ARG, "blob must hold %zu floats, got %zu", kBlobFloats, nfloats);
    c->blob.assign(blob, blob + nfloats);
    return upload_weights(c);
}

extern "C" int ethcnn_load_synthetic(ethcnn_ctx* c, uint64_t seed, double head_gain) {
    if (!c) return ETHCNN_ERR_ARG;
    c->blob.resize(kBlobFloats);
    synth_blob(seed, head_gain, c->blob.data());
    return upload_weights(c);
}

extern "C" int ethcnn_load_checkpoint(ethcnn_ctx* c, const char* prefix) {
    if (!c || !prefix) return ETHCNN_ERR_ARG;
    std::vector<float> blob(kBlobFloats);
    char err[400];
    const int rc = ckpt_load_blob(prefix, blob.data(), err, sizeof err);
    if (rc) return set_err(c, rc, "%s", err);
    c->blob.swap(blob);
    return upload_weights(c);
}

extern "C" int ethcnn_get_blob(const ethcnn_ctx* c, float* out, size_t nfloats) {
    if (!c || !out || nfloats != kBlobFloats || !c->have_weights) return ETHCNN_ERR_ARG;
    std::memcpy(out, c->blob.data(), nfloats * 4);
    return ETHCNN_OK;
}

// ------------------------------------------------ config #5: ETH-LSTM one step -------
static int upload_lstm(ethcnn_ctx* c) {
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_lstm) HIPCHK(c, hipMalloc((void**)&c->d_lstm, (kLstmBlobFloats + kLstmPackFloats) * sizeof(float)));
    std::vector<float> pack(kLstmPackFloats);  // the LSTMCell kernels in the cell kernel's load order, behind the blob
    pack_lstm_kernels(c->lstm_blob.data(), pack.data());
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpyAsync(c->d_lstm, c->lstm_blob.data(), kLstmBlobFloats * sizeof(float), hipMemcpyHostToDevice,
                             c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_lstm + kLstmBlobFloats, pack.data(), kLstmPackFloats * sizeof(float), hipMemcpyHostToDevice,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_lstm = true;
    return ETHCNN_OK;
}

extern "C" int ethcnn_load_lstm_blob(ethcnn_ctx* c, const float* blob, size_t nfloats) {
    if (!c || !blob) return ETHCNN_ERR_ARG;
    if (nfloats != kLstmBlobFloats)
        return set_err(c, ETHCNN_ERR_ARG, "LSTM blob must hold %zu floats, got %zu", kLstmBlobFloats, nfloats);
    c->lstm_blob.assign(blob, blob + nfloats);
    return upload_lstm(c);
}

extern "C" int ethcnn_load_lstm_synthetic(ethcnn_ctx* c, uint64_t seed, double head_gain) {
    if (!c) return ETHCNN_ERR_ARG;
    c->lstm_blob.resize(kLstmBlobFloats);
    synth_lstm_blob(seed, head_gain, c->lstm_blob.data());
    return upload_lstm(c);
}

extern "C" int ethcnn_load_lstm_checkpoint(ethcnn_ctx* c, const char* prefix) {
    if (!c || !prefix) return ETHCNN_ERR_ARG;
    std::vector<float> blob(kLstmBlobFloats);
    char err[400];
    const int rc = ckpt_load_table(prefix, kLstmTensors, kNumLstmTensors, blob.data(), err, sizeof err);
    if (rc) return set_err(c, rc, "%s", err);
    c->lstm_blob.swap(blob);
    return upload_lstm(c);
}

extern "C" int ethcnn_get_lstm_blob(const ethcnn_ctx* c, float* out, size_t nfloats) {
    if (!c || !out || nfloats != kLstmBlobFloats || !c->have_lstm) return ETHCNN_ERR_ARG;
    std::memcpy(out, c->lstm_blob.data(), nfloats * 4);
    return ETHCNN_OK;
}

extern "C" int ethcnn_set_debug_capture(ethcnn_ctx* c, int on) {
    if (!c) return ETHCNN_ERR_ARG;
    c->debug_capture = (on != 0);
    return ETHCNN_OK;
}

extern "C" int ethcnn_debug_fetch(ethcnn_ctx* c, int which, float* out, size_t nfloats) {
    if (!c || !out) return ETHCNN_ERR_ARG;
    if (which >= ETHCNN_DBG_FC2 && which <= ETHCNN_DBG_RAW_PROBS && !c->debug_capture)
        return set_err(c, ETHCNN_ERR_ARG, "debug_fetch(%d): call ethcnn_set_debug_capture(ctx, 1) before the pass", which);
    const float* src = nullptr;
    size_t per = 0;
    switch (which) {
        case ETHCNN_DBG_FEATURES: src = c->ws.feat; per = kNFeat; break;
        case ETHCNN_DBG_FC1: src = ws_view(c, c->last_parity).h1; per = kNVec; break;
        case ETHCNN_DBG_FC2: src = c->ws.h2; per = kNFc2; break;
        case ETHCNN_DBG_LOGITS: src = c->ws.logits; per = kNOut; break;
        case ETHCNN_DBG_RAW_PROBS: src = c->ws.raw; per = kNOut; break;
        default: return set_err(c, ETHCNN_ERR_ARG, "unknown debug tensor %d", which);
    }
    if (!src || nfloats > (size_t)c->last_n * per) return set_err(c, ETHCNN_ERR_ARG, "debug_fetch: last pass had %d CTUs", c->last_n);
    if (which != ETHCNN_DBG_FEATURES) return ethcnn_memcpy_d2h(c, out, src, nfloats * 4);
    if (c->last_fast) {
        // plans 2 / 3: the trunk left every feature as two fp16 pieces: add them back, (h0 + h1) / scale (plan 2: equal to the feature to
        // 2^-24 relative)
        const int plan = c->last_fast == 3 ? 2 : c->last_fast, np = fast_pieces(plan);  // (plan 3 writes plan 2's form)
        const size_t n = (nfloats + kNFeat - 1) / kNFeat, pairs = (n + 31) / 32;
        std::vector<uint16_t> rawb(pairs * (size_t)(fast_pair_bytes(plan) / 2));
        int rc = ethcnn_memcpy_d2h(c, rawb.data(), c->ws.featb, rawb.size() * 2);
        if (rc) return rc;
        auto hf = [](uint16_t h) { return f16_f32(h); };
        const float inv = 1.0f / c->dw.fast_scale_a;
        for (size_t row = 0; row * kNFeat < nfloats; ++row)
            for (int ch = 0; ch < kFastChunks; ++ch)
                for (int kh = 0; kh < 2; ++kh)
                    for (int idx = 0; idx < 8; ++idx) {
                        const size_t o = row * kNFeat + (size_t)fast_feature_k(ch, kh, idx);
                        if (o >= nfloats) continue;
                        const uint16_t* rec = rawb.data() + ((row / 32) * kFastChunks + ch) * np * 512 + (kh * 32 + row % 32) * 8 + idx;
                        out[o] = (hf(rec[0]) + hf(rec[512])) * inv;
                    }
        return ETHCNN_OK;
    }
    // features live as [group of 16 CTUs][k/4][16][4] (ethcnn_dense.hip); hand back [n][2688]
    const size_t n = (nfloats + kNFeat - 1) / kNFeat, groups = (n + 15) / 16;
    std::vector<float> rawf(groups * 16 * kNFeat);
    int rc = ethcnn_memcpy_d2h(c, rawf.data(), src, rawf.size() * 4);
    if (rc) return rc;
    for (size_t i = 0; i < nfloats; ++i) {
        const size_t row = i / kNFeat, k = i % kNFeat;
        out[i] = rawf[((row / 16) * (kNFeat / 4) + k / 4) * 64 + (row % 16) * 4 + (k % 4)];
    }
    return ETHCNN_OK;
}
