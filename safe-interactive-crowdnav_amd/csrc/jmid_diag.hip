// libjmid_hip.so -- jmid_dbg_* single-kernel entry points (diagnostics flavour only).
#include "jmid_ctx.hpp"
#include "jmid_launch.hpp"

#ifdef JMID_DIAGNOSTICS
// ---------------------------------------------------------------------------------------------- diagnostics
// Single-op entry points used by the unit tests (host buffers only).
extern "C" {

// ---------------------------------------------------------------------------------------------- diagnostics
// Single-op entry points used by the unit tests (host buffers only).
int jmid_dbg_plan_chunks(int net_kind, int nhead, int lanes, int chunk_episodes, int E, int tokens_per_episode, int* sizes, int cap) {
    if (E <= 0 || tokens_per_episode <= 0 || nhead <= 0 || !sizes || cap <= 0) return JMID_EINVAL;
    jmid_ctx ctx;                      // host fields only: the planner reads net_kind, nhead, lanes, chunk_eps and the tuning
    ctx.net_kind = net_kind; ctx.nhead = nhead; ctx.lanes = lanes; ctx.chunk_eps = chunk_episodes;
    TuneScope tune_scope(&ctx.tune);
    const std::vector<int> plan = plan_chunks(&ctx, E, tokens_per_episode);
    for (size_t i = 0; i < plan.size() && (int)i < cap; ++i) sizes[i] = plan[i];
    return (int)plan.size();
}

// ... in arithmetic mode `precision` (the plan of a small batch depends on it: at most 2 560 tokens stay ONE chunk in JMID_PREC_F16MX)
int jmid_dbg_plan_chunks_mode(int net_kind, int nhead, int lanes, int chunk_episodes, int E, int tokens_per_episode, int precision, int* sizes, int cap) {
    if (E <= 0 || tokens_per_episode <= 0 || nhead <= 0 || !sizes || cap <= 0) return JMID_EINVAL;
    if (precision != JMID_PREC_F32 && precision != JMID_PREC_F16X3 && precision != JMID_PREC_F16X2 && precision != JMID_PREC_F16MX) return JMID_EINVAL;
    jmid_ctx ctx;                      // (model dimensions at their defaults: d_model 512)
    ctx.net_kind = net_kind; ctx.nhead = nhead; ctx.lanes = lanes; ctx.chunk_eps = chunk_episodes;
    ctx.mx = precision == JMID_PREC_F16MX;
    ctx.x2 = precision == JMID_PREC_F16X2 || ctx.mx;
    TuneScope tune_scope(&ctx.tune);
    const std::vector<int> plan = plan_chunks(&ctx, E, tokens_per_episode);
    for (size_t i = 0; i < plan.size() && (int)i < cap; ++i) sizes[i] = plan[i];
    return (int)plan.size();
}

int jmid_dbg_gemm(jmid_handle_t h, int M, int N, int K, const float* A, const float* Wt, const float* bias, int relu,
                  int precision, float* C) {
    if (!h || !A || !Wt || !C) return JMID_EINVAL;
    if (precision != JMID_PREC_F32 && precision != JMID_PREC_F16X3 && precision != JMID_PREC_F16X2 && precision != JMID_PREC_F16MX)
        return fail(h, JMID_EINVAL, "bad precision");
    h->mx = precision == JMID_PREC_F16MX;
    h->x2 = precision == JMID_PREC_F16X2 || h->mx;
    HIPCHK(h, hipSetDevice(h->device));
    TuneScope tune_scope(&h->tune);
    if (!h->range_flag) {
        HIPCHK(h, hipMalloc((void**)&h->range_flag, sizeof(int)));
        HIPCHK(h, hipMemsetAsync(h->range_flag, 0, sizeof(int), h->stream));
    }
    float *dA, *dW, *dB = nullptr, *dC;
    HIPCHK(h, hipMalloc((void**)&dA, (size_t)M * K * 4));
    HIPCHK(h, hipMalloc((void**)&dW, (size_t)N * K * 4));
    HIPCHK(h, hipMalloc((void**)&dC, (size_t)M * N * 4));
    HIPCHK(h, hipMemcpy(dA, A, (size_t)M * K * 4, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(dW, Wt, (size_t)N * K * 4, hipMemcpyHostToDevice));
    if (bias) {
        HIPCHK(h, hipMalloc((void**)&dB, (size_t)N * 4));
        HIPCHK(h, hipMemcpy(dB, bias, (size_t)N * 4, hipMemcpyHostToDevice));
    }
    int rc = 0;
    half_t *ah = nullptr, *al = nullptr, *wh = nullptr, *wl = nullptr;
    jmid_ctx::W8Image w8img;
    if (precision == JMID_PREC_F32) {
        GemmArgs g{};
        g.A = dA; g.lda = K; g.W = dW; g.ldw = K; g.bias = dB; g.C = dC; g.ldc = N; g.M = M; g.N = N; g.K = K;
        rc = relu ? run_gemm<EPI_BIAS_RELU>(h, KC_GEMM_QKV, g) : run_gemm<EPI_BIAS>(h, KC_GEMM_QKV, g);
    } else {
        const size_t pa = blk_plane_elems(M, K) * 2, pw = blk_plane_elems(N, K) * 2;
        HIPCHK(h, hipMalloc((void**)&ah, pa));
        HIPCHK(h, hipMalloc((void**)&al, pa));
        HIPCHK(h, hipMalloc((void**)&wh, pw));
        HIPCHK(h, hipMalloc((void**)&wl, pw));
        for (auto pr : {std::make_pair(ah, pa), std::make_pair(al, pa), std::make_pair(wh, pw), std::make_pair(wl, pw)})
            HIPCHK(h, hipMemsetAsync(pr.first, 0, pr.second, h->stream));
        hipLaunchKernelGGL(split_planes_blocked_kernel, dim3(512), dim3(256), 0, h->stream, dA, ah, al, M, K,
                           h->range_flag, 1.0f);
        hipLaunchKernelGGL(split_planes_blocked_kernel, dim3(512), dim3(256), 0, h->stream, dW, wh, wl, N, K,
                           h->range_flag, kWScale);
        GemmHArgs g{};
        g.Ahi = ah; g.Alo = al; g.Whi = wh; g.Wlo = wl; g.bias = dB; g.C = dC; g.ldc = N;
        g.M = M; g.N = N; g.K = K;
        if (h->mx && N % 32 == 0 && K % 64 == 0) {
            if (int rc8 = make_w8(h, dW, N, K, &w8img)) return rc8;
            g.W8 = w8img.p;
        }
        rc = relu ? run_gemm_h<EPI_BIAS_RELU, OUT_F32>(h, KC_GEMM_QKV, g) : run_gemm_h<EPI_BIAS, OUT_F32>(h, KC_GEMM_QKV, g);
    }
    if (!rc) {
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(h, JMID_EHIP, hipGetErrorString(e));
    }
    if (!rc) HIPCHK(h, hipMemcpy(C, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dW); hipFree(dC);
    if (dB) hipFree(dB);
    for (half_t* p : {ah, al, wh, wl})
        if (p) hipFree(p);
    if (w8img.p) hipFree(w8img.p);
    return rc;
}

int jmid_dbg_attention(jmid_handle_t h, int nseq, int S, const float* QKV, int precision, float* OUT) {
    if (!h || !QKV || !OUT) return JMID_EINVAL;
    if (precision != JMID_PREC_F32 && precision != JMID_PREC_F16X3 && precision != JMID_PREC_F16X2 && precision != JMID_PREC_F16MX)
        return fail(h, JMID_EINVAL, "bad precision");
    h->mx = precision == JMID_PREC_F16MX;
    h->x2 = precision == JMID_PREC_F16X2 || h->mx;
    HIPCHK(h, hipSetDevice(h->device));
    TuneScope tune_scope(&h->tune);
    if (!h->range_flag) {
        HIPCHK(h, hipMalloc((void**)&h->range_flag, sizeof(int)));
        HIPCHK(h, hipMemsetAsync(h->range_flag, 0, sizeof(int), h->stream));
    }
    const size_t Mt = (size_t)nseq * S;
    const int d = h->d, hd = h->d / h->nhead;
    float *dQ, *dO;
    HIPCHK(h, hipMalloc((void**)&dQ, Mt * 3 * d * 4));
    HIPCHK(h, hipMalloc((void**)&dO, Mt * d * 4));
    HIPCHK(h, hipMemcpy(dQ, QKV, Mt * 3 * d * 4, hipMemcpyHostToDevice));
    int rc = 0;
    std::vector<half_t*> tmp;
    if (precision == JMID_PREC_F32) {
        AttnArgs aa{dQ, dO, S, d, h->nhead, 1.0f / sqrtf((float)hd), nullptr, nullptr};
        ProfScope ps(h, KC_ATTN);
        hipError_t e = launch_attn_f32(aa, nseq, hd, h->stream);
        if (e != hipSuccess) rc = fail(h, JMID_EHIP, hipGetErrorString(e));
    } else {
        const int Spad = vt_spad(S);
        const size_t vt = (size_t)nseq * d * Spad;
        half_t* b[8];
        const size_t sz[8] = {Mt * d, Mt * d, Mt * d, Mt * d, vt, vt, blk_plane_elems(Mt, d), blk_plane_elems(Mt, d)};
        for (int i = 0; i < 8; ++i) {
            HIPCHK(h, hipMalloc((void**)&b[i], sz[i] * sizeof(half_t)));
            HIPCHK(h, hipMemsetAsync(b[i], 0, sz[i] * sizeof(half_t), h->stream));
            tmp.push_back(b[i]);
        }
        hipLaunchKernelGGL(qkv_to_planes_kernel, dim3(512), dim3(256), 0, h->stream, dQ, b[0], b[1], b[2], b[3], b[4],
                           b[5], Mt, d, hd, S, Spad, 1.4426950408889634f / sqrtf((float)hd));
        int ns = 1;
        float *opart = nullptr, *mlpart = nullptr;
        if (hd == 128) ns = attn_pick_nsplit(((S + 127) / 128) * h->nhead * nseq, S);
        if (ns > 1) {
            HIPCHK(h, hipMalloc((void**)&opart, (size_t)ns * Mt * d * 4));
            HIPCHK(h, hipMalloc((void**)&mlpart, (size_t)ns * Mt * h->nhead * 2 * 4));
            tmp.push_back(reinterpret_cast<half_t*>(opart));
            tmp.push_back(reinterpret_cast<half_t*>(mlpart));
        }
        AttnHArgs aa{b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], S, Spad, d, h->nhead, 1.0f / sqrtf((float)hd),
                     h->range_flag, ns, opart, mlpart, h->x2};
        {
            ProfScope ps(h, KC_ATTN);
            hipError_t e = launch_attn_f16x3(aa, nseq, hd, h->stream);
            if (e != hipSuccess) rc = fail(h, JMID_EHIP, hipGetErrorString(e));
        }
        hipLaunchKernelGGL(merge_planes_kernel, dim3(512), dim3(256), 0, h->stream, b[6], b[7], dO, (int)Mt, d);
    }
    if (!rc) {
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(h, JMID_EHIP, hipGetErrorString(e));
    }
    if (!rc) HIPCHK(h, hipMemcpy(OUT, dO, Mt * d * 4, hipMemcpyDeviceToHost));
    hipFree(dQ); hipFree(dO);
    for (half_t* p : tmp) hipFree(p);
    return rc;
}

int jmid_dbg_gemm_ln_mx(jmid_handle_t h, int M, int K, const float* A, const float* Wt, const float* bias, const float* gamma,
                        const float* beta, float* X, int fused) {
    // X <- LayerNorm(X + A . Wt^T + bias) in JMID_PREC_F16MX at d_model 512 with the second-generation kernels
    // (gemm_ln2_mx.hpp): fused = 1 the row-complete kernel, 0 the GEMM + add_ln2 pair.  X comes back as hi + bf8(lo).
    if (!h || !A || !Wt || !bias || !gamma || !beta || !X || M <= 0 || K % 64 != 0) return JMID_EINVAL;
    constexpr int N = GLN_BN;
    HIPCHK(h, hipSetDevice(h->device));
    TuneScope tune_scope(&h->tune);
    h->mx = 1;
    h->x2 = 1;
    if (!h->range_flag) {
        HIPCHK(h, hipMalloc((void**)&h->range_flag, sizeof(int)));
        HIPCHK(h, hipMemsetAsync(h->range_flag, 0, sizeof(int), h->stream));
    }
    std::vector<void*> tmp;
    auto dalloc = [&](size_t bytes, const void* host) -> void* {
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        tmp.push_back(p);
        if (host) (void)hipMemcpy(p, host, bytes, hipMemcpyHostToDevice);
        else (void)hipMemsetAsync(p, 0, bytes, h->stream);
        return p;
    };
    float* dA = (float*)dalloc((size_t)M * K * 4, A);
    float* dW = (float*)dalloc((size_t)N * K * 4, Wt);
    float* dB = (float*)dalloc(N * 4, bias);
    float* dG = (float*)dalloc(N * 4, gamma);
    float* dT = (float*)dalloc(N * 4, beta);
    float* dX = (float*)dalloc((size_t)M * N * 4, X);
    float* dY = (float*)dalloc((size_t)(M + 63) / 64 * 64 * N * 4, nullptr);
    const size_t pa = blk_plane_elems(M, K) * 2, pw = blk_plane_elems(N, K) * 2, px = blk_plane_elems(M, N) * 2;
    half_t* ah = (half_t*)dalloc(pa, nullptr);
    half_t* al = (half_t*)dalloc(pa, nullptr);
    half_t* wh = (half_t*)dalloc(pw, nullptr);
    half_t* wl = (half_t*)dalloc(pw, nullptr);
    half_t* w16h = (half_t*)dalloc((size_t)N * K * 2, nullptr);
    half_t* w16l = (half_t*)dalloc((size_t)N * K * 2, nullptr);
    half_t* xh = (half_t*)dalloc(px, nullptr);
    unsigned char* xl8 = (unsigned char*)dalloc(px, nullptr);
    for (void* p : tmp)
        if (!p) return fail(h, JMID_ENOMEM, "jmid_dbg_gemm_ln_mx: allocation failed");
    hipLaunchKernelGGL(split_planes_blocked_kernel, dim3(512), dim3(256), 0, h->stream, dA, ah, al, M, K, h->range_flag, 1.0f);
    hipLaunchKernelGGL(split_planes_blocked_kernel, dim3(512), dim3(256), 0, h->stream, dW, wh, wl, N, K, h->range_flag, kWScale);
    hipLaunchKernelGGL(split_planes_k16_kernel, dim3(256), dim3(256), 0, h->stream, dW, w16h, w16l, N, K);
    hipLaunchKernelGGL(split_planes_lo8_kernel, dim3(512), dim3(256), 0, h->stream, dX, xh, xl8, M);
    jmid_ctx::W8Image img;
    if (int rc = make_w8(h, dW, N, K, &img)) return rc;
    tmp.push_back(img.p);
    int rc = 0;
    if (fused == 1) {
        GemmLn2Args g2{ah, w16h, img.p, dB, dG, dT, xh, xl8, M, K, 1e-5f, h->range_flag, 0};
        hipError_t e = launch_gemm_ln2_mx(g2, h->stream);
        if (e != hipSuccess) rc = fail(h, JMID_EHIP, hipGetErrorString(e));
    } else {
        GemmHArgs g{};
        g.Ahi = ah; g.Alo = al; g.Whi = wh; g.Wlo = wl; g.W8 = img.p; g.bias = dB; g.C = dY; g.ldc = N; g.M = M; g.N = N; g.K = K;
        if (fused == 3) {        // the small-launch kernel with the statistics exchange (gemm_small.hpp, OUT_LNX)
            unsigned long long* xs = (unsigned long long*)dalloc(kLnxWords * sizeof(unsigned), nullptr);
            if (!xs || !small_lnx_fits(M, K)) return fail(h, JMID_EINVAL, "jmid_dbg_gemm_ln_mx: shape does not take the small kernel with the statistics exchange");
            g.ln_gamma = dG; g.ln_beta = dT; g.ln_xh = xh; g.ln_xl = nullptr; g.ln_xl8 = xl8; g.ln_xchg = xs; g.ln_eps = 1e-5f; g.ln_no_lo = 0;
            (void)hipMemsetAsync(h->range_flag, 0, sizeof(int), h->stream);
            rc = run_gemm_lnx_small(h, KC_GEMM_OUT, g);
            if (!rc) {
                int flag = 0;
                (void)hipMemcpyAsync(&flag, h->range_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream);
                (void)hipStreamSynchronize(h->stream);
                if (flag & 2) rc = fail(h, JMID_EHIP, "jmid_dbg_gemm_ln_mx: a workgroup gave up waiting for its row tile's statistics");
            }
        } else
        {
        rc = run_gemm_h<EPI_BIAS, OUT_F32>(h, KC_GEMM_OUT, g);
        if (!rc) rc = run_add_ln(h, nullptr, dY, dG, dT, M, N, xh, reinterpret_cast<half_t*>(xl8), true, 0);
        }
    }
    if (!rc) {
        hipLaunchKernelGGL(merge_planes_lo8_kernel, dim3(512), dim3(256), 0, h->stream, xh, xl8, dX, M);
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(h, JMID_EHIP, hipGetErrorString(e));
    }
    if (!rc) HIPCHK(h, hipMemcpy(X, dX, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    for (void* p : tmp) hipFree(p);
    return rc;
}

int jmid_dbg_add_layernorm(jmid_handle_t h, int M, int d, float* X, const float* Y, const float* gamma,
                           const float* beta) {
    if (!h || !X || !Y || !gamma || !beta) return JMID_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    TuneScope tune_scope(&h->tune);
    float *dX, *dY, *dG, *dB;
    HIPCHK(h, hipMalloc((void**)&dX, (size_t)M * d * 4));
    HIPCHK(h, hipMalloc((void**)&dY, (size_t)M * d * 4));
    HIPCHK(h, hipMalloc((void**)&dG, (size_t)d * 4));
    HIPCHK(h, hipMalloc((void**)&dB, (size_t)d * 4));
    HIPCHK(h, hipMemcpy(dX, X, (size_t)M * d * 4, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(dY, Y, (size_t)M * d * 4, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(dG, gamma, (size_t)d * 4, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(dB, beta, (size_t)d * 4, hipMemcpyHostToDevice));
    int rc = run_add_ln(h, dX, dY, dG, dB, M, d);
    if (!rc) {
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(h, JMID_EHIP, hipGetErrorString(e));
    }
    if (!rc) HIPCHK(h, hipMemcpy(X, dX, (size_t)M * d * 4, hipMemcpyDeviceToHost));
    hipFree(dX); hipFree(dY); hipFree(dG); hipFree(dB);
    return rc;
}

}  // extern "C"
#endif  // JMID_DIAGNOSTICS

