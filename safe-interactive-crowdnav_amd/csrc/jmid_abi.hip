// libjmid_hip.so -- the C ABI proper (include/jmid_hip.h): handle lifetime, the compute entry points, knobs, streams.
#include "jmid_ctx.hpp"

namespace jmid_host {

std::string& thread_error() {
    static thread_local std::string e;
    return e;
}

int fail(jmid_ctx* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    thread_error() = msg;
    return code;
}

// the two KDE launches on device buffers (pos [E, K, A, T, 2], bw [T] or null -> sel, logw); the ll / Y workspace is the handle's
int topk_on_device(jmid_ctx* h, int E, int A, int K, int T, int k, const float* pos, const float* bw, float* sel, float* logw) {
    const int d = 2 * A;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t y_bytes = kde_y_in_lds(A, K) ? 0 : up((size_t)E * T * K * d * 8);
    const size_t o_Y = up((size_t)E * T * K * 8), need = o_Y + y_bytes;
    if (need > h->kde_ws_bytes) {
        if (h->kde_ws) {
            HIPCHK(h, hipStreamSynchronize(h->stream));
            HIPCHK(h, hipFree(h->kde_ws));
            h->kde_ws = nullptr;
            h->kde_ws_bytes = 0;
        }
        if (hipMalloc((void**)&h->kde_ws, need) != hipSuccess) return fail(h, JMID_ENOMEM, "jmid_topk workspace allocation failed");
        h->kde_ws_bytes = need;
    }
    KdeArgs g{};
    g.E = E; g.A = A; g.K = K; g.T = T; g.k = k;
    g.ll = reinterpret_cast<double*>(h->kde_ws);
    g.Y = reinterpret_cast<double*>(h->kde_ws + o_Y);
    g.pos = pos; g.bw = bw; g.sel = sel; g.logw = logw;
    ProfScope ps(h, KC_TOPK);
    HIPCHK(h, launch_kde(g, h->stream));
    return 0;
}

}  // namespace jmid_host

// ================================================================================================ C ABI
extern "C" {

const char* jmid_version(void) {
#ifdef JMID_DIAGNOSTICS
    return "jmid_hip 0.7.0+diagnostics (gfx950; f32-mfma + f16x3 / f16x2 split-mfma + f16mx fp8-correction)";
#else
    return "jmid_hip 0.7.0 (gfx950; f32-mfma + f16x3 / f16x2 split-mfma + f16mx fp8-correction)";
#endif
}

int jmid_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* jmid_last_error(jmid_handle_t h) { return h ? h->err.c_str() : thread_error().c_str(); }

int jmid_create(jmid_handle_t* out, int device_id, int net_kind, int ctx_dim, int tf_layer, int nhead, int hist_len) {
    if (!out) return JMID_EINVAL;
    *out = nullptr;
    if (net_kind != JMID_NET_IMID && net_kind != JMID_NET_JMID) return fail(nullptr, JMID_EINVAL, "bad net_kind");
    if (ctx_dim < 32 || ctx_dim % 32 != 0 || ctx_dim > 512)
        return fail(nullptr, JMID_EINVAL, "ctx_dim must be a multiple of 32 in [32, 512]");
    if (tf_layer < 1 || tf_layer > 16) return fail(nullptr, JMID_EINVAL, "bad tf_layer");
    const int d = 2 * ctx_dim;
    if (nhead < 1 || d % nhead != 0) return fail(nullptr, JMID_EINVAL, "nhead must divide d_model");
    const int hd = d / nhead;
    if (hd != 16 && hd != 32 && hd != 64 && hd != 128)
        return fail(nullptr, JMID_EINVAL, "head_dim must be one of 16, 32, 64, 128");
    if (hist_len < 1 || hist_len > ENC_MAX_TH) return fail(nullptr, JMID_EINVAL, "hist_len out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, JMID_EHIP, "no HIP device available (libjmid_hip has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, JMID_EINVAL, "device_id out of range");
    jmid_ctx* h = new jmid_ctx();
    h->device = device_id;
    h->net_kind = net_kind;
    h->ctx_dim = ctx_dim;
    h->tf_layer = tf_layer;
    h->nhead = nhead;
    h->hist_len = hist_len;
    h->d = d;
    h->ff = 4 * ctx_dim;
    h->dmid = ctx_dim;
    h->dlow = ctx_dim / 2;
    h->H = ctx_dim / 2;
    h->hl = make_hyper_layout(h->d, h->dmid, h->dlow);
    register_shapes(h);
    // NON-BLOCKING streams: a blocking stream is implicitly ordered against the legacy null stream, so once ANYTHING in the process
    // (torch on its default stream, or this handle's own device-mode ordering events) has put work on the null stream, every launch on
    // the handle's stream pays for that coupling - one cfg2 call went from 10.1 to 13.0 ms after a single device-mode call on the handle
    // (tools/predict_probe.py).  The handle orders itself against the caller's stream EXPLICITLY (order_in / order_out: events), host-mode
    // calls synchronise the stream before they return, and nothing in the library uses the null stream.
    bool ok = hipSetDevice(device_id) == hipSuccess && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming) == hipSuccess;
    for (int l = 0; ok && l < jmid_ctx::kMaxLanes - 1; ++l)
        ok = hipStreamCreateWithFlags(&h->lane_stream[l], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&h->ev_join[l], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        delete h;
        return fail(nullptr, JMID_EHIP, "cannot create a HIP stream");
    }
    // compute units of this device (or of its partition): the kernels whose workgroups wait for each other launch only when all of
    // them are resident at once (gemm_small.hpp::small_lnx_fits)
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) h->tune.cus = cus;
    *out = h;
    return JMID_OK;
}

int jmid_destroy(jmid_handle_t h) {
    if (h) {
        if (h->pin) (void)hipHostFree(h->pin);
        if (h->io_dev) (void)hipFree(h->io_dev);
        h->pin = h->io_dev = nullptr;
    }
    if (!h) return JMID_OK;
    hipSetDevice(h->device);
    sync_lanes(h);
    drop_graphs(h);
    for (auto& kv : h->w) hipFree(kv.second.p);
    for (auto* m : {&h->wsplit, &h->w16})
        for (auto& kv : *m) {
            hipFree(kv.second.hi);
            hipFree(kv.second.lo);
        }
    for (auto& kv : h->w8) hipFree(kv.second.p);
    if (h->range_flag) hipFree(h->range_flag);
    if (h->ev_in) hipEventDestroy(h->ev_in);
    if (h->ev_out) hipEventDestroy(h->ev_out);
    for (float* p : {h->pe, h->Whyp, h->bhyp, h->thyp, h->attW1T, h->attW2T})
        if (p) hipFree(p);
    for (auto& l : h->lstmT)
        for (float* p : l)
            if (p) hipFree(p);
    if (h->arena) hipFree(h->arena);
    if (h->kde_ws) hipFree(h->kde_ws);
    for (int c = 0; c < KC_COUNT; ++c)
        for (auto& ev : h->prof_ev[c]) {
            hipEventDestroy(ev.a);
            hipEventDestroy(ev.b);
        }
    for (auto& ev : h->ev_pool) {
        hipEventDestroy(ev.a);
        hipEventDestroy(ev.b);
    }
    hipStreamDestroy(h->stream);
    for (int l = 0; l < jmid_ctx::kMaxLanes - 1; ++l) {
        hipStreamDestroy(h->lane_stream[l]);
        hipEventDestroy(h->ev_join[l]);
    }
    hipEventDestroy(h->ev_fork);
    delete h;
    return JMID_OK;
}

int jmid_denoise_ddpm(jmid_handle_t h, int E, int A, int K, int T, const float* x_T, const float* z, const float* ctx,
                      const float* p0, float dt, int precision, float* vel_out, float* pos_out, int mem) {
    if (!h) return JMID_EINVAL;
    if (!z) return fail(h, JMID_EINVAL, "null z");
    return run_network(h, E, A, K, T, x_T, ctx, p0, dt, precision, -1, vel_out, pos_out, nullptr, mem, z);
}

int jmid_encode(jmid_handle_t h, int n_agents, const float* x_st, const float* nbr_sum, const float* edge_mask,
                float* ctx_out, int mem) {
    if (!h) return JMID_EINVAL;
    if (!h->finalized) return fail(h, JMID_ENOWEIGHT, "jmid_finalize_weights has not been called");
    if (n_agents <= 0 || !x_st || !nbr_sum || !edge_mask || !ctx_out) return fail(h, JMID_EINVAL, "bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    TuneScope tune_scope(&h->tune);
    if (int rc = order_in(h, mem)) return rc;
    const int Th = h->hist_len, H = h->H;
    const size_t n = n_agents;
    const float *xs = x_st, *ns = nbr_sum, *em = edge_mask;
    float* co = ctx_out;
    if (mem == JMID_MEM_HOST) {
        Carver c0(nullptr);
        c0.take(n * Th * 6); c0.take(n * 2 * Th * 6); c0.take(n * 2); c0.take(n * 2 * H);
        if (int rc = ensure_arena(h, c0.off)) return rc;
        h->last_pos = nullptr;        // the staging buffers below overwrite the workspace the last positions live in
        Carver c(h->arena);
        float* dx = c.take(n * Th * 6);
        float* dn = c.take(n * 2 * Th * 6);
        float* de = c.take(n * 2);
        co = c.take(n * 2 * H);
        HIPCHK(h, hipMemcpyAsync(dx, x_st, n * Th * 6 * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(dn, nbr_sum, n * 2 * Th * 6 * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(de, edge_mask, n * 2 * sizeof(float), hipMemcpyHostToDevice, h->stream));
        xs = dx; ns = dn; em = de;
    }
    {
        ProfScope ps(h, KC_ENCODER);
        EncArgs ea{};
        ea.x_st = xs; ea.nbr_sum = ns; ea.edge_mask = em;
        ea.hist = LstmW{h->lstmT[0][0], h->lstmT[0][1], h->lstmT[0][2]};
        ea.edge[0] = LstmW{h->lstmT[1][0], h->lstmT[1][1], h->lstmT[1][2]};
        ea.edge[1] = LstmW{h->lstmT[2][0], h->lstmT[2][1], h->lstmT[2][2]};
        ea.W1T = h->attW1T; ea.W2T = h->attW2T; ea.v = W(h, "PEDESTRIAN/edge_influence_encoder.v.weight");
        ea.ctx = co; ea.n = n_agents; ea.Th = Th; ea.H = H;
        HIPCHK(h, launch_encoder(ea, h->stream));
    }
    if (mem == JMID_MEM_HOST) {
        HIPCHK(h, hipMemcpyAsync(ctx_out, co, n * 2 * H * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return order_out(h, mem);
}

int jmid_denoise(jmid_handle_t h, int E, int A, int K, int T, const float* x_T, const float* ctx, const float* p0,
                 float dt, int precision, float* vel_out, float* pos_out, int mem) {
    if (!h) return JMID_EINVAL;
    return run_network(h, E, A, K, T, x_T, ctx, p0, dt, precision, -1, vel_out, pos_out, nullptr, mem);
}

int jmid_net_eval(jmid_handle_t h, int E, int A, int K, int T, int step_idx, const float* x, const float* ctx,
                  int precision, float* e_out, int mem) {
    if (!h) return JMID_EINVAL;
    if (!e_out) return fail(h, JMID_EINVAL, "null e_out");
    if (int rc = check_ready(h)) return rc;
    if (step_idx < 0 || step_idx >= (int)h->beta.size()) return fail(h, JMID_EINVAL, "step_idx out of range");
    return run_network(h, E, A, K, T, x, ctx, nullptr, 0.f, precision, step_idx, nullptr, nullptr, e_out, mem);
}

int jmid_episode_metrics(jmid_handle_t h, int E, int A, int K, int T, const float* pos, const float* gt,
                         float* out, int mem) {
    if (!h || !pos || !gt || !out || E <= 0 || A <= 0 || K <= 0 || T <= 0) return fail(h, JMID_EINVAL, "bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    TuneScope tune_scope(&h->tune);
    if (int rc = order_in(h, mem)) return rc;
    const size_t np_ = (size_t)E * K * A * T * 2, ng = (size_t)E * A * T * 2;
    const float *dp = pos, *dg = gt;
    float* dout = out;
    if (mem == JMID_MEM_HOST) {
        Carver c0(nullptr);
        c0.take(np_); c0.take(ng); c0.take((size_t)E * 4);
        if (int rc = ensure_arena(h, c0.off)) return rc;
        h->last_pos = nullptr;        // (as in jmid_encode)
        Carver c(h->arena);
        float* a = c.take(np_);
        float* b = c.take(ng);
        dout = c.take((size_t)E * 4);
        HIPCHK(h, hipMemcpyAsync(a, pos, np_ * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(b, gt, ng * 4, hipMemcpyHostToDevice, h->stream));
        dp = a; dg = b;
    }
    if (int rc = launch_episode_metrics(h, dp, dg, dout, E, K, A, T)) return rc;
    if (mem == JMID_MEM_HOST) {
        HIPCHK(h, hipMemcpyAsync(out, dout, (size_t)E * 4 * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return order_out(h, mem);
}

int jmid_topk(jmid_handle_t h, int E, int A, int K, int T, int k, const float* pos, const float* bw, float* sel, float* logw,
              int mem) {
    if (!h || !sel || !logw || E <= 0 || A <= 0 || K <= 1 || T <= 0) return fail(h, JMID_EINVAL, "bad argument");
    if (k < 1 || k > K) return fail(h, JMID_EINVAL, "k must be in 1..K");
    if (A > 32 || K > 1024 || T > 24) return fail(h, JMID_EINVAL, "jmid_topk supports A <= 32, K <= 1024, T <= 24");
    HIPCHK(h, hipSetDevice(h->device));
    TuneScope tune_scope(&h->tune);
    if (!pos) {
        if (!h->last_pos || h->last_pos_dims[0] != E || h->last_pos_dims[1] != A || h->last_pos_dims[2] != K || h->last_pos_dims[3] != T)
            return fail(h, JMID_EINVAL, "pos = NULL needs a preceding jmid_denoise with p0 and the same E, A, K, T on this handle");
    }
    if (int rc = order_in(h, mem)) return rc;
    const int d = 2 * A;
    const size_t n_pos = (size_t)E * K * A * T * 2, n_sel = (size_t)E * A * k * T * 2, n_lw = (size_t)E * A * k;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    // the global buffer of whitened points only when they do not fit in LDS (E = 64, T = 12, K = 1024, A = 32 would be 400 MB)
    const size_t y_bytes = kde_y_in_lds(A, K) ? 0 : up((size_t)E * T * K * d * 8);
    const size_t o_ll = 0, o_Y = up((size_t)E * T * K * 8), o_bw = o_Y + y_bytes, o_pos = o_bw + up(T * 4),
                 o_sel = o_pos + (pos && mem == JMID_MEM_HOST ? up(n_pos * 4) : 0), o_lw = o_sel + (mem == JMID_MEM_HOST ? up(n_sel * 4) : 0),
                 need = o_lw + (mem == JMID_MEM_HOST ? up(n_lw * 4) : 0);
    if (need > h->kde_ws_bytes) {
        if (h->kde_ws) {
            HIPCHK(h, hipStreamSynchronize(h->stream));
            HIPCHK(h, hipFree(h->kde_ws));
            h->kde_ws = nullptr;
            h->kde_ws_bytes = 0;
        }
        if (hipMalloc((void**)&h->kde_ws, need) != hipSuccess) return fail(h, JMID_ENOMEM, "jmid_topk workspace allocation failed");
        h->kde_ws_bytes = need;
    }
    KdeArgs g{};
    g.E = E; g.A = A; g.K = K; g.T = T; g.k = k;
    g.ll = reinterpret_cast<double*>(h->kde_ws + o_ll);
    g.Y = reinterpret_cast<double*>(h->kde_ws + o_Y);
    g.pos = pos ? pos : h->last_pos;
    g.sel = sel; g.logw = logw;
    if (bw) {
        float* dbw = reinterpret_cast<float*>(h->kde_ws + o_bw);
        HIPCHK(h, hipMemcpyAsync(dbw, bw, T * sizeof(float), mem == JMID_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, h->stream));
        g.bw = dbw;
    }
    if (mem == JMID_MEM_HOST) {
        if (pos) {
            float* dp = reinterpret_cast<float*>(h->kde_ws + o_pos);
            HIPCHK(h, hipMemcpyAsync(dp, pos, n_pos * 4, hipMemcpyHostToDevice, h->stream));
            g.pos = dp;
        }
        g.sel = reinterpret_cast<float*>(h->kde_ws + o_sel);
        g.logw = reinterpret_cast<float*>(h->kde_ws + o_lw);
    }
    {
        ProfScope ps(h, KC_TOPK);
        HIPCHK(h, launch_kde(g, h->stream));
    }
    if (mem == JMID_MEM_HOST) {
        HIPCHK(h, hipMemcpyAsync(sel, g.sel, n_sel * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(logw, g.logw, n_lw * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return order_out(h, mem);
}

int jmid_predict(jmid_handle_t h, int E, int A, int K, int T, int k, const float* x_st, const float* nbr_sum, const float* edge_mask,
                 const float* x_T, const float* p0, float dt, int precision, const float* bw, float* sel, float* logw, float* pos_out) {
    if (!h) return JMID_EINVAL;
    if (int rc = check_ready(h)) return rc;
    if (E <= 0 || A <= 0 || K <= 0 || T <= 0 || k < 1 || k > K) return fail(h, JMID_EINVAL, "jmid_predict: bad dimensions");
    if (!x_st || !nbr_sum || !edge_mask || !x_T || !p0) return fail(h, JMID_EINVAL, "jmid_predict: null input");
    const bool rank = k < K;
    if (rank && (!sel || !logw)) return fail(h, JMID_EINVAL, "jmid_predict: k < K needs sel and logw");
    if (!rank && !pos_out) return fail(h, JMID_EINVAL, "jmid_predict: k == K needs pos_out");
    if (rank && (A > 32 || K > 1024 || T > 24)) return fail(h, JMID_EINVAL, "jmid_predict: the device top-k supports A <= 32, K <= 1024, T <= 24");
    if (h->ddpm) return fail(h, JMID_EINVAL, "jmid_predict samples with DDIM (MID.eval_sicnav: sampling=\"ddim\", MID/mid.py:333)");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t Th = h->hist_len, n = (size_t)E * A, H2 = 2 * (size_t)h->H;
    const size_t n_xs = n * Th * 6, n_nb = n * 2 * Th * 6, n_em = n * 2, n_xT = (size_t)E * K * A * T * 2, n_p0 = n * 2, n_bw = rank && bw ? T : 0;
    const size_t n_sel = rank ? n * k * T * 2 : 0, n_lw = rank ? n * k : 0, n_pos = pos_out ? n_xT : 0;
    auto up = [](size_t floats) { return (floats + 63) / 64 * 64; };
    // upload block | ctx | download block (flag, sel, logw, pos)
    const size_t o_xs = 0, o_nb = o_xs + up(n_xs), o_em = o_nb + up(n_nb), o_xT = o_em + up(n_em), o_p0 = o_xT + up(n_xT), o_bw = o_p0 + up(n_p0),
                 in_floats = o_bw + up(n_bw), o_ctx = in_floats, o_out = o_ctx + up(n * H2), o_flag = o_out, o_sel = o_flag + 64, o_lw = o_sel + up(n_sel),
                 o_pos = o_lw + up(n_lw), total = o_pos + up(n_pos), out_floats = total - o_out;
    if (total * 4 > h->io_dev_bytes) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->io_dev) HIPCHK(h, hipFree(h->io_dev));
        h->io_dev = nullptr;
        h->io_dev_bytes = 0;
        if (hipMalloc((void**)&h->io_dev, total * 4) != hipSuccess) return fail(h, JMID_ENOMEM, "jmid_predict: device staging allocation failed");
        h->io_dev_bytes = total * 4;
    }
    const size_t pin_need = (in_floats + out_floats) * 4;
    if (pin_need > h->pin_bytes) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->pin) HIPCHK(h, hipHostFree(h->pin));
        h->pin = nullptr;
        h->pin_bytes = 0;
        if (hipHostMalloc((void**)&h->pin, pin_need, hipHostMallocDefault) != hipSuccess) return fail(h, JMID_ENOMEM, "jmid_predict: pinned staging allocation failed");
        h->pin_bytes = pin_need;
    }
    float* pin = reinterpret_cast<float*>(h->pin);
    float* dev = reinterpret_cast<float*>(h->io_dev);
    std::memcpy(pin + o_xs, x_st, n_xs * 4);
    std::memcpy(pin + o_nb, nbr_sum, n_nb * 4);
    std::memcpy(pin + o_em, edge_mask, n_em * 4);
    std::memcpy(pin + o_xT, x_T, n_xT * 4);
    std::memcpy(pin + o_p0, p0, n_p0 * 4);
    if (n_bw) std::memcpy(pin + o_bw, bw, n_bw * 4);
    HIPCHK(h, hipMemcpyAsync(dev, pin, in_floats * 4, hipMemcpyHostToDevice, h->stream));
    int rc = 0;
    h->chained = true;
    {
        TuneScope tune_scope(&h->tune);
        ProfScope ps(h, KC_ENCODER);
        EncArgs ea{};
        ea.x_st = dev + o_xs; ea.nbr_sum = dev + o_nb; ea.edge_mask = dev + o_em;
        ea.hist = LstmW{h->lstmT[0][0], h->lstmT[0][1], h->lstmT[0][2]};
        ea.edge[0] = LstmW{h->lstmT[1][0], h->lstmT[1][1], h->lstmT[1][2]};
        ea.edge[1] = LstmW{h->lstmT[2][0], h->lstmT[2][1], h->lstmT[2][2]};
        ea.W1T = h->attW1T; ea.W2T = h->attW2T; ea.v = W(h, "PEDESTRIAN/edge_influence_encoder.v.weight");
        ea.ctx = dev + o_ctx; ea.n = (int)n; ea.Th = (int)Th; ea.H = h->H;
        if (launch_encoder(ea, h->stream) != hipSuccess) rc = fail(h, JMID_EHIP, "jmid_predict: encoder launch failed");
    }
    if (!rc) rc = run_network(h, E, A, K, T, dev + o_xT, dev + o_ctx, dev + o_p0, dt, precision, -1, nullptr, pos_out ? dev + o_pos : nullptr,
                              nullptr, JMID_MEM_DEVICE);
    if (!rc && rank) {
        TuneScope tune_scope(&h->tune);
        rc = topk_on_device(h, E, A, K, T, k, h->last_pos, n_bw ? dev + o_bw : nullptr, dev + o_sel, dev + o_lw);
    }
    h->chained = false;
    if (rc) return rc;
    const bool flagged = precision != JMID_PREC_F32;
    if (flagged) HIPCHK(h, hipMemcpyAsync(dev + o_flag, h->range_flag, sizeof(int), hipMemcpyDeviceToDevice, h->stream));
    float* pout = pin + in_floats;
    HIPCHK(h, hipMemcpyAsync(pout, dev + o_out, out_floats * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (flagged && *reinterpret_cast<const int*>(pout + (o_flag - o_out)))
        return jmid_host::flagged_call(h, *reinterpret_cast<const int*>(pout + (o_flag - o_out)));
    if (rank) {
        std::memcpy(sel, pout + (o_sel - o_out), n_sel * 4);
        std::memcpy(logw, pout + (o_lw - o_out), n_lw * 4);
    }
    if (pos_out) std::memcpy(pos_out, pout + (o_pos - o_out), n_pos * 4);
    return JMID_OK;
}

int jmid_set_chunk_episodes(jmid_handle_t h, int episodes) {
    if (!h || episodes < 0) return JMID_EINVAL;
    h->chunk_eps = episodes;
    drop_graphs(h);
    return JMID_OK;
}

int jmid_set_tuning(jmid_handle_t h, const char* key, int value) {
    if (!h || !key) return JMID_EINVAL;
    const std::string k(key);
    struct Knob {
        const char* name;
        int Tuning::*field;
        int lo, hi;
    };
    // every knob belongs to the handle (h->tune); none is process-wide
#ifdef JMID_DIAGNOSTICS
    static const Knob knobs[] = {
        {"gemm_h_variant", &Tuning::gemm_h_variant, 0, 8},     // 0 auto, 1..8 force a tile variant of the split GEMM
        {"attn_pack", &Tuning::attn_pack, 0, 1},               // 0: one short sequence per wave, 1: packed (iMID)
        {"fuse_embed", &Tuning::fuse_embed, 0, 1},             // 0: separate embed_kernel at the start of every step
        {"bystander_lds", &Tuning::bystander_lds, 0, 160 * 1024},   // unused dynamic LDS requested by row-wise kernels
        {"ln_rows", &Tuning::ln_rows, 0, 128},                 // row tile of the fused GEMM + LayerNorm: 0 auto, 64, 128
        {"ln_fuse", &Tuning::ln_fuse, 0, 2},                   // 0 auto (M >= 7168 tokens), 1 always, 2 never
        {"no_vt_direct", &Tuning::no_vt_direct, 0, 1},         // 1: always V row-major + v_transpose_kernel
        {"gemm_ng", &Tuning::gemm_ng, 0, 64},                  // N-tiles per L2 group of the 256x128 GEMM (0 = auto)
        {"attn_h_variant", &Tuning::attn_h_variant, 0, 2},
        {"vt_stage", &Tuning::vt_stage, 0, 3},                 // V^T of the 256x256 QKV kernel through LDS: 0 / 1 on, 2 off
        {"graph", &Tuning::graph, 0, 2},                       // captured denoise loop of one-chunk calls: 1 on, 0 / 2 off
        {"attn_nsplit", &Tuning::attn_nsplit, 0, 16},
        {"attn_mx", &Tuning::attn_mx, 0, 3},
        {"out_traj", &Tuning::out_traj, 0, 2},
        {"attn_pf", &Tuning::attn_pf, 0, 2},
        {"attn_one_wg", &Tuning::attn_one_wg, 0, 1},
        {"attn_sm", &Tuning::attn_sm, 0, 2},
        {"attn_prio", &Tuning::attn_prio, 0, 2},
        {"mx_ln", &Tuning::mx_ln, 0, 2},
        {"csl_swap", &Tuning::csl_swap, 0, 3},
        {"h1_stage", &Tuning::h1_stage, 0, 2},
        {"gemm_small", &Tuning::gemm_small, 0, 2},             // 1: no deep-ring small-launch GEMM (the round-3 64 x 64 / 128 x 128 shapes)
        {"gemm_pn", &Tuning::gemm_pn, 0, 8},
        {"small_lanes", &Tuning::small_lanes, 0, 2},
        {"small_cmb", &Tuning::small_cmb, 0, 2},               // 2: attn_combine_kernel instead of the split-KV merge inside the out-projection's one-launch GEMM + LayerNorm
        {"small_lnx", &Tuning::small_lnx, 0, 2},               // the one-launch GEMM + LayerNorm with the statistics exchange: 0 on, 2 off (GEMM + add_ln2)
        {"cus", &Tuning::cus, 0, 4096},                         // compute units OUT_LNX may count on (0 = ask the device again, as jmid_create did)
        {"lnx_polls", &Tuning::lnx_polls, 0, 1 << 20},
        {"lnx_withhold", &Tuning::lnx_withhold, 0, 1},
        {"small_lnx2", &Tuning::small_lnx2, 0, 2},             // the same at 33 ... 64 row tiles, two workgroups per CU: 0 on, 2 off
        {"small_qk", &Tuning::small_qk, 0, 2},
        {"small_pn", &Tuning::small_pn, 0, 8},                 // column groups of its XCD tile order: 0 auto
#ifdef JMID_ABLATIONS
        {"attn_abl", &Tuning::attn_abl, 0, 1 << 30},           // timing ablations: results are WRONG (tools/attn_abl.py)
        {"gemm_abl", &Tuning::gemm_abl, 0, 1 << 30},
#endif
    };
#endif
    if (k == "lanes") {     // chunks of the denoise loop in flight at once: 1..4
        if (value < 1 || value > jmid_ctx::kMaxLanes) return fail(h, JMID_EINVAL, "lanes must be 1..4");
        h->lanes = value;
        return JMID_OK;
    }
#ifdef JMID_DIAGNOSTICS
    for (const Knob& kn : knobs)
        if (k == kn.name) {
            if (value < kn.lo || value > kn.hi || (k == "ln_rows" && value != 0 && value != 64 && value != 128))
                return fail(h, JMID_EINVAL, k + " out of range");
            h->tune.*(kn.field) = value;
            if (k == "cus" && value == 0) {
                int cus = 0;
                h->tune.cus = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0 ? cus : 256;
            }
            drop_graphs(h);          // captured loops hold the kernel variants the old knobs selected
            return JMID_OK;
        }
#endif
    return fail(h, JMID_EINVAL, "unknown tuning key " + k);
}

int64_t jmid_graph_replays(jmid_handle_t h) { return h ? h->graph_replays : -1; }

int64_t jmid_erange_count(jmid_handle_t h) { return h ? h->erange_calls : -1; }
int64_t jmid_timeout_count(jmid_handle_t h) { return h ? h->lnx_timeouts : -1; }

int jmid_set_caller_stream(jmid_handle_t h, void* stream) {
    if (!h) return JMID_EINVAL;
    h->caller_stream = reinterpret_cast<hipStream_t>(stream);
    return JMID_OK;
}

int jmid_synchronize(jmid_handle_t h) {
    if (!h) return JMID_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return JMID_OK;
}

}  // extern "C"

