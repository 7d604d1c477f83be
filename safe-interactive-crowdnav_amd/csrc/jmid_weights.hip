// libjmid_hip.so -- weight registry, operand planes of the split-fp16 modes, sampler step tables.
#include "jmid_ctx.hpp"

namespace jmid_host {

void register_shapes(jmid_ctx* h) {
    auto& E = h->expected;
    const size_t d = h->d, ff = h->ff, c = h->ctx_dim + 3, H = h->H;
    auto csl = [&](const std::string& p, size_t din, size_t dout) {
        E[p + "._layer.weight"] = {dout, din};
        E[p + "._layer.bias"] = {dout};
        E[p + "._hyper_bias.weight"] = {dout, c};
        E[p + "._hyper_gate.weight"] = {dout, c};
        E[p + "._hyper_gate.bias"] = {dout};
    };
    csl("concat1", 2, d);
    for (int l = 0; l < h->tf_layer; ++l) {
        std::string p = "transformer_encoder.layers." + std::to_string(l);
        E[p + ".self_attn.in_proj_weight"] = {3 * d, d};
        E[p + ".self_attn.in_proj_bias"] = {3 * d};
        E[p + ".self_attn.out_proj.weight"] = {d, d};
        E[p + ".self_attn.out_proj.bias"] = {d};
        E[p + ".linear1.weight"] = {ff, d};
        E[p + ".linear1.bias"] = {ff};
        E[p + ".linear2.weight"] = {d, ff};
        E[p + ".linear2.bias"] = {d};
        E[p + ".norm1.weight"] = {d};
        E[p + ".norm1.bias"] = {d};
        E[p + ".norm2.weight"] = {d};
        E[p + ".norm2.bias"] = {d};
    }
    csl("concat3", d, h->dmid);
    csl("concat4", h->dmid, h->dlow);
    csl("linear", h->dlow, 2);
    const char* lstm[3] = {"PEDESTRIAN/node_history_encoder", "PEDESTRIAN->PEDESTRIAN/edge_encoder",
                           "PEDESTRIAN->JRDB_ROBOT/edge_encoder"};
    for (int i = 0; i < 3; ++i) {
        std::string p = lstm[i];
        size_t in = i == 0 ? 6 : 12;
        E[p + ".weight_ih_l0"] = {4 * H, in};
        E[p + ".weight_hh_l0"] = {4 * H, H};
        E[p + ".bias_ih_l0"] = {4 * H};
        E[p + ".bias_hh_l0"] = {4 * H};
    }
    E["PEDESTRIAN/edge_influence_encoder.w1.weight"] = {H, H};
    E["PEDESTRIAN/edge_influence_encoder.w2.weight"] = {H, H};
    E["PEDESTRIAN/edge_influence_encoder.v.weight"] = {1, H};
}

int dev_alloc_copy(jmid_ctx* h, float** out, const std::vector<float>& host) {
    HIPCHK(h, hipMalloc((void**)out, host.size() * sizeof(float)));
    HIPCHK(h, hipMemcpy(*out, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int fetch_host(jmid_ctx* h, const std::string& name, std::vector<float>& out) {
    auto it = h->w.find(name);
    if (it == h->w.end()) return fail(h, JMID_ENOWEIGHT, "missing weight " + name);
    out.resize(it->second.n);
    HIPCHK(h, hipMemcpy(out.data(), it->second.p, out.size() * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// upload the per-step time part of the four hyper nets: thyp[i][j] = w0*beta + w1*sin(beta) + w2*cos(beta)
int upload_time_table(jmid_ctx* h) {
    drop_graphs(h);        // the captured loops hold the old table's pointer and the old step coefficients
    if (!h->finalized || h->beta.empty()) return 0;
    const int n = (int)h->beta.size(), tot = h->hl.total;
    std::vector<float> t((size_t)n * tot);
    for (int i = 0; i < n; ++i) {
        const float b = h->beta[i], sb = sinf(b), cb = cosf(b);
        for (int j = 0; j < tot; ++j) {
            const float* w3 = &h->time_w[(size_t)j * 3];
            t[(size_t)i * tot + j] = w3[0] * b + w3[1] * sb + w3[2] * cb;
        }
    }
    if (h->thyp) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipFree(h->thyp));
        h->thyp = nullptr;
    }
    return dev_alloc_copy(h, &h->thyp, t);
}


// bf8 image of W_lo for a device-resident fp32 weight [N, K] (N % 32 == 0, K % 64 == 0)
int make_w8(jmid_ctx* h, const float* dW, int N, int K, jmid_ctx::W8Image* out) {
    HIPCHK(h, hipMalloc((void**)&out->p, (size_t)N * K));
    hipLaunchKernelGGL(w8_image_kernel, dim3(256), dim3(256), 0, h->stream, dW, out->p, N, K, kWScale);
    HIPCHK(h, hipGetLastError());
    return 0;
}


}  // namespace jmid_host

extern "C" {

int jmid_load_weight(jmid_handle_t h, const char* name, const float* host_data, size_t n_elems) {
    if (!h || !name || !host_data) return JMID_EINVAL;
    auto it = h->expected.find(name);
    if (it == h->expected.end()) return fail(h, JMID_EINVAL, std::string("unknown weight name ") + name);
    if (numel(it->second) != n_elems)
        return fail(h, JMID_EINVAL, std::string("size mismatch for ") + name + ": expected " +
                                        std::to_string(numel(it->second)) + ", got " + std::to_string(n_elems));
    HIPCHK(h, hipSetDevice(h->device));
    DevBuf& b = h->w[name];
    if (!b.p) HIPCHK(h, hipMalloc((void**)&b.p, n_elems * sizeof(float)));
    b.n = n_elems;
    drop_graphs(h);
    HIPCHK(h, hipMemcpy(b.p, host_data, n_elems * sizeof(float), hipMemcpyHostToDevice));
    h->finalized = false;
    return JMID_OK;
}

int jmid_finalize_weights(jmid_handle_t h) {
    if (!h) return JMID_EINVAL;
    for (auto& kv : h->expected)
        if (!h->w.count(kv.first)) return fail(h, JMID_ENOWEIGHT, "missing weight " + kv.first);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (float** p : {&h->pe, &h->Whyp, &h->bhyp, &h->attW1T, &h->attW2T})
        if (*p) {
            hipFree(*p);
            *p = nullptr;
        }
    for (auto& l : h->lstmT)
        for (float*& p : l)
            if (p) {
                hipFree(p);
                p = nullptr;
            }
    const int d = h->d, C = h->ctx_dim, CC = C + 3;
    // positional encoding table, max_len = 24 (MID/models/common.py:37-51; diffusion.py:116-118)
    {
        std::vector<float> pe((size_t)24 * d);
        const float coef = (float)(-std::log(10000.0) / (double)d);  // python scalar -> fp32, as torch does
        for (int pos = 0; pos < 24; ++pos)
            for (int i = 0; i < d; i += 2) {
                const float div = (float)std::exp((double)((float)i * coef));
                const float arg = (float)pos * div;
                pe[(size_t)pos * d + i] = (float)std::sin((double)arg);
                if (i + 1 < d) pe[(size_t)pos * d + i + 1] = (float)std::cos((double)arg);
            }
        if (int rc = dev_alloc_copy(h, &h->pe, pe)) return rc;
    }
    // packed ctx-part of the hyper nets [hl.total, C], their biases, and the 3 time columns (host)
    {
        const HyperLayout& L = h->hl;
        std::vector<float> Wp((size_t)L.total * C), bp(L.total, 0.f);
        h->time_w.assign((size_t)L.total * 3, 0.f);
        struct Part {
            const char* prefix;
            int goff, boff, dout;
        } parts[4] = {{"concat1", L.g1, L.b1, d}, {"concat3", L.g3, L.b3, h->dmid}, {"concat4", L.g4, L.b4, h->dlow},
                      {"linear", L.go, L.bo, 2}};
        for (auto& pt : parts) {
            std::vector<float> wg, bg, wb;
            if (int rc = fetch_host(h, std::string(pt.prefix) + "._hyper_gate.weight", wg)) return rc;
            if (int rc = fetch_host(h, std::string(pt.prefix) + "._hyper_gate.bias", bg)) return rc;
            if (int rc = fetch_host(h, std::string(pt.prefix) + "._hyper_bias.weight", wb)) return rc;
            for (int j = 0; j < pt.dout; ++j) {
                for (int c = 0; c < C; ++c) {
                    Wp[(size_t)(pt.goff + j) * C + c] = wg[(size_t)j * CC + 3 + c];
                    Wp[(size_t)(pt.boff + j) * C + c] = wb[(size_t)j * CC + 3 + c];
                }
                bp[pt.goff + j] = bg[j];
                for (int c = 0; c < 3; ++c) {
                    h->time_w[(size_t)(pt.goff + j) * 3 + c] = wg[(size_t)j * CC + c];
                    h->time_w[(size_t)(pt.boff + j) * 3 + c] = wb[(size_t)j * CC + c];
                }
            }
        }
        if (int rc = dev_alloc_copy(h, &h->Whyp, Wp)) return rc;
        if (int rc = dev_alloc_copy(h, &h->bhyp, bp)) return rc;
    }
    // transposed LSTM / attention weights for the encoder kernel
    {
        const char* lstm[3] = {"PEDESTRIAN/node_history_encoder", "PEDESTRIAN->PEDESTRIAN/edge_encoder",
                               "PEDESTRIAN->JRDB_ROBOT/edge_encoder"};
        const int H = h->H, H4 = 4 * H;
        for (int i = 0; i < 3; ++i) {
            const int in = i == 0 ? 6 : 12;
            std::vector<float> wih, whh, bih, bhh;
            std::string p = lstm[i];
            if (int rc = fetch_host(h, p + ".weight_ih_l0", wih)) return rc;
            if (int rc = fetch_host(h, p + ".weight_hh_l0", whh)) return rc;
            if (int rc = fetch_host(h, p + ".bias_ih_l0", bih)) return rc;
            if (int rc = fetch_host(h, p + ".bias_hh_l0", bhh)) return rc;
            std::vector<float> wihT((size_t)in * H4), whhT((size_t)H * H4), b(H4);
            for (int r = 0; r < H4; ++r) {
                for (int k = 0; k < in; ++k) wihT[(size_t)k * H4 + r] = wih[(size_t)r * in + k];
                for (int k = 0; k < H; ++k) whhT[(size_t)k * H4 + r] = whh[(size_t)r * H + k];
                b[r] = bih[r] + bhh[r];
            }
            if (int rc = dev_alloc_copy(h, &h->lstmT[i][0], wihT)) return rc;
            if (int rc = dev_alloc_copy(h, &h->lstmT[i][1], whhT)) return rc;
            if (int rc = dev_alloc_copy(h, &h->lstmT[i][2], b)) return rc;
        }
        std::vector<float> w1, w2;
        if (int rc = fetch_host(h, "PEDESTRIAN/edge_influence_encoder.w1.weight", w1)) return rc;
        if (int rc = fetch_host(h, "PEDESTRIAN/edge_influence_encoder.w2.weight", w2)) return rc;
        std::vector<float> w1T((size_t)H * H), w2T((size_t)H * H);
        for (int r = 0; r < H; ++r)
            for (int k = 0; k < H; ++k) {
                w1T[(size_t)k * H + r] = w1[(size_t)r * H + k];
                w2T[(size_t)k * H + r] = w2[(size_t)r * H + k];
            }
        if (int rc = dev_alloc_copy(h, &h->attW1T, w1T)) return rc;
        if (int rc = dev_alloc_copy(h, &h->attW2T, w2T)) return rc;
    }
    // hi/lo fp16 planes of every GEMM weight (split once; activations are split by the producing kernels)
    {
        for (auto& kv : h->wsplit) {
            hipFree(kv.second.hi);
            hipFree(kv.second.lo);
        }
        h->wsplit.clear();
        for (auto& kv : h->w8) hipFree(kv.second.p);
        h->w8.clear();
        for (auto& kv : h->w16) {
            hipFree(kv.second.hi);
            hipFree(kv.second.lo);
        }
        h->w16.clear();
        if (!h->range_flag) {
            HIPCHK(h, hipMalloc((void**)&h->range_flag, sizeof(int)));
        }
        HIPCHK(h, hipMemsetAsync(h->range_flag, 0, sizeof(int), h->stream));
        std::vector<std::string> names = {"concat3._layer.weight", "concat4._layer.weight"};
        for (int l = 0; l < h->tf_layer; ++l) {
            const std::string p = "transformer_encoder.layers." + std::to_string(l);
            names.push_back(p + ".self_attn.in_proj_weight");
            names.push_back(p + ".self_attn.out_proj.weight");
            names.push_back(p + ".linear1.weight");
            names.push_back(p + ".linear2.weight");
        }
        for (const auto& nm : names) {
            const DevBuf& b = h->w[nm];
            const std::vector<size_t>& shp = h->expected[nm];   // [N, K]
            const size_t pe = blk_plane_elems(shp[0], (int)shp[1]);
            HalfPair hp;
            HIPCHK(h, hipMalloc((void**)&hp.hi, pe * sizeof(half_t)));
            HIPCHK(h, hipMalloc((void**)&hp.lo, pe * sizeof(half_t)));
            HIPCHK(h, hipMemsetAsync(hp.hi, 0, pe * sizeof(half_t), h->stream));
            HIPCHK(h, hipMemsetAsync(hp.lo, 0, pe * sizeof(half_t), h->stream));
            hipLaunchKernelGGL(split_planes_blocked_kernel, dim3(256), dim3(256), 0, h->stream, b.p, hp.hi, hp.lo,
                               (int)shp[0], (int)shp[1], h->range_flag, kWScale);
            HIPCHK(h, hipGetLastError());
            h->wsplit[nm] = hp;
            if (shp[0] % 32 == 0 && shp[1] % 64 == 0) {
                jmid_ctx::W8Image img;
                if (int rc = make_w8(h, b.p, (int)shp[0], (int)shp[1], &img)) return rc;
                h->w8[nm] = img;
            }
        }
        if (h->d == GLN_BN) {   // k16-panel copies for gemm_ln_f16x3_kernel (row-complete tiles need N == 512) and tail_f16x3_kernel
            std::vector<std::string> k16names;
            for (int l = 0; l < h->tf_layer; ++l) {
                const std::string p = "transformer_encoder.layers." + std::to_string(l);
                k16names.push_back(p + ".self_attn.out_proj.weight");
                k16names.push_back(p + ".linear2.weight");
            }
            {
                for (const std::string& nm : k16names) {
                    const DevBuf& b = h->w[nm];
                    const std::vector<size_t>& shp = h->expected[nm];   // [512, K]
                    HalfPair hp;
                    HIPCHK(h, hipMalloc((void**)&hp.hi, shp[0] * shp[1] * sizeof(half_t)));
                    HIPCHK(h, hipMalloc((void**)&hp.lo, shp[0] * shp[1] * sizeof(half_t)));
                    hipLaunchKernelGGL(split_planes_k16_kernel, dim3(256), dim3(256), 0, h->stream, b.p, hp.hi, hp.lo,
                                       (int)shp[0], (int)shp[1]);
                    HIPCHK(h, hipGetLastError());
                    h->w16[nm] = hp;
                }
            }
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        int flag = 0;
        HIPCHK(h, hipMemcpy(&flag, h->range_flag, sizeof(int), hipMemcpyDeviceToHost));
        h->weights_in_half_range = flag == 0;
        HIPCHK(h, hipMemsetAsync(h->range_flag, 0, sizeof(int), h->stream));
    }
    h->finalized = true;
    return upload_time_table(h);
}

int jmid_set_ddim_table(jmid_handle_t h, int n_steps, const float* beta, const float* c_e, const float* c_x,
                        const float* n_x, const float* n_e) {
    if (!h || n_steps <= 0 || !beta || !c_e || !c_x || !n_x || !n_e) return fail(h, JMID_EINVAL, "bad ddim table");
    h->beta.assign(beta, beta + n_steps);
    h->c_e.assign(c_e, c_e + n_steps);
    h->c_x.assign(c_x, c_x + n_steps);
    h->n_x.assign(n_x, n_x + n_steps);
    h->n_e.assign(n_e, n_e + n_steps);
    h->ddpm = false;
    HIPCHK(h, hipSetDevice(h->device));
    return upload_time_table(h);
}

int jmid_set_ddpm_table(jmid_handle_t h, int n_steps, const float* beta, const float* c0, const float* c1,
                        const float* sigma, const int* use_noise) {
    if (!h || n_steps <= 0 || !beta || !c0 || !c1 || !sigma || !use_noise) return fail(h, JMID_EINVAL, "bad ddpm table");
    h->beta.assign(beta, beta + n_steps);
    h->p_c0.assign(c0, c0 + n_steps);
    h->p_c1.assign(c1, c1 + n_steps);
    h->p_sigma.assign(sigma, sigma + n_steps);
    h->p_noise.assign(use_noise, use_noise + n_steps);
    h->c_e.assign(n_steps, 0.f);
    h->c_x.assign(n_steps, 1.f);
    h->n_x.assign(n_steps, 1.f);
    h->n_e.assign(n_steps, 0.f);
    h->ddpm = true;
    HIPCHK(h, hipSetDevice(h->device));
    return upload_time_table(h);
}

}  // extern "C"

