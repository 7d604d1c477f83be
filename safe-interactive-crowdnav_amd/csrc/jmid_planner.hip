// libjmid_hip.so -- chunk plan, step workspace, one net evaluation, the denoise loop.
#include "jmid_ctx.hpp"
#include "jmid_launch.hpp"

namespace jmid_host {

void drop_graphs(jmid_ctx* h) {
    for (auto& kv : h->graphs)
        if (kv.second.exec) hipGraphExecDestroy(kv.second.exec);
    h->graphs.clear();
}

// Error paths and arena owners: nothing may still run on a lane stream when the arena is reused or freed.
void sync_lanes(jmid_ctx* h) {
    for (int l = 0; l + 1 < jmid_ctx::kMaxLanes; ++l)
        if (h->lane_stream[l]) (void)hipStreamSynchronize(h->lane_stream[l]);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
}

int ensure_arena(jmid_ctx* h, size_t bytes) {
    if (bytes <= h->arena_bytes) return 0;
    drop_graphs(h);
    h->last_pos = nullptr;
    if (h->arena) {
        sync_lanes(h);
        HIPCHK(h, hipFree(h->arena));
        h->arena = nullptr;
        h->arena_bytes = 0;
    }
    hipError_t e = hipMalloc((void**)&h->arena, bytes);
    if (e != hipSuccess) return fail(h, JMID_ENOMEM, "workspace allocation of " + std::to_string(bytes) + " bytes failed");
    h->arena_bytes = bytes;
    return 0;
}

struct StepBuffers {
    float *X, *QKV, *ATT, *Y, *H1, *Y3, *Y4;
    // F16X3 path: hi/lo planes
    half_t *Xh, *Xl, *Qh, *Ql, *Kh, *Kl, *Vh, *Vl, *Vth, *Vtl, *Ah, *Al, *H1h, *H1l, *Y3h, *Y3l;
    size_t vt_elems;
    int attn_nsplit;          // split-KV factor of the attention launch (1 = off)
    float *Opart, *MLpart;
    unsigned long long* ln_xchg;   // exchange granules of the small-launch GEMM + LayerNorm (gemm_small.hpp, OUT_LNX): kLnxWords words, zeroed once per call
};

half_t* take_half(Carver& c, size_t n) { return reinterpret_cast<half_t*>(c.take((n + 1) / 2)); }

// attention geometry of a chunk
struct SeqGeom {
    int nseq, S, Spad;
};
SeqGeom seq_geom(const jmid_ctx* h, int Ec, int A, int K, int T) {
    SeqGeom g;
    g.nseq = h->net_kind == JMID_NET_JMID ? Ec : Ec * K * A;
    g.S = h->net_kind == JMID_NET_JMID ? K * A * T : T;
    g.Spad = vt_spad(g.S);
    return g;
}

size_t step_ws_floats(const jmid_ctx* h, size_t Mc, int precision, const SeqGeom& sg, int nsplit, StepBuffers* sb,
                      char* base) {
    Carver c(base);
    StepBuffers s{};
    s.X = c.take(Mc * h->d);
    s.Y = c.take((Mc + 63) / 64 * 64 * h->d);     // (whole 64-row tiles)
    s.ln_xchg = reinterpret_cast<unsigned long long*>(c.take(kLnxWords));
    s.Y4 = c.take(Mc * h->dlow);
    if (precision == JMID_PREC_F32) {
        s.QKV = c.take(Mc * 3 * h->d);
        s.ATT = c.take(Mc * h->d);
        s.H1 = c.take(Mc * h->ff);
        s.Y3 = c.take(Mc * h->dmid);
    } else {
        s.Xh = take_half(c, blk_plane_elems(Mc, h->d));
        s.Xl = take_half(c, blk_plane_elems(Mc, h->d));
        if (h->net_kind == JMID_NET_JMID) {
            s.Qh = take_half(c, Mc * h->d);
            s.Ql = take_half(c, Mc * h->d);
            s.Kh = take_half(c, Mc * h->d);
            s.Kl = take_half(c, Mc * h->d);
            s.Vh = take_half(c, Mc * h->d);
            s.Vl = take_half(c, Mc * h->d);
            s.vt_elems = (size_t)sg.nseq * h->d * sg.Spad;
            s.Vth = take_half(c, s.vt_elems);
            s.Vtl = take_half(c, s.vt_elems);
            s.attn_nsplit = nsplit;
            if (s.attn_nsplit > 1) {
                s.Opart = c.take((size_t)s.attn_nsplit * Mc * h->d);
                s.MLpart = c.take((size_t)s.attn_nsplit * Mc * h->nhead * 2);
            }
        } else {
            s.QKV = c.take(Mc * 3 * h->d);  // iMID: sequences of T tokens, exact-fp32 attention kernel
        }
        s.Ah = take_half(c, blk_plane_elems(Mc, h->d));
        s.Al = take_half(c, blk_plane_elems(Mc, h->d));
        s.H1h = take_half(c, blk_plane_elems(Mc, h->ff));
        s.H1l = take_half(c, blk_plane_elems(Mc, h->ff));
        s.Y3h = take_half(c, blk_plane_elems(Mc, h->dmid));
        s.Y3l = take_half(c, blk_plane_elems(Mc, h->dmid));
    }
    if (sb) *sb = s;
    return c.off;
}

// one evaluation of the net on a chunk of whole episodes + (optionally) the DDIM update
int net_step(jmid_ctx* h, const StepBuffers& sb, int Ec, int A, int K, int T, int step_idx, float* x_chunk,
             const float* hyp_chunk, float* e_out, int precision, const float* z_chunk = nullptr,
             bool embed_done = false, int next_step = -1) {
    // embed_done: the previous step's output kernel already embedded x for this step; next_step >= 0: this step's
    // output kernel does the same for step `next_step` (same chunk, same buffers)
    const bool split = precision != JMID_PREC_F32;
    const int R = Ec * K * A, M = R * T;
    const int d = h->d, ff = h->ff;
    const float* thyp = h->thyp + (size_t)step_idx * h->hl.total;
    const RowMap rm = make_rowmap(T, A, K * A, (unsigned long long)M);
    // JMID_PREC_F16MX at d_model 512: second-generation LayerNorm kernels (gemm_ln2_mx.hpp) - the lo plane of the residual stream
    // is a byte plane (it lives in the memory of the fp16 one), the row statistics are summed in that file's order
    const bool mxv2 = split && h->mx && d == GLN_BN && tune().mx_ln != 2;
    unsigned char* Xl8 = mxv2 ? reinterpret_cast<unsigned char*>(sb.Xl) : nullptr;
    const auto embed_args = [&](const float* th) {
        return EmbedArgs{x_chunk, W(h, "concat1._layer.weight"), W(h, "concat1._layer.bias"), h->pe, hyp_chunk, th,
                         split ? nullptr : sb.X, M, d, h->hl.total, h->hl.g1, h->hl.b1, rm, split ? sb.Xh : nullptr,
                         split && !mxv2 ? sb.Xl : nullptr, Xl8};
    };
    if (!embed_done) {
        ProfScope ps(h, KC_EMBED);
        EmbedArgs ea = embed_args(thyp);
        const long total = (long)M * (d / 4);
        int blocks = (int)std::min<long>((total + 255) / 256, 256L * 16);
        hipLaunchKernelGGL(embed_kernel, dim3(blocks), dim3(256), bystander_lds(embed_kernel), h->stream, ea);
        HIPCHK(h, hipGetLastError());
    }
    const SeqGeom sg = seq_geom(h, Ec, A, K, T);
    const int nseq = sg.nseq, S = sg.S;
    const int hd = d / h->nhead;
    const float att_scale = 1.0f / sqrtf((float)hd);
    if (!split) {
        for (int l = 0; l < h->tf_layer; ++l) {
            const std::string p = "transformer_encoder.layers." + std::to_string(l);
            GemmArgs g{};
            g.rmap = rm;
            // QKV projection
            g.A = sb.X; g.lda = d; g.W = W(h, p + ".self_attn.in_proj_weight"); g.ldw = d;
            g.bias = W(h, p + ".self_attn.in_proj_bias"); g.C = sb.QKV; g.ldc = 3 * d; g.M = M; g.N = 3 * d; g.K = d;
            if (int rc = run_gemm<EPI_BIAS>(h, KC_GEMM_QKV, g)) return rc;
            {
                ProfScope ps(h, KC_ATTN);
                AttnArgs aa{sb.QKV, sb.ATT, S, d, h->nhead, att_scale, nullptr, nullptr};
                HIPCHK(h, launch_attn_f32(aa, nseq, hd, h->stream));
            }
            // attention output projection + residual + LN1
            g.A = sb.ATT; g.lda = d; g.W = W(h, p + ".self_attn.out_proj.weight"); g.ldw = d;
            g.bias = W(h, p + ".self_attn.out_proj.bias"); g.C = sb.Y; g.ldc = d; g.N = d; g.K = d;
            if (int rc = run_gemm<EPI_BIAS>(h, KC_GEMM_OUT, g)) return rc;
            if (int rc = run_add_ln(h, sb.X, sb.Y, W(h, p + ".norm1.weight"), W(h, p + ".norm1.bias"), M, d)) return rc;
            // feed-forward
            g.A = sb.X; g.lda = d; g.W = W(h, p + ".linear1.weight"); g.ldw = d; g.bias = W(h, p + ".linear1.bias");
            g.C = sb.H1; g.ldc = ff; g.N = ff; g.K = d;
            if (int rc = run_gemm<EPI_BIAS_RELU>(h, KC_GEMM_FF1, g)) return rc;
            g.A = sb.H1; g.lda = ff; g.W = W(h, p + ".linear2.weight"); g.ldw = ff; g.bias = W(h, p + ".linear2.bias");
            g.C = sb.Y; g.ldc = d; g.N = d; g.K = ff;
            if (int rc = run_gemm<EPI_BIAS>(h, KC_GEMM_FF2, g)) return rc;
            if (int rc = run_add_ln(h, sb.X, sb.Y, W(h, p + ".norm2.weight"), W(h, p + ".norm2.bias"), M, d)) return rc;
        }
        // tail: concat3, concat4 (ConcatSquash epilogues)
        GemmArgs g{};
        g.rmap = rm; g.hyp = hyp_chunk; g.thyp = thyp; g.hyp_ld = h->hl.total; g.M = M;
        g.A = sb.X; g.lda = d; g.W = W(h, "concat3._layer.weight"); g.ldw = d; g.bias = W(h, "concat3._layer.bias");
        g.C = sb.Y3; g.ldc = h->dmid; g.N = h->dmid; g.K = d; g.goff = h->hl.g3; g.boff = h->hl.b3;
        if (int rc = run_gemm<EPI_CSL>(h, KC_GEMM_TAIL, g)) return rc;
        g.A = sb.Y3; g.lda = h->dmid; g.W = W(h, "concat4._layer.weight"); g.ldw = h->dmid;
        g.bias = W(h, "concat4._layer.bias"); g.C = sb.Y4; g.ldc = h->dlow; g.N = h->dlow; g.K = h->dmid;
        g.goff = h->hl.g4; g.boff = h->hl.b4;
        if (int rc = run_gemm<EPI_CSL>(h, KC_GEMM_TAIL, g)) return rc;
    } else {
        const bool joint = h->net_kind == JMID_NET_JMID;
        for (int l = 0; l < h->tf_layer; ++l) {
            const std::string p = "transformer_encoder.layers." + std::to_string(l);
            GemmHArgs g{};
            g.rmap = rm; g.M = M;
            bool cmb_in_gemm = false;
            int cmb_ns = 0;
            size_t cmb_Mtot = 0;
            const HalfPair& win = h->wsplit[p + ".self_attn.in_proj_weight"];
            g.Ahi = sb.Xh; g.Alo = sb.Xl; g.Whi = win.hi; g.Wlo = win.lo;
            set_w8(h, g, p + ".self_attn.in_proj_weight");
            g.bias = W(h, p + ".self_attn.in_proj_bias"); g.N = 3 * d; g.K = d;
            if (joint) {
                // S % 4 == 0: the QKV epilogue writes V^T itself; otherwise V row-major + v_transpose_kernel
                const bool vt_direct = (S % 4 == 0) && !tune().no_vt_direct;
                g.Chi = sb.Qh; g.Clo = sb.Ql; g.Khi = sb.Kh; g.Klo = sb.Kl;
                g.Vthi = vt_direct ? sb.Vth : sb.Vh; g.Vtlo = vt_direct ? sb.Vtl : sb.Vl; g.vt_direct = vt_direct;
                g.d = d; g.hd = hd; g.S = S; g.Spad = sg.Spad; g.qscale = att_scale * 1.4426950408889634f;
                // JMID_PREC_F16MX, head_dim 128 (the LDS-DMA attention kernel): bf8 images of K_hi / K_lo in the K_lo plane's memory, for
                // the logits' correction terms as bf8 MFMAs (attention 7 % faster; "attn_mx" = 2: fp16 terms as in F16X2)
                // (the register-staged GEMM variants a knob can force are F16X2's kernels: no image stores)
                const bool k8 = h->mx && hd == 128 && tune().attn_h_variant == 0 && tune().attn_mx != 2 &&
                                tune().gemm_h_variant != 1 && tune().gemm_h_variant != 2;
                unsigned char* k8h = k8 ? reinterpret_cast<unsigned char*>(sb.Kl) : nullptr;
                unsigned char* k8l = k8 ? k8h + (size_t)M * d : nullptr;
                unsigned char* q8l = k8 && tune().attn_mx != 3 ? reinterpret_cast<unsigned char*>(sb.Ql) : nullptr;   // 3: Q_lo as fp16 (A/B)
                g.K8h = k8h; g.K8l = k8l; g.Q8l = q8l;
                if (int rc = (run_gemm_h<EPI_BIAS, OUT_QKV>(h, KC_GEMM_QKV, g))) return rc;
                if (!vt_direct) {
                    ProfScope ps(h, KC_VTRANS);
                    hipLaunchKernelGGL(v_transpose_kernel, dim3((S + 63) / 64, d / 64, nseq), dim3(256), 0, h->stream,
                                       sb.Vh, sb.Vl, sb.Vth, sb.Vtl, S, sg.Spad, d, hd);
                    HIPCHK(h, hipGetLastError());
                }
                ProfScope ps(h, KC_ATTN);
                const int ns = sb.attn_nsplit;   // per call, not per chunk (run_network)
                AttnHArgs aa{sb.Qh, sb.Ql, sb.Kh, sb.Kl, sb.Vth, sb.Vtl, sb.Ah, sb.Al, S, sg.Spad, d, h->nhead,
                             att_scale, h->range_flag, ns, sb.Opart, sb.MLpart, h->x2, k8h, k8l, q8l};
                // one scene: the partial outputs are merged in front of the out-projection's K loop (gemm_small.hpp, lnx_combine) when
                // that launch is the one with the LayerNorm inside (the same conditions as below)
                cmb_in_gemm = mxv2 && !h->lnx_off && !(tune().ln_fuse != 2 && (tune().ln_fuse == 1 || M >= 7168)) && small_lnx_fits(M, d) &&
                              small_cmb_fits(ns, hd, h->x2);
                aa.skip_combine = cmb_in_gemm;
                cmb_ns = ns;
                cmb_Mtot = (size_t)nseq * S;
                HIPCHK(h, launch_attn_f16x3(aa, nseq, hd, h->stream));
                g.K8h = nullptr; g.K8l = nullptr; g.Q8l = nullptr;
            } else {
                g.C = sb.QKV; g.ldc = 3 * d;
                if (int rc = (run_gemm_h<EPI_BIAS, OUT_F32>(h, KC_GEMM_QKV, g))) return rc;
                ProfScope ps(h, KC_ATTN);
                AttnArgs aa{sb.QKV, nullptr, S, d, h->nhead, att_scale, sb.Ah, sb.Al};
                HIPCHK(h, launch_attn_f32(aa, nseq, hd, h->stream));
            }
            // row-complete GEMM with residual + LayerNorm fused in (gemm_ln_f16x3.hpp) from 7168 tokens (6 episodes
            // per launch: 36.8 vs 39.1 ms per 12-episode call; 5: 33.8 vs 33.5, 4: 29.6 vs 28.8)
            // (enough row tiles to occupy the chip); otherwise GEMM -> fp32 Y -> add_ln.  Both give bit-identical rows.
            const bool ln_fused = d == GLN_BN && tune().ln_fuse != 2 && (tune().ln_fuse == 1 || M >= 7168);
            // one scene in F16MX (one chunk of <= 2048 rows, byte lo plane of the second-generation LayerNorm: mxv2): GEMM + residual +
            // LayerNorm in ONE small launch whose workgroups exchange the row statistics (gemm_small.hpp, OUT_LNX; two launches per
            // layer fewer); a handle on which such a kernel ever gave up waiting (lnx_off) stays on the pair
            if (ln_fused && mxv2) {
                GemmLn2Args g2{sb.Ah, h->w16[p + ".self_attn.out_proj.weight"].hi, h->w8[p + ".self_attn.out_proj.weight"].p,
                               W(h, p + ".self_attn.out_proj.bias"), W(h, p + ".norm1.weight"), W(h, p + ".norm1.bias"), sb.Xh, Xl8,
                               M, d, 1e-5f, h->range_flag, 0};
                ProfScope ps(h, KC_GEMM_OUT);
                HIPCHK(h, launch_gemm_ln2_mx(g2, h->stream));
            } else if (ln_fused) {
                const HalfPair& w16 = h->w16[p + ".self_attn.out_proj.weight"];
                GemmLnArgs gl{sb.Ah, sb.Al, w16.hi, w16.lo, W(h, p + ".self_attn.out_proj.bias"), W(h, p + ".norm1.weight"),
                              W(h, p + ".norm1.bias"), sb.Xh, sb.Xl, M, d, 1e-5f, h->range_flag, h->x2};
                if (h->mx) {
                    auto it8 = h->w8.find(p + ".self_attn.out_proj.weight");
                    if (it8 != h->w8.end()) gl.W8 = it8->second.p;
                }
                ProfScope ps(h, KC_GEMM_OUT);
                HIPCHK(h, launch_gemm_ln(gl, h->stream));
            } else {
                const HalfPair& wout = h->wsplit[p + ".self_attn.out_proj.weight"];
                g.Ahi = sb.Ah; g.Alo = sb.Al; g.Whi = wout.hi; g.Wlo = wout.lo;
                set_w8(h, g, p + ".self_attn.out_proj.weight");
                g.bias = W(h, p + ".self_attn.out_proj.bias"); g.C = sb.Y; g.ldc = d; g.N = d; g.K = d;
                if (cmb_in_gemm && !(mxv2 && !h->lnx_off && small_lnx_fits(M, g.K))) return fail(h, JMID_EINVAL, "split-KV merge left to a launch that does not exist");
                if (mxv2 && !h->lnx_off && small_lnx_fits(M, g.K)) {
                    g.cmb_O = cmb_in_gemm ? sb.Opart : nullptr; g.cmb_ML = sb.MLpart; g.cmb_ns = cmb_ns; g.cmb_nhead = h->nhead; g.cmb_Mtot = (unsigned)cmb_Mtot;
                    g.ln_gamma = W(h, p + ".norm1.weight"); g.ln_beta = W(h, p + ".norm1.bias"); g.ln_xh = sb.Xh; g.ln_xl = nullptr;
                    g.ln_xl8 = Xl8; g.ln_xchg = sb.ln_xchg; g.ln_eps = 1e-5f; g.ln_no_lo = 0;
                    if (int rc = run_gemm_lnx_small(h, KC_GEMM_OUT, g)) return rc;
                } else
                {
                if (int rc = (run_gemm_h<EPI_BIAS, OUT_F32>(h, KC_GEMM_OUT, g))) return rc;
                if (int rc = run_add_ln(h, sb.X, sb.Y, W(h, p + ".norm1.weight"), W(h, p + ".norm1.bias"), M, d, sb.Xh,
                                        sb.Xl, mxv2, 0))
                    return rc;
                }
            }
            const HalfPair& w1 = h->wsplit[p + ".linear1.weight"];
            g.Ahi = sb.Xh; g.Alo = sb.Xl; g.Whi = w1.hi; g.Wlo = w1.lo;
            set_w8(h, g, p + ".linear1.weight");
            g.bias = W(h, p + ".linear1.bias"); g.Chi = sb.H1h; g.Clo = sb.H1l; g.ldc = ff; g.N = ff; g.K = d;
            if (int rc = (run_gemm_h<EPI_BIAS_RELU, OUT_SPLIT>(h, KC_GEMM_FF1, g))) return rc;
            if (ln_fused && mxv2) {
                GemmLn2Args g2{sb.H1h, h->w16[p + ".linear2.weight"].hi, h->w8[p + ".linear2.weight"].p, W(h, p + ".linear2.bias"),
                               W(h, p + ".norm2.weight"), W(h, p + ".norm2.bias"), sb.Xh, Xl8, M, ff, 1e-5f, h->range_flag,
                               l + 1 == h->tf_layer};       // the residual stream ends here: concat3 reads X_hi only
                ProfScope ps(h, KC_GEMM_FF2);
                HIPCHK(h, launch_gemm_ln2_mx(g2, h->stream));
            } else if (ln_fused) {
                const HalfPair& w16 = h->w16[p + ".linear2.weight"];
                GemmLnArgs gl{sb.H1h, sb.H1l, w16.hi, w16.lo, W(h, p + ".linear2.bias"), W(h, p + ".norm2.weight"),
                              W(h, p + ".norm2.bias"), sb.Xh, sb.Xl, M, ff, 1e-5f, h->range_flag, h->x2};
                gl.no_lo_out = h->x2 && l + 1 == h->tf_layer;     // the residual stream ends here: concat3 reads X_hi only
                if (h->mx) {
                    auto it8 = h->w8.find(p + ".linear2.weight");
                    if (it8 != h->w8.end()) gl.W8 = it8->second.p;
                }
                ProfScope ps(h, KC_GEMM_FF2);
                HIPCHK(h, launch_gemm_ln(gl, h->stream));
            } else {
                const HalfPair& w2 = h->wsplit[p + ".linear2.weight"];
                g.Ahi = sb.H1h; g.Alo = sb.H1l; g.Whi = w2.hi; g.Wlo = w2.lo;
                set_w8(h, g, p + ".linear2.weight");
                g.bias = W(h, p + ".linear2.bias"); g.C = sb.Y; g.ldc = d; g.N = d; g.K = ff;
                if (mxv2 && !h->lnx_off && small_lnx_fits(M, g.K)) {
                    g.cmb_O = nullptr;
                    g.ln_gamma = W(h, p + ".norm2.weight"); g.ln_beta = W(h, p + ".norm2.bias"); g.ln_xh = sb.Xh; g.ln_xl = nullptr;
                    g.ln_xl8 = Xl8; g.ln_xchg = sb.ln_xchg; g.ln_eps = 1e-5f; g.ln_no_lo = mxv2 && l + 1 == h->tf_layer;
                    if (int rc = run_gemm_lnx_small(h, KC_GEMM_FF2, g)) return rc;
                } else
                {
                if (int rc = (run_gemm_h<EPI_BIAS, OUT_F32>(h, KC_GEMM_FF2, g))) return rc;
                if (int rc = run_add_ln(h, sb.X, sb.Y, W(h, p + ".norm2.weight"), W(h, p + ".norm2.bias"), M, d, sb.Xh,
                                        sb.Xl, mxv2, l + 1 == h->tf_layer))
                    return rc;
                }
            }
        }
        // concat3 -> concat4 as two launches, the output layer + sampler update + next embedding as a third (one fused kernel for all
        // three was built in round 3 and measured slower at every batch size: docs/NOTEBOOK.md)
        {
        GemmHArgs g{};
        g.rmap = rm; g.hyp = hyp_chunk; g.thyp = thyp; g.hyp_ld = h->hl.total; g.M = M;
        const HalfPair& w3 = h->wsplit["concat3._layer.weight"];
        g.Ahi = sb.Xh; g.Alo = sb.Xl; g.Whi = w3.hi; g.Wlo = w3.lo;
        set_w8(h, g, "concat3._layer.weight");
        g.bias = W(h, "concat3._layer.bias"); g.Chi = sb.Y3h; g.Clo = sb.Y3l; g.ldc = h->dmid; g.N = h->dmid; g.K = d;
        g.goff = h->hl.g3; g.boff = h->hl.b3;
        if (int rc = (run_gemm_h<EPI_CSL, OUT_SPLIT>(h, KC_GEMM_TAIL, g))) return rc;
        const HalfPair& w4 = h->wsplit["concat4._layer.weight"];
        g.Ahi = sb.Y3h; g.Alo = sb.Y3l; g.Whi = w4.hi; g.Wlo = w4.lo;
        set_w8(h, g, "concat4._layer.weight");
        g.bias = W(h, "concat4._layer.bias"); g.C = sb.Y4; g.ldc = h->dlow; g.N = h->dlow; g.K = h->dmid;
        g.goff = h->hl.g4; g.boff = h->hl.b4;
        g.x2 = h->x2; g.range_flag = h->range_flag;
        if (int rc = (run_gemm_h<EPI_CSL, OUT_F32>(h, KC_GEMM_TAIL, g))) return rc;
        }
    }
    {
        ProfScope ps(h, KC_OUT_DDIM);
        OutArgs oa{sb.Y4, W(h, "linear._layer.weight"), W(h, "linear._layer.bias"), hyp_chunk, thyp, x_chunk, e_out,
                   M, h->dlow, h->hl.total, h->hl.go, h->hl.bo,
                   h->c_e[step_idx], h->c_x[step_idx], h->n_x[step_idx], h->n_e[step_idx], rm,
                   nullptr, 0, 0.f, 0.f, 0.f};
        if (h->ddpm && !e_out) {
            oa.ddpm = 1;
            oa.z = h->p_noise[step_idx] ? z_chunk : nullptr;
            oa.c0 = h->p_c0[step_idx];
            oa.c1 = h->p_c1[step_idx];
            oa.sigma = h->p_sigma[step_idx];
        }
        if (d <= 512 && M % T == 0 && tune().out_traj != 2 && (tune().out_traj == 1 || M >= 4096 * 4)) {
            // one wave per trajectory (T tokens) - or per piece of one, the largest divisor of T that still leaves >= 4096 waves -
            // once there are enough tokens to fill the chip that way: one scene (100 trajectories) takes 14.0 instead of
            // 12.7 ms per call with whole trajectories, a 51-episode chunk 150.3 instead of 151.2
            int tpw = T;
            while (tpw > 1 && (M / tpw < 4096 || T % tpw != 0)) --tpw;
            if (tune().out_traj == 1) tpw = T;
            const int nw = M / tpw;
            if (next_step >= 0 && !e_out)
                hipLaunchKernelGGL(out_ddim_traj_kernel<true>, dim3((nw + 3) / 4), dim3(256), bystander_lds(out_ddim_traj_kernel<true>),
                                   h->stream, oa, embed_args(h->thyp + (size_t)next_step * h->hl.total), tpw);
            else
                hipLaunchKernelGGL(out_ddim_traj_kernel<false>, dim3((nw + 3) / 4), dim3(256), bystander_lds(out_ddim_traj_kernel<false>),
                                   h->stream, oa, EmbedArgs{}, tpw);
        } else if (next_step >= 0 && !e_out)
            hipLaunchKernelGGL(out_ddim_kernel<true>, dim3((M + 3) / 4), dim3(256), bystander_lds(out_ddim_kernel<true>),
                               h->stream, oa, embed_args(h->thyp + (size_t)next_step * h->hl.total));
        else
            hipLaunchKernelGGL(out_ddim_kernel<false>, dim3((M + 3) / 4), dim3(256), bystander_lds(out_ddim_kernel<false>),
                               h->stream, oa, EmbedArgs{});
        HIPCHK(h, hipGetLastError());
    }
    return 0;
}

// Episodes per pass of the 50-step loop when nothing is forced: large enough to fill the chip several times over per
// launch, and - for JMID - a whole number of "rounds" of the attention launch: that kernel runs 2 workgroups per CU
// (512 slots) and one episode contributes nhead * ceil(S/128) workgroups, so a chunk of floor(k*512 / that) episodes
// leaves no partially filled last round (20 -> 51 episodes: +15 % attention throughput on BASELINE cfg3).
int auto_chunk(const jmid_ctx* h, int E, int tokens_per_episode) {
    const long max_tokens = 65536;
    if (h->net_kind == JMID_NET_JMID) {
        const long bpe = (long)h->nhead * ((tokens_per_episode + 127) / 128);
        for (int k = 4; k >= 1; --k) {
            const long c = (k * 512L) / bpe;
            if (c >= 1 && c * tokens_per_episode <= max_tokens) return (int)std::min<long>(c, E);
        }
    }
    long c = max_tokens / std::max(1, tokens_per_episode);
    if (c < 1) c = 1;
    return (int)std::min<long>(c, E);
}

// The chunks of a call: `c` episodes each (jmid_set_chunk_episodes, or auto_chunk).  A short ragged tail (less than a
// quarter of a chunk, e.g. 256 = 5 x 51 + 1) would run all 50 steps at single-scene latency, so it is spread over the
// full chunks instead (52 + 4 x 51) - only with the automatic size: a forced size is taken literally.
std::vector<int> plan_chunks(const jmid_ctx* h, int E, int tokens_per_episode) {
    int c = h->chunk_eps > 0 ? std::min(E, h->chunk_eps) : auto_chunk(h, E, tokens_per_episode);
    if (h->chunk_eps <= 0 && h->lanes >= 2 && E >= 2 && tune().graph != 1) {     // (a captured loop is a one-chunk call)
        // Two chunks in flight want an EVEN number of chunks of equal size.  A batch that fits one chunk is split in two halves: its
        // kernels do not fill the chip, and two half-size launches side by side finish 5-13 % sooner than one (4 / 8 / 16 / 32 / 48
        // episodes: 23.3 -> 22.2, 36.0 -> 31.8, 60.9 -> 58.0, 111.3 -> 97.2, 140.5 -> 132.5 ms per call; tools/small_batch_lanes.py).
        // A larger batch runs as the smallest even number of chunks that fit, balanced: 256 episodes = 4 x 43 + 2 x 42 instead of
        // 52 + 4 x 51 (an odd count leaves the last chunk alone on the chip) or, in JMID_PREC_F16MX until round 4, 10 x 26 -
        // with the leaner kernels of that round's last session the half-size chunks lost their edge: 256 episodes 564 -> 540 ms,
        // 512: 1142 -> 1086, 160: 356 -> 347, 104: 230 -> 224 (F16MX); F16X2 / F16X3 within 0.4 % either way (tools/chunk_fine.py).
        // The split-KV factor of a call does not depend on its chunk plan (run_network), so neither do the results.
        if (E <= c) {
            // ... unless the whole batch is at most 2 560 tokens in F16MX at d_model 512: as ONE chunk its out-projection / linear2 launches
            // carry the LayerNorm and the split-KV merge (gemm_small.hpp, OUT_LNX - only while nothing else of the handle is in flight),
            // nine launches less per denoise step: two cfg2 scenes (2 400 tokens) 14.45 -> 13.90 ms per call, 12.63 with the split-KV factor chosen for that launch (run_network).  Three (3 600 tokens, 456
            // workgroups of that kernel) are better off as 2 + 1 side by side: 16.42 against 16.90 (profiles/r05s_lnx_two_per_cu.log)
            const bool lnx_call = h->mx && h->d == jmid::GLN_BN && !h->lnx_off && (long)E * tokens_per_episode <= 2560 &&
                                  ((long)E * tokens_per_episode + 63) / 64 * 8 <= 2L * tune().cus &&
                                  tune().small_lnx != 2 && tune().small_lnx2 != 2 && tune().gemm_small != 1 && tune().gemm_h_variant == 0;
            c = lnx_call ? E : (E + 1) / 2;
        } else {
            int n = (E + c - 1) / c;
            if ((n & 1) && n < E) ++n;      // (never more chunks than episodes: c == 1 with an odd E stays at E chunks of one)
            std::vector<int> sizes(n, E / n);
            for (int i = 0; i < E % n; ++i) sizes[i] += 1;
            return sizes;
        }
    }
    std::vector<int> sizes(E / c, c);
    const int tail = E % c;
    if (tail) {
        if (h->chunk_eps > 0 || sizes.empty() || tail * 4 >= c || (tail + sizes.size() - 1) / sizes.size() > (size_t)c / 8)
            sizes.push_back(tail);
        else
            for (int i = 0; i < tail; ++i) sizes[i % sizes.size()] += 1;
    }
    return sizes;
}

// Device-mode calls read and write the caller's buffers on the handle's private stream.  They are ordered against
// the stream the caller works on (jmid_set_caller_stream; default: the legacy null stream): the handle's stream waits
// for everything the caller enqueued before the call, and the caller's stream waits for the call's last kernel, so
// neither a producer kernel of an input nor a consumer (or the allocator's reuse) of an output can race with it.
int order_in(jmid_ctx* h, int mem) {
    if (mem != JMID_MEM_DEVICE || h->chained) return 0;
    HIPCHK(h, hipEventRecord(h->ev_in, h->caller_stream));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_in, 0));
    return 0;
}
int order_out(jmid_ctx* h, int mem) {
    if (mem != JMID_MEM_DEVICE || h->chained) return 0;
    HIPCHK(h, hipEventRecord(h->ev_out, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->caller_stream, h->ev_out, 0));
    return 0;
}

int check_ready(jmid_ctx* h) {
    if (!h) return JMID_EINVAL;
    if (!h->finalized) return fail(h, JMID_ENOWEIGHT, "jmid_finalize_weights has not been called");
    if (h->beta.empty() || !h->thyp) return fail(h, JMID_EINVAL, "jmid_set_ddim_table has not been called");
    return 0;
}

int run_network(jmid_ctx* h, int E, int A, int K, int T, const float* x_in, const float* ctx, const float* p0, float dt,
                int precision, int single_step, float* vel_out, float* pos_out, float* e_out, int mem,
                const float* z_in) {
    if (int rc = check_ready(h)) return rc;
    if (E <= 0 || A <= 0 || K <= 0 || T <= 0) return fail(h, JMID_EINVAL, "E, A, K, T must be positive");
    if (T > 24) return fail(h, JMID_EINVAL, "T exceeds the positional-encoding table (max_len=24, diffusion.py:116-118)");
    if (precision != JMID_PREC_F32 && precision != JMID_PREC_F16X3 && precision != JMID_PREC_F16X2 && precision != JMID_PREC_F16MX)
        return fail(h, JMID_EINVAL, "precision must be JMID_PREC_F32, JMID_PREC_F16X3, JMID_PREC_F16X2 or JMID_PREC_F16MX (JMID_PREC_F16 is not built)");
    h->mx = precision == JMID_PREC_F16MX;
    h->x2 = precision == JMID_PREC_F16X2 || h->mx;
    if (precision != JMID_PREC_F32 && !h->weights_in_half_range && ++h->erange_calls)
        return fail(h, JMID_ERANGE, "a weight exceeds the fp16 range: use JMID_PREC_F32");
    if (!x_in || !ctx) return fail(h, JMID_EINVAL, "null input");
    if (pos_out && !p0) return fail(h, JMID_EINVAL, "pos_out requested without p0");
    h->last_pos = nullptr;     // (the staging buffer is about to be reused)
    if (single_step < 0 && h->ddpm && !z_in) return fail(h, JMID_EINVAL, "DDPM table installed: use jmid_denoise_ddpm (needs z)");
    if (single_step < 0 && !h->ddpm && z_in) return fail(h, JMID_EINVAL, "jmid_denoise_ddpm needs jmid_set_ddpm_table");
    HIPCHK(h, hipSetDevice(h->device));
    TuneScope tune_scope(&h->tune);
    if (int rc = order_in(h, mem)) return rc;
    const size_t R = (size_t)E * K * A, M = R * T, EA = (size_t)E * A;
    const std::vector<int> chunk_sizes = plan_chunks(h, E, K * A * T);
    std::vector<int> chunk_start(chunk_sizes.size(), 0);
    for (size_t i = 1; i < chunk_sizes.size(); ++i) chunk_start[i] = chunk_start[i - 1] + chunk_sizes[i - 1];
    const int Ec = *std::max_element(chunk_sizes.begin(), chunk_sizes.end());
    const size_t Mc = (size_t)Ec * K * A * T;
    // Split-KV factor of the attention launches: chosen ONCE per call from the automatic chunk size, never from the
    // chunk at hand - a ragged last chunk or a forced chunk size must not change the order in which a sequence's keys
    // are summed (results are bit-identical for every chunking of the same call).
    int ns_call = 1;
    if (h->net_kind == JMID_NET_JMID && precision != JMID_PREC_F32 && h->d / h->nhead == 128) {
        const int S = K * A * T;
        // sized for ONE launch of the default plan (two chunks in flight: a batch that fits one chunk runs as two halves) - a
        // function of the call's shape only, whatever the chunk size or number of lanes actually set
        const int c_auto = auto_chunk(h, E, S);
        // (a batch of at most 2 560 tokens in F16MX is ONE launch by default - plan_chunks: two cfg2 scenes take 3 key ranges x 80 blocks,
        //  13.43 ms per call, where the 6 x 80 of the halves' choice take 14.14-14.37; shape and mode only, no knob: the bits of a call
        //  must not depend on one)
        const bool one_launch = h->mx && h->d == jmid::GLN_BN && (long)E * S <= 2560;
        ns_call = attn_pick_nsplit(((S + 127) / 128) * h->nhead * (E >= 2 ? (one_launch ? E : (c_auto + 1) / 2) : 1), S);
        if (tune().attn_nsplit > 0) ns_call = std::min(tune().attn_nsplit, (S + 31) / 32);
    }
    // ---- workspace
    size_t io_off;
    {
        Carver c(nullptr);
        c.take(M * 2);                 // x_cur
        c.take(EA * h->ctx_dim);       // ctx
        c.take(EA * h->hl.total);      // hyp
        c.take(EA * 2);                // p0
        c.take(M * 2);                 // e / pos staging
        if (z_in && mem == JMID_MEM_HOST) c.take(M * 2 * h->beta.size());   // DDPM noise
        io_off = c.off;
    }
    const SeqGeom sg_full = seq_geom(h, Ec, A, K, T);
    // Independent chunks run `lanes` at a time on separate streams: the partially filled last round of one chunk's
    // kernels and its bandwidth-bound kernels overlap with another chunk's MFMA kernels.  Each lane has its own step
    // workspace; results do not depend on the number of lanes.
    const int nchunks = (int)chunk_sizes.size();
    const int lanes = single_step < 0 ? std::max(1, std::min(h->lanes, nchunks)) : 1;
    // the small-launch GEMMs (gemm_small.hpp: one workgroup per CU, most of its LDS) only while one chunk is in flight
    struct SmallNow {
        Tuning& t;
        SmallNow(Tuning& t_, int v, int one) : t(t_) { t.small_now = v; t.one_chunk = one; }
        ~SmallNow() { t.small_now = 1; t.one_chunk = 1; }
    } small_now_scope(h->tune, lanes == 1 || h->tune.small_lanes == 1 ? 1 : h->tune.small_lanes == 2 ? 2 : 0, nchunks == 1 && tune().graph != 1);      // (a captured loop would replay the launch tags of OUT_LNX)
    const size_t lane_floats = step_ws_floats(h, Mc, precision, sg_full, ns_call, nullptr, nullptr);
    const size_t need = io_off + lanes * lane_floats;
    if (int rc = ensure_arena(h, need)) return rc;
    Carver c(h->arena);
    float* x_cur = c.take(M * 2);
    float* ctx_d = c.take(EA * h->ctx_dim);
    float* hyp = c.take(EA * h->hl.total);
    float* p0_d = c.take(EA * 2);
    float* stage = c.take(M * 2);
    const float* z_use = z_in;
    if (z_in && mem == JMID_MEM_HOST) {
        float* zd = c.take(M * 2 * h->beta.size());
        HIPCHK(h, hipMemcpyAsync(zd, z_in, M * 2 * h->beta.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
        z_use = zd;
    }
    StepBuffers sbs[jmid_ctx::kMaxLanes];
    for (int l = 0; l < lanes; ++l) step_ws_floats(h, Mc, precision, sg_full, ns_call, &sbs[l], h->arena + io_off + l * lane_floats);
    const StepBuffers& sb = sbs[0];
    if (precision != JMID_PREC_F32) {
        HIPCHK(h, hipMemsetAsync(h->range_flag, 0, sizeof(int), h->stream));
        for (int l = 0; l < lanes; ++l) HIPCHK(h, hipMemsetAsync(sbs[l].ln_xchg, 0, kLnxWords * sizeof(unsigned), h->stream));
        for (int l = 0; l < lanes; ++l)
            if (sbs[l].Vth && sg_full.Spad != sg_full.S) {  // padding keys of V^T must be finite (they meet P = 0)
                HIPCHK(h, hipMemsetAsync(sbs[l].Vth, 0, sbs[l].vt_elems * sizeof(half_t), h->stream));
                HIPCHK(h, hipMemsetAsync(sbs[l].Vtl, 0, sbs[l].vt_elems * sizeof(half_t), h->stream));
            }
    }

    const hipMemcpyKind kin = mem == JMID_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    const hipMemcpyKind kout = mem == JMID_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    HIPCHK(h, hipMemcpyAsync(x_cur, x_in, M * 2 * sizeof(float), kin, h->stream));
    const float* ctx_use = ctx;
    if (mem == JMID_MEM_HOST) {
        HIPCHK(h, hipMemcpyAsync(ctx_d, ctx, EA * h->ctx_dim * sizeof(float), kin, h->stream));
        ctx_use = ctx_d;
    }
    const float* p0_use = p0;
    if (p0 && mem == JMID_MEM_HOST) {
        HIPCHK(h, hipMemcpyAsync(p0_d, p0, EA * 2 * sizeof(float), kin, h->stream));
        p0_use = p0_d;
    }
    // ---- ctx part of the four hyper nets, once per call (ctx is constant over the denoise steps)
    {
        GemmArgs g{};
        g.A = ctx_use; g.lda = h->ctx_dim; g.W = h->Whyp; g.ldw = h->ctx_dim; g.bias = h->bhyp; g.C = hyp;
        g.ldc = h->hl.total; g.M = (int)EA; g.N = h->hl.total; g.K = h->ctx_dim;
        if (int rc = run_gemm<EPI_BIAS>(h, KC_HYPER, g)) return rc;
    }
    const int n_steps = (int)h->beta.size();
    if (lanes > 1) {   // everything enqueued so far (inputs, hyper nets, memsets) precedes the extra lanes as well
        HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
        for (int l = 1; l < lanes; ++l) HIPCHK(h, hipStreamWaitEvent(h->lane_stream[l - 1], h->ev_fork, 0));
    }
    // Opt-in (jmid_set_tuning "graph" = 1) for one-chunk calls: the whole denoise loop - n_steps x ~28 dependent launches on
    // workspace buffers only - is captured into a hipGraph the second time a shape is seen and replayed afterwards: one
    // graph launch instead of ~1400 kernel launches per call, bit-identical.  Measured on MI355X / ROCm 7.2
    // (tools/graph_latency.py): it does not pay - the GPU-side time is the same chain of kernels (a kernel boundary costs
    // the same inside a graph) and the replay itself is slower than the eager launches that run ahead of the GPU: one
    // scene 13.51 vs 13.08 ms per call, 4 scenes 28.40 vs 28.29, 8 scenes equal.  Off by default.
    // Inputs / outputs (copies, hyper-net GEMM, integrator) stay outside the graph.
    jmid_ctx::LoopGraph* lg = nullptr;
    bool capturing = false;
    if (single_step < 0 && lanes == 1 && nchunks == 1 && !h->prof_mask && !z_use && !h->ddpm && tune().graph == 1 &&
        tune().bystander_lds == 0) {
        const std::string key = std::to_string(E) + "," + std::to_string(A) + "," + std::to_string(K) + "," + std::to_string(T) +
                                "," + std::to_string(precision);
        lg = &h->graphs[key];
        if (lg->exec && lg->arena != h->arena) {        // never true today (ensure_arena drops the graphs); cheap to keep
            hipGraphExecDestroy(lg->exec);
            lg->exec = nullptr;
        }
        if (lg->exec) {
            HIPCHK(h, hipGraphLaunch(lg->exec, h->stream));
            ++h->graph_replays;
        } else if (lg->warm) {
            HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            capturing = true;
        }
    }
    for (int c0 = 0; c0 < nchunks && !(lg && lg->exec); c0 += lanes) {
        if (single_step >= 0) {
            const int e0 = chunk_start[c0], ec = chunk_sizes[c0];
            float* eo = stage + (size_t)e0 * K * A * T * 2;
            if (int rc = net_step(h, sb, ec, A, K, T, single_step, x_cur + (size_t)e0 * K * A * T * 2,
                                  hyp + (size_t)e0 * A * h->hl.total, eo, precision))
                return rc;
            continue;
        }
        // the steps of the chunks of this round are enqueued alternately so that all queues stay fed
        for (int i = 0; i < n_steps; ++i) {
            for (int l = 0; l < lanes; ++l) {
                if (c0 + l >= nchunks) break;
                const int el = chunk_start[c0 + l], ec = chunk_sizes[c0 + l];
                float* xc = x_cur + (size_t)el * K * A * T * 2;
                const float* hc = hyp + (size_t)el * A * h->hl.total;
                const float* zc = z_use ? z_use + ((size_t)i * M + (size_t)el * K * A * T) * 2 : nullptr;
                if (l > 0) std::swap(h->stream, h->lane_stream[l - 1]);   // net_step launches on h->stream
                const int rc = net_step(h, sbs[l], ec, A, K, T, i, xc, hc, nullptr, precision, zc, tune().fuse_embed && i > 0,
                                        tune().fuse_embed && i + 1 < n_steps ? i + 1 : -1);
                if (l > 0) std::swap(h->stream, h->lane_stream[l - 1]);
                if (rc) {
                    if (capturing) {
                        hipGraph_t dead = nullptr;
                        (void)hipStreamEndCapture(h->stream, &dead);
                        if (dead) hipGraphDestroy(dead);
                    }
                    // the lanes share the one arena: what they already hold must have drained before the caller (the f32
                    // rerun, ensure_arena, jmid_destroy) touches it again on h->stream
                    sync_lanes(h);
                    return rc;
                }
            }
        }
    }
    if (capturing) {
        hipGraph_t graph = nullptr;
        HIPCHK(h, hipStreamEndCapture(h->stream, &graph));
        hipError_t ge = hipGraphInstantiate(&lg->exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (ge != hipSuccess) {
            lg->exec = nullptr;
            return fail(h, JMID_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ge));
        }
        lg->arena = h->arena;
        HIPCHK(h, hipGraphLaunch(lg->exec, h->stream));
        ++h->graph_replays;
    } else if (lg && !lg->exec) {
        lg->warm = true;
    }
    for (int l = 1; l < lanes; ++l) {
        HIPCHK(h, hipEventRecord(h->ev_join[l - 1], h->lane_stream[l - 1]));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join[l - 1], 0));
    }
    if (single_step >= 0) {
        HIPCHK(h, hipMemcpyAsync(e_out, stage, M * 2 * sizeof(float), kout, h->stream));
    } else {
        if (vel_out) HIPCHK(h, hipMemcpyAsync(vel_out, x_cur, M * 2 * sizeof(float), kout, h->stream));
        if (p0_use) {      // integrated whenever p0 is given: the positions stay in the workspace for jmid_topk(pos = NULL)
            {
                ProfScope ps(h, KC_INTEGRATE);
                const int n = (int)R * 2;
                hipLaunchKernelGGL(integrate_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, x_cur, p0_use,
                                   stage, (int)R, T, A, K * A, dt);
                HIPCHK(h, hipGetLastError());
            }
            h->last_pos = stage;
            h->last_pos_dims[0] = E; h->last_pos_dims[1] = A; h->last_pos_dims[2] = K; h->last_pos_dims[3] = T;
            if (pos_out) HIPCHK(h, hipMemcpyAsync(pos_out, stage, M * 2 * sizeof(float), kout, h->stream));
        }
    }
    if (int rc = order_out(h, mem)) return rc;
    if (h->chained) return 0;          // (jmid_predict reads the range flag with its one download)
    if (precision != JMID_PREC_F32) {
        // an activation outside the fp16 range poisons the split operands: report it instead of returning garbage
        int flag = 0;
        HIPCHK(h, hipMemcpyAsync(&flag, h->range_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (flag) return flagged_call(h, flag);
    } else if (mem == JMID_MEM_HOST) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return 0;
}


// A call whose range flag came back set.  Bit 1 (gemm_small.hpp, OUT_LNX): a workgroup gave up waiting for a partner - nothing to do with
// the arithmetic: the handle drops that kernel for good and the caller repeats the call in the SAME precision (JMID_ETIMEOUT).  Otherwise
// bit 0: an activation left the fp16 range (JMID_ERANGE: repeat in JMID_PREC_F32).  Either way the outputs are undefined.
int flagged_call(jmid_ctx* h, int flag) {
    h->last_pos = nullptr;     // the integrated positions are poisoned too: jmid_topk(pos = NULL) must not rank them
    if (flag & 2) {
        h->lnx_off = true;
        ++h->lnx_timeouts;
        return fail(h, JMID_ETIMEOUT, "a workgroup of a one-launch GEMM + LayerNorm gave up waiting for its partners (not all workgroups of the launch "
                                      "were resident): this handle now runs the unfused kernels - repeat the call in the same precision");
    }
    ++h->erange_calls;
    return fail(h, JMID_ERANGE, "an activation left the fp16 range in JMID_PREC_F16X3 / F16X2 / F16MX: rerun with JMID_PREC_F32");
}

int launch_episode_metrics(jmid_ctx* h, const float* pos, const float* gt, float* out, int E, int K, int A, int T) {
    ProfScope ps(h, KC_METRICS);
    hipLaunchKernelGGL(episode_metrics_kernel, dim3(E), dim3(256), 0, h->stream, pos, gt, out, K, A, T);
    HIPCHK(h, hipGetLastError());
    return 0;
}

}  // namespace jmid_host

