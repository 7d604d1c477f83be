// libjmid_hip.so  --  C ABI (include/jmid_hip.h) + host orchestration of the HIP kernels.
// gfx950 only.  No CPU fallback: every entry point that computes needs a HIP device.
#include "../../include/jmid_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "attn_f16x3.hpp"
#include "attn_f32.hpp"
#include "common.hpp"
#include "elementwise.hpp"
#include "encoder.hpp"
#include "gemm_f16x3.hpp"
#include "gemm_ln_f16x3.hpp"
#include "gemm_ln2_mx.hpp"
#include "gemm_small.hpp"
#include "gemm_f32.hpp"
#include "kde.hpp"
#include "tail_f16x3.hpp"

using namespace jmid;

namespace {

enum KClass {
    KC_GEMM_QKV = 0,
    KC_GEMM_OUT,
    KC_GEMM_FF1,
    KC_GEMM_FF2,
    KC_GEMM_TAIL,
    KC_ATTN,
    KC_ADD_LN,
    KC_EMBED,
    KC_OUT_DDIM,
    KC_HYPER,
    KC_ENCODER,
    KC_INTEGRATE,
    KC_METRICS,
    KC_VTRANS,
    KC_TOPK,
    KC_COUNT
};
const char* kClassNames[KC_COUNT] = {"gemm_qkv", "gemm_attn_out", "gemm_ff1", "gemm_ff2", "gemm_tail", "attention",
                                     "add_layernorm", "embed", "out_ddim", "hyper", "encoder", "integrate",
                                     "episode_metrics", "v_transpose", "kde_topk"};

struct DevBuf {
    float* p = nullptr;
    size_t n = 0;
};

struct EvPair {
    hipEvent_t a, b;
};

struct HalfPair {
    half_t* hi = nullptr;
    half_t* lo = nullptr;
};

}  // namespace

struct jmid_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    static constexpr int kMaxLanes = 4;
    hipStream_t lane_stream[kMaxLanes - 1] = {nullptr, nullptr, nullptr};   // extra lanes of the chunk loop
    hipEvent_t ev_fork = nullptr, ev_join[kMaxLanes - 1] = {nullptr, nullptr, nullptr};
    // chunks in flight at once, 1..4 (jmid_set_tuning "lanes").  Two by default: the partially filled last round of one
    // chunk's kernels and its bandwidth-bound kernels overlap with the other chunk's MFMA kernels (+2-4 % traj/s), and the
    // results are bit-identical to one chunk in flight.  (They were not in round 1: a row-wise kernel sharing a CU with
    // attention workgroups of the other lane computed a few wrong values per run - packed-fp32 instructions with crossed
    // operand selects, which the library is no longer built with; build.py, docs/NOTEBOOK.md section 3.)
    int lanes = 2;
    Tuning tune;         // jmid_set_tuning knobs of THIS handle (installed per call by TuneScope)
    hipStream_t caller_stream = nullptr;   // stream device-mode buffers are ordered on (jmid_set_caller_stream)
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    // captured denoise loops of small calls (one chunk): key = (E, A, K, T, precision) -> executable graph
    struct LoopGraph {
        hipGraphExec_t exec = nullptr;
        char* arena = nullptr;      // the workspace the graph's kernels point into
        bool warm = false;          // the loop ran eagerly once with this key (per-device kernel attributes are set)
    };
    std::map<std::string, LoopGraph> graphs;
    int64_t graph_replays = 0;
    // the positions of the most recent jmid_denoise (integrated into the workspace whether or not they were copied out): what
    // jmid_topk ranks when it is given no pos pointer
    const float* last_pos = nullptr;
    int last_pos_dims[4] = {0, 0, 0, 0};     // E, A, K, T
    char* kde_ws = nullptr;                  // jmid_topk's own workspace (it must not move the arena last_pos points into)
    size_t kde_ws_bytes = 0;
    // jmid_predict: pinned host staging + device I/O buffers of the chained call, grown on demand
    char* pin = nullptr;
    size_t pin_bytes = 0;
    char* io_dev = nullptr;
    size_t io_dev_bytes = 0;
    bool chained = false;       // the running run_network is a stage of jmid_predict: no caller-stream ordering, no flag round trip
    int64_t erange_calls = 0;   // calls on this handle that ended with JMID_ERANGE (jmid_erange_count)
    int x2 = 0;          // the running call is JMID_PREC_F16X2 (set by the entry points, read by the launch helpers)
    int net_kind = 1, ctx_dim = 256, tf_layer = 3, nhead = 4, hist_len = 6;
    int d = 512, ff = 1024, dmid = 256, dlow = 128, H = 128;
    HyperLayout hl;
    std::map<std::string, std::vector<size_t>> expected;  // name -> shape
    std::map<std::string, DevBuf> w;
    std::map<std::string, HalfPair> wsplit;  // hi/lo fp16 planes of the GEMM weights (F16X3 path)
    struct W8Image { unsigned char* p = nullptr; };
    std::map<std::string, W8Image> w8;       // JMID_PREC_F16MX: fp8 images of W_lo (w8_image_kernel), keyed like wsplit
    int mx = 0;          // the running call is JMID_PREC_F16MX (x2 is set as well: everything not on the fp8 path runs as F16X2)
    std::map<std::string, HalfPair> w16;     // k16-panel copies of out_proj / linear2 for the fused GEMM + LayerNorm
    int* range_flag = nullptr;               // device word: an fp16 operand left the fp16 range
    bool weights_in_half_range = true;
    bool finalized = false;
    // derived device buffers
    float* pe = nullptr;
    float* Whyp = nullptr;
    float* bhyp = nullptr;
    float* thyp = nullptr;  // [n_steps, hl.total]
    std::vector<float> time_w;  // host [hl.total][3] time columns of the hyper nets
    float* lstmT[3][3] = {{nullptr}};  // [hist, edge_ped, edge_robot] x [WihT, WhhT, b]
    float* attW1T = nullptr;
    float* attW2T = nullptr;
    // sampler step table (host): DDIM coefficients, or DDPM ones when ddpm is set
    std::vector<float> beta, c_e, c_x, n_x, n_e;
    bool ddpm = false;
    std::vector<float> p_c0, p_c1, p_sigma;
    std::vector<int> p_noise;
    // workspace arena
    char* arena = nullptr;
    size_t arena_bytes = 0;
    // I/O staging
    int chunk_eps = 0;
    // profiling
    uint32_t prof_mask = 0;
    std::vector<EvPair> prof_ev[KC_COUNT];
    std::vector<EvPair> ev_pool;
    double prof_ms[KC_COUNT] = {0};
    int64_t prof_n[KC_COUNT] = {0};
    std::string err;
};

namespace {

thread_local std::string g_err;

int fail(jmid_ctx* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    g_err = msg;
    return code;
}

#define HIPCHK(h, expr)                                                                               \
    do {                                                                                              \
        hipError_t e__ = (expr);                                                                      \
        if (e__ != hipSuccess)                                                                        \
            return fail(h, JMID_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__));            \
    } while (0)

struct ProfScope {
    jmid_ctx* h;
    int cls;
    bool on;
    EvPair ev;
    ProfScope(jmid_ctx* h_, int cls_) : h(h_), cls(cls_), on((h_->prof_mask >> cls_) & 1u) {
        if (on) {
            if (!h->ev_pool.empty()) {
                ev = h->ev_pool.back();
                h->ev_pool.pop_back();
            } else {
                hipEventCreate(&ev.a);
                hipEventCreate(&ev.b);
            }
            hipEventRecord(ev.a, h->stream);
        }
    }
    ~ProfScope() {
        if (on) {
            hipEventRecord(ev.b, h->stream);
            h->prof_ev[cls].push_back(ev);
        }
    }
};

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

size_t numel(const std::vector<size_t>& s) {
    size_t n = 1;
    for (size_t v : s) n *= v;
    return n;
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

const float* W(jmid_ctx* h, const std::string& name) { return h->w[name].p; }

void drop_graphs(jmid_ctx* h) {
    for (auto& kv : h->graphs)
        if (kv.second.exec) hipGraphExecDestroy(kv.second.exec);
    h->graphs.clear();
}

int ensure_arena(jmid_ctx* h, size_t bytes) {
    if (bytes <= h->arena_bytes) return 0;
    drop_graphs(h);
    h->last_pos = nullptr;
    if (h->arena) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipFree(h->arena));
        h->arena = nullptr;
        h->arena_bytes = 0;
    }
    hipError_t e = hipMalloc((void**)&h->arena, bytes);
    if (e != hipSuccess) return fail(h, JMID_ENOMEM, "workspace allocation of " + std::to_string(bytes) + " bytes failed");
    h->arena_bytes = bytes;
    return 0;
}

struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(char* b) : base(b) {}
    float* take(size_t nfloats) {
        float* p = reinterpret_cast<float*>(base ? base + off : nullptr);
        off += ((nfloats * sizeof(float) + 255) / 256) * 256;
        return p;
    }
};

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


// ---------------------------------------------------------------------------------------------- launch helpers
template <int EPI>
int run_gemm(jmid_ctx* h, int cls, GemmArgs& g) {
    if (g.K % GEMM_BK != 0) return fail(h, JMID_EINVAL, "GEMM K must be a multiple of 32");
    ProfScope ps(h, cls);
    HIPCHK(h, launch_gemm_f32<EPI>(g, h->stream));
    return 0;
}

template <int EPI, int OUT>
int run_gemm_h(jmid_ctx* h, int cls, GemmHArgs& g) {
    if (g.K % GEMMH_BK != 0) return fail(h, JMID_EINVAL, "GEMM K must be a multiple of 32");
    g.range_flag = h->range_flag;
    g.x2 = h->x2;
    ProfScope ps(h, cls);
    HIPCHK(h, (launch_gemm_h<EPI, OUT>(g, h->stream)));
    return 0;
}

// out_proj / linear2 + residual + LayerNorm as ONE small launch (gemm_small.hpp, OUT_LN); g carries the GEMM, the ln_* fields the tail
int run_gemm_ln_small(jmid_ctx* h, int cls, GemmHArgs& g) {
    g.range_flag = h->range_flag;
    g.x2 = h->x2;
    ProfScope ps(h, cls);
    HIPCHK(h, (launch_gemm_small<EPI_BIAS, OUT_LN>(g, 2, h->stream)));
    return 0;
}

// JMID_PREC_F16MX: hand the GEMM the fp8 image of this weight's lo plane (the kernels that have no fp8 path ignore it)
void set_w8(jmid_ctx* h, GemmHArgs& g, const std::string& name) {
    g.W8 = nullptr;
    if (!h->mx) return;
    auto it = h->w8.find(name);
    if (it == h->w8.end()) return;
    g.W8 = it->second.p;
}

// bf8 image of W_lo for a device-resident fp32 weight [N, K] (N % 32 == 0, K % 64 == 0)
int make_w8(jmid_ctx* h, const float* dW, int N, int K, jmid_ctx::W8Image* out) {
    HIPCHK(h, hipMalloc((void**)&out->p, (size_t)N * K));
    hipLaunchKernelGGL(w8_image_kernel, dim3(256), dim3(256), 0, h->stream, dW, out->p, N, K, kWScale);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int run_add_ln(jmid_ctx* h, float* X, const float* Y, const float* gm, const float* bt, int M, int d,
               half_t* Xh = nullptr, half_t* Xl = nullptr, bool mxv2 = false, int no_lo_out = 0) {
    ProfScope ps(h, KC_ADD_LN);
    if (mxv2) {      // gemm_ln2_mx.hpp: byte lo plane, that file's summation order (d == 512); 4 rows per wave
        hipLaunchKernelGGL(add_ln2_kernel, dim3((M + 15) / 16), dim3(256), bystander_lds(add_ln2_kernel), h->stream, Y, gm, bt, M, 1e-5f,
                           Xh, reinterpret_cast<unsigned char*>(Xl), no_lo_out, h->range_flag);
        HIPCHK(h, hipGetLastError());
        return 0;
    }
    const int rows_per_block = 4;
    dim3 grid((M + rows_per_block - 1) / rows_per_block);
    const int vpl = (d + 255) / 256;
    const bool planes = Xh != nullptr;   // split-fp16 mode: the residual stream lives only in its planes
#define JMID_LN(V)                                                                                                    \
    if (planes) hipLaunchKernelGGL((add_ln_kernel<V, true>), grid, dim3(256), bystander_lds(add_ln_kernel<V, true>), h->stream, X, Y, gm, bt, M, d, 1e-5f, Xh, Xl); \
    else hipLaunchKernelGGL((add_ln_kernel<V, false>), grid, dim3(256), bystander_lds(add_ln_kernel<V, false>), h->stream, X, Y, gm, bt, M, d, 1e-5f, Xh, Xl);
    switch (vpl) {
        case 1: JMID_LN(1) break;
        case 2: JMID_LN(2) break;
        case 3:
        case 4: JMID_LN(4) break;
        default: return fail(h, JMID_EINVAL, "d_model too large for add_ln");
    }
#undef JMID_LN
    HIPCHK(h, hipGetLastError());
    return 0;
}

struct StepBuffers {
    float *X, *QKV, *ATT, *Y, *H1, *Y3, *Y4;
    // F16X3 path: hi/lo planes
    half_t *Xh, *Xl, *Qh, *Ql, *Kh, *Kl, *Vh, *Vl, *Vth, *Vtl, *Ah, *Al, *H1h, *H1l, *Y3h, *Y3l;
    size_t vt_elems;
    int attn_nsplit;          // split-KV factor of the attention launch (1 = off)
    float *Opart, *MLpart;
    unsigned* ln_cnt;         // arrival counters of the small-launch GEMM + LayerNorm (gemm_small.hpp, OUT_LN): kLnCounters words
};
constexpr size_t kLnCounters = 256;

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
    s.Y = c.take((Mc + 63) / 64 * 64 * h->d);     // (whole 64-row tiles: the hand-off layout of gemm_small.hpp's LayerNorm tail)
    s.ln_cnt = reinterpret_cast<unsigned*>(c.take(kLnCounters));
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
    bool tail_fused = false;
    const int R = Ec * K * A, M = R * T;
    const int d = h->d, ff = h->ff;
    const float* thyp = h->thyp + (size_t)step_idx * h->hl.total;
    RowMap rm{T, A, K * A};
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
            // small launches (one scene ... a few): GEMM + residual + LayerNorm in one kernel, the LayerNorm by the last-arriving
            // workgroup of each 64-row tile (gemm_small.hpp; bit-identical to the pair below it replaces, two launches per layer fewer).
            // F16MX: only with the byte lo plane of the second-generation LayerNorm (mxv2), whose order the tail reproduces
            const bool ln_small = d == GLN_BN && (!h->mx || mxv2);
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
                if (ln_small && small_ln_fits(M, g.K)) {
                    g.ln_gamma = W(h, p + ".norm1.weight"); g.ln_beta = W(h, p + ".norm1.bias"); g.ln_xh = sb.Xh; g.ln_xl = sb.Xl;
                    g.ln_xl8 = Xl8; g.ln_cnt = sb.ln_cnt; g.ln_eps = 1e-5f; g.ln_no_lo = 0;
                    if (int rc = run_gemm_ln_small(h, KC_GEMM_OUT, g)) return rc;
                } else {
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
                if (ln_small && small_ln_fits(M, g.K)) {
                    g.ln_gamma = W(h, p + ".norm2.weight"); g.ln_beta = W(h, p + ".norm2.bias"); g.ln_xh = sb.Xh; g.ln_xl = sb.Xl;
                    g.ln_xl8 = Xl8; g.ln_cnt = sb.ln_cnt; g.ln_eps = 1e-5f; g.ln_no_lo = mxv2 && l + 1 == h->tf_layer;
                    if (int rc = run_gemm_ln_small(h, KC_GEMM_FF2, g)) return rc;
                } else {
                if (int rc = (run_gemm_h<EPI_BIAS, OUT_F32>(h, KC_GEMM_FF2, g))) return rc;
                if (int rc = run_add_ln(h, sb.X, sb.Y, W(h, p + ".norm2.weight"), W(h, p + ".norm2.bias"), M, d, sb.Xh,
                                        sb.Xl, mxv2, l + 1 == h->tf_layer))
                    return rc;
                }
            }
        }
        // concat3 -> concat4 -> output layer -> sampler update -> next embedding in ONE kernel (tail_f16x3.hpp; bit-identical
        // to the three launches below it replaces) at the shipped width.  Opt-in: it saves two launches per step but runs
        // four waves per CU, and measured slower than the three well-occupied kernels at every batch size (one scene
        // 13.65 vs 13.23 ms per call, a 51-episode chunk +1.3 %; tools/single_scene_sweep.py tail_fuse=2,1)
        tail_fused = d == TAIL_D && h->dmid == TAIL_DM && h->dlow == TAIL_DL && tune().tail_fuse == 1 && !h->mx;   // the fused kernel has no fp8-correction K loop
        if (!tail_fused) {
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
        if (int rc = (run_gemm_h<EPI_CSL, OUT_F32>(h, KC_GEMM_TAIL, g))) return rc;
        }
    }
    {
        ProfScope ps(h, tail_fused ? KC_GEMM_TAIL : KC_OUT_DDIM);
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
        if (tail_fused) {
            const HalfPair& w3 = h->w16["concat3._layer.weight"];
            const HalfPair& w4 = h->w16["concat4._layer.weight"];
            TailArgs ta{sb.Xh, sb.Xl, w3.hi, w3.lo, w4.hi, w4.lo, W(h, "concat3._layer.bias"), W(h, "concat4._layer.bias"),
                        hyp_chunk, thyp, h->hl.total, h->hl.g3, h->hl.b3, h->hl.g4, h->hl.b4, rm, M, h->range_flag};
            const bool en = next_step >= 0 && !e_out;
            // 32-row tiles while 64-row ones would leave CUs idle (a few scenes), 64-row tiles otherwise
            const int rows = tune().tail_rows ? tune().tail_rows : (M < 64 * 256 ? 32 : 64);
            HIPCHK(h, launch_tail(ta, oa, en ? embed_args(h->thyp + (size_t)next_step * h->hl.total) : EmbedArgs{}, en, rows,
                                  h->x2 != 0, h->stream));
        } else if (d <= 512 && M % T == 0 && tune().out_traj != 2 && (tune().out_traj == 1 || M >= 4096 * 4)) {
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
        // Two chunks in flight want two chunks.  A batch that fits one chunk is split in two halves: its kernels do not fill the
        // chip, and two half-size launches side by side finish 5-13 % sooner than one (4 / 8 / 16 / 32 / 48 episodes: 23.3 ->
        // 22.2, 36.0 -> 31.8, 60.9 -> 58.0, 111.3 -> 97.2, 140.5 -> 132.5 ms per call; tools/small_batch_lanes.py).  In
        // JMID_PREC_F16MX larger batches run in half-size chunks too (2 x 26 episodes in flight instead of 51 + 51: -1.2 ... -2.6 %
        // on 104 / 256 / 512 episodes; F16X2 -0.4 %, F16X3 +0.9 %: left alone; tools/chunk26_check.py).  The split-KV factor of a
        // call does not depend on its chunk plan (run_network), so neither do the results.
        if (E <= c) c = (E + 1) / 2;
        else if (h->mx) c = (c + 1) / 2;
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
                const float* z_in = nullptr) {
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
        ns_call = attn_pick_nsplit(((S + 127) / 128) * h->nhead * (E >= 2 ? (c_auto + 1) / 2 : 1), S);
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
        SmallNow(Tuning& t_, int v) : t(t_) { t.small_now = v; }
        ~SmallNow() { t.small_now = 1; }
    } small_now_scope(h->tune, lanes == 1 ? 1 : 0);
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
        for (int l = 0; l < lanes; ++l) HIPCHK(h, hipMemsetAsync(sbs[l].ln_cnt, 0, kLnCounters * sizeof(unsigned), h->stream));
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
        if (flag) ++h->erange_calls;
        if (flag) h->last_pos = nullptr;     // the integrated positions are poisoned too: jmid_topk(pos = NULL) must not rank them
        if (flag) return fail(h, JMID_ERANGE, "an activation left the fp16 range in JMID_PREC_F16X3 / F16X2 / F16MX: rerun with JMID_PREC_F32");
    } else if (mem == JMID_MEM_HOST) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return 0;
}

}  // namespace

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

const char* jmid_last_error(jmid_handle_t h) { return h ? h->err.c_str() : g_err.c_str(); }

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
    bool ok = hipSetDevice(device_id) == hipSuccess && hipStreamCreate(&h->stream) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming) == hipSuccess;
    for (int l = 0; ok && l < jmid_ctx::kMaxLanes - 1; ++l)
        ok = hipStreamCreate(&h->lane_stream[l]) == hipSuccess &&
             hipEventCreateWithFlags(&h->ev_join[l], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        delete h;
        return fail(nullptr, JMID_EHIP, "cannot create a HIP stream");
    }
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
    hipStreamSynchronize(h->stream);
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
        HIPCHK(h, hipMemset(h->range_flag, 0, sizeof(int)));
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
            if (h->dmid == TAIL_DM && h->dlow == TAIL_DL) {
                k16names.push_back("concat3._layer.weight");
                k16names.push_back("concat4._layer.weight");
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
        HIPCHK(h, hipMemset(h->range_flag, 0, sizeof(int)));
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
    {
        ProfScope ps(h, KC_METRICS);
        hipLaunchKernelGGL(episode_metrics_kernel, dim3(E), dim3(256), 0, h->stream, dp, dg, dout, K, A, T);
        HIPCHK(h, hipGetLastError());
    }
    if (mem == JMID_MEM_HOST) {
        HIPCHK(h, hipMemcpyAsync(out, dout, (size_t)E * 4 * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return order_out(h, mem);
}

}  // extern "C"

namespace {
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
}  // namespace

extern "C" {

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
        if (hipMalloc((void**)&h->io_dev, total * 4) != hipSuccess) return fail(h, JMID_ENOMEM, "jmid_predict: device staging allocation failed");
        h->io_dev_bytes = total * 4;
    }
    const size_t pin_need = (in_floats + out_floats) * 4;
    if (pin_need > h->pin_bytes) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->pin) HIPCHK(h, hipHostFree(h->pin));
        h->pin = nullptr;
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
    if (flagged && *reinterpret_cast<const int*>(pout + (o_flag - o_out))) {
        ++h->erange_calls;
        h->last_pos = nullptr;
        return fail(h, JMID_ERANGE, "an activation left the fp16 range in JMID_PREC_F16X3 / F16X2 / F16MX: rerun with JMID_PREC_F32");
    }
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
        {"gemm_h_variant", &Tuning::gemm_h_variant, 0, 6},     // 0 auto, 1..6 force a tile variant of the split GEMM
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
        {"tail_fuse", &Tuning::tail_fuse, 0, 2},               // concat3 -> concat4 -> output -> update in one kernel: 1 on, 0 / 2 off
        {"attn_mx", &Tuning::attn_mx, 0, 3},
        {"out_traj", &Tuning::out_traj, 0, 2},
        {"attn_pf", &Tuning::attn_pf, 0, 2},
        {"mx_ln", &Tuning::mx_ln, 0, 2},
        {"csl_swap", &Tuning::csl_swap, 0, 3},
        {"gemm_small", &Tuning::gemm_small, 0, 2},             // 1: no deep-ring small-launch GEMM (the round-3 64 x 64 / 128 x 128 shapes)
        {"small_ln", &Tuning::small_ln, 0, 2},                 // 2: no fused LayerNorm tail in small launches
        {"small_pn", &Tuning::small_pn, 0, 8},                 // column groups of its XCD tile order: 0 auto
        {"tail_rows", &Tuning::tail_rows, 0, 64},              // row tile of that kernel: 0 auto, 32, 64          // split-KV factor (head_dim 128): 0 auto, 1..16 forced
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
    if (k == "print_occupancy") {   // diagnostics: resident workgroups per CU of the main kernels
        int n = -1;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_f16x3_dma_kernel<false>, 256, ATT_DMA_LDS);
        fprintf(stderr, "attn_f16x3_dma_kernel: %d workgroups/CU (LDS %zu B)\n", n, ATT_DMA_LDS);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_f16x3_dma256_kernel<EPI_BIAS, OUT_F32>, 512, DMA2_LDS_BYTES);
        fprintf(stderr, "gemm_f16x3_dma256_kernel: %d workgroups/CU (LDS %zu B)\n", n, DMA2_LDS_BYTES);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_f16x3_kernel<2, 2, EPI_BIAS, OUT_F32>, 256,
                                                     gemm_h_lds_bytes<2, 2>());
        fprintf(stderr, "gemm_f16x3_kernel<2,2>: %d workgroups/CU (LDS %zu B)\n", n, gemm_h_lds_bytes<2, 2>());
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_f32_kernel<128, 4>, 256, 0);
        fprintf(stderr, "attn_f32_kernel<128,4>: %d workgroups/CU\n", n);
        return JMID_OK;
    }
    for (const Knob& kn : knobs)
        if (k == kn.name) {
            if (value < kn.lo || value > kn.hi || (k == "ln_rows" && value != 0 && value != 64 && value != 128) ||
                (k == "tail_rows" && value != 0 && value != 32 && value != 64))
                return fail(h, JMID_EINVAL, k + " out of range");
            h->tune.*(kn.field) = value;
            drop_graphs(h);          // captured loops hold the kernel variants the old knobs selected
            return JMID_OK;
        }
#endif
    return fail(h, JMID_EINVAL, "unknown tuning key " + k);
}

int64_t jmid_graph_replays(jmid_handle_t h) { return h ? h->graph_replays : -1; }

int64_t jmid_erange_count(jmid_handle_t h) { return h ? h->erange_calls : -1; }

int jmid_set_caller_stream(jmid_handle_t h, void* stream) {
    if (!h) return JMID_EINVAL;
    h->caller_stream = reinterpret_cast<hipStream_t>(stream);
    return JMID_OK;
}

int jmid_profile_enable(jmid_handle_t h, uint32_t class_mask) {
    if (!h) return JMID_EINVAL;
    h->prof_mask = class_mask;
    return JMID_OK;
}

static int prof_collect(jmid_ctx* h) {
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int c = 0; c < KC_COUNT; ++c) {
        for (auto& ev : h->prof_ev[c]) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
                h->prof_ms[c] += ms;
                h->prof_n[c] += 1;
            }
            h->ev_pool.push_back(ev);
        }
        h->prof_ev[c].clear();
    }
    return 0;
}

int jmid_profile_reset(jmid_handle_t h) {
    if (!h) return JMID_EINVAL;
    if (int rc = prof_collect(h)) return rc;
    for (int c = 0; c < KC_COUNT; ++c) {
        h->prof_ms[c] = 0;
        h->prof_n[c] = 0;
    }
    return JMID_OK;
}

int jmid_profile_get(jmid_handle_t h, int cls, int64_t* n_launches, double* total_ms) {
    if (!h || cls < 0 || cls >= KC_COUNT) return JMID_EINVAL;
    if (int rc = prof_collect(h)) return rc;
    if (n_launches) *n_launches = h->prof_n[cls];
    if (total_ms) *total_ms = h->prof_ms[cls];
    return JMID_OK;
}

int jmid_kernel_class_count(void) { return KC_COUNT; }
const char* jmid_kernel_class_name(int cls) { return (cls >= 0 && cls < KC_COUNT) ? kClassNames[cls] : ""; }

int jmid_synchronize(jmid_handle_t h) {
    if (!h) return JMID_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return JMID_OK;
}

#ifdef JMID_DIAGNOSTICS
// ---------------------------------------------------------------------------------------------- diagnostics
// Single-op entry points used by the unit tests (host buffers only).
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
        HIPCHK(h, hipMemset(h->range_flag, 0, sizeof(int)));
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
        HIPCHK(h, hipMemset(h->range_flag, 0, sizeof(int)));
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
        HIPCHK(h, hipMemset(h->range_flag, 0, sizeof(int)));
    }
    std::vector<void*> tmp;
    auto dalloc = [&](size_t bytes, const void* host) -> void* {
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        tmp.push_back(p);
        if (host) (void)hipMemcpy(p, host, bytes, hipMemcpyHostToDevice);
        else (void)hipMemset(p, 0, bytes);
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
    if (fused) {
        GemmLn2Args g2{ah, w16h, img.p, dB, dG, dT, xh, xl8, M, K, 1e-5f, h->range_flag, 0};
        hipError_t e = launch_gemm_ln2_mx(g2, h->stream);
        if (e != hipSuccess) rc = fail(h, JMID_EHIP, hipGetErrorString(e));
    } else {
        GemmHArgs g{};
        g.Ahi = ah; g.Alo = al; g.Whi = wh; g.Wlo = wl; g.W8 = img.p; g.bias = dB; g.C = dY; g.ldc = N; g.M = M; g.N = N; g.K = K;
        if (fused == 2) {        // the small-launch kernel with the LayerNorm tail (gemm_small.hpp, OUT_LN)
            unsigned* cnt = (unsigned*)dalloc(kLnCounters * sizeof(unsigned), nullptr);
            if (!cnt || !small_ln_fits(M, K)) return fail(h, JMID_EINVAL, "jmid_dbg_gemm_ln_mx: shape does not take the small fused kernel");
            g.ln_gamma = dG; g.ln_beta = dT; g.ln_xh = xh; g.ln_xl = nullptr; g.ln_xl8 = xl8; g.ln_cnt = cnt; g.ln_eps = 1e-5f; g.ln_no_lo = 0;
            rc = run_gemm_ln_small(h, KC_GEMM_OUT, g);
        } else {
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

#endif  // JMID_DIAGNOSTICS

}  // extern "C"
