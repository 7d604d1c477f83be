// libjmid_hip.so -- what every translation unit of the host side shares: the handle, error / profiling helpers and the
// internal entry points between the units.  gfx950 only.  No CPU fallback: every entry point that computes needs a HIP device.
//   jmid_abi.hip      the C ABI proper (include/jmid_hip.h): handle lifetime, encode / denoise / topk / predict, knobs, stream
//   jmid_weights.hip  weight registry, operand planes (fp16 hi / lo, bf8 images, k16 panels), sampler step tables
//   jmid_planner.hip  chunk plan, step workspace, one net evaluation (net_step), the denoise loop (run_network)
//   jmid_profile.hip  per-kernel-class HIP-event profiling
//   jmid_diag.hip     jmid_dbg_* single-kernel entry points (-DJMID_DIAGNOSTICS only)
#pragma once
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

using namespace jmid;

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
extern const char* const kClassNames[KC_COUNT];

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
    unsigned lnx_epoch = 0;     // launch tag of the small-launch GEMM + LayerNorm with the statistics exchange (gemm_small.hpp, OUT_LNX)
    bool lnx_off = false;       // a workgroup of that kernel once gave up waiting for a partner (range flag bit 1): the handle stays on GEMM + add_ln2
    int64_t lnx_timeouts = 0;   // calls on this handle that ended with JMID_ETIMEOUT for that reason (jmid_timeout_count)
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

constexpr size_t kLnxWords = 2 * jmid::SM_LNX_GRANULES;      // 32-bit words of the exchange granules (8 bytes each) of the small-launch GEMM + LayerNorm per step workspace (gemm_small.hpp, OUT_LNX)

namespace jmid_host {

std::string& thread_error();      // the last error of this thread (jmid_last_error(NULL))
int fail(jmid_ctx* h, int code, const std::string& msg);

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

inline size_t numel(const std::vector<size_t>& s) {
    size_t n = 1;
    for (size_t v : s) n *= v;
    return n;
}

inline const float* W(jmid_ctx* h, const std::string& name) { return h->w[name].p; }

// jmid_weights.hip
void register_shapes(jmid_ctx* h);
int dev_alloc_copy(jmid_ctx* h, float** out, const std::vector<float>& host);
int fetch_host(jmid_ctx* h, const std::string& name, std::vector<float>& out);
int upload_time_table(jmid_ctx* h);
int make_w8(jmid_ctx* h, const float* dW, int N, int K, jmid_ctx::W8Image* out);
// jmid_planner.hip
void drop_graphs(jmid_ctx* h);
void sync_lanes(jmid_ctx* h);
int ensure_arena(jmid_ctx* h, size_t bytes);
std::vector<int> plan_chunks(const jmid_ctx* h, int E, int tokens_per_episode);
int check_ready(jmid_ctx* h);
int order_in(jmid_ctx* h, int mem);
int order_out(jmid_ctx* h, int mem);
int run_network(jmid_ctx* h, int E, int A, int K, int T, const float* x_in, const float* ctx, const float* p0, float dt,
                int precision, int single_step, float* vel_out, float* pos_out, float* e_out, int mem,
                const float* z_in = nullptr);
int flagged_call(jmid_ctx* h, int flag);      // the status of a call whose range flag came back set (JMID_ETIMEOUT / JMID_ERANGE)
int launch_episode_metrics(jmid_ctx* h, const float* pos, const float* gt, float* out, int E, int K, int A, int T);
// jmid_profile.hip
int prof_collect(jmid_ctx* h);

}  // namespace jmid_host
using namespace jmid_host;
