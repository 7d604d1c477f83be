// libjmid_hip.so -- per-kernel-class HIP-event profiling (jmid_profile_*).
#include "jmid_ctx.hpp"

const char* const kClassNames[KC_COUNT] = {"gemm_qkv", "gemm_attn_out", "gemm_ff1", "gemm_ff2", "gemm_tail", "attention",
                                     "add_layernorm", "embed", "out_ddim", "hyper", "encoder", "integrate",
                                     "episode_metrics", "v_transpose", "kde_topk"};


namespace jmid_host {

int prof_collect(jmid_ctx* h) {
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

}  // namespace jmid_host

extern "C" {

int jmid_profile_enable(jmid_handle_t h, uint32_t class_mask) {
    if (!h) return JMID_EINVAL;
    h->prof_mask = class_mask;
    return JMID_OK;
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

}  // extern "C"

