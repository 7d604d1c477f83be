// Launch helpers shared by the planner and the diagnostics entry points: one GEMM / LayerNorm launch on the handle's stream,
// bracketed by the profiling events of its kernel class.
#pragma once
#include "jmid_ctx.hpp"

namespace jmid_host {

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

// out_proj / linear2 + residual + LayerNorm as ONE small launch, the row statistics exchanged between the workgroups of a row tile (gemm_small.hpp, OUT_LNX; F16MX at d_model 512)
inline int run_gemm_lnx_small(jmid_ctx* h, int cls, GemmHArgs& g) {
    g.range_flag = h->range_flag;
    g.x2 = h->x2;
    if (++h->lnx_epoch == 0) h->lnx_epoch = 1;        // (0 is what the zeroed granules hold)
    g.ln_epoch = h->lnx_epoch;
    g.ln_polls = tune().lnx_polls;
    g.ln_withhold = tune().lnx_withhold;
    // the kernel is written for the F16MX operand set (byte lo plane of the residual stream, bf8 image of W_lo) only
    if (!(g.x2 && g.W8 && g.ln_xl8)) return fail(h, JMID_EINVAL, "one-launch GEMM + LayerNorm without the F16MX operand set");
    ProfScope ps(h, cls);
    HIPCHK(h, (launch_gemm_small<EPI_BIAS, OUT_LNX>(g, small_lnx_fits(g.M, g.K), h->stream)));      // (2: one workgroup per CU, 9: two)
    return 0;
}


// JMID_PREC_F16MX: hand the GEMM the fp8 image of this weight's lo plane (the kernels that have no fp8 path ignore it)
inline void set_w8(jmid_ctx* h, GemmHArgs& g, const std::string& name) {
    g.W8 = nullptr;
    if (!h->mx) return;
    auto it = h->w8.find(name);
    if (it == h->w8.end()) return;
    g.W8 = it->second.p;
}

inline int run_add_ln(jmid_ctx* h, float* X, const float* Y, const float* gm, const float* bt, int M, int d,
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

}  // namespace jmid_host
