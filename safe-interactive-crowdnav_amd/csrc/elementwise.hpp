// Bandwidth-bound pieces of one denoise step (all fp32):
//   embed        concat1 (ConcatSquashLinear 2 -> d) + positional encoding    diffusion.py:183-185, common.py:37-72
//   add_ln       residual + post-LayerNorm of nn.TransformerEncoderLayer       (norm_first=False, eps 1e-5)
//   out_ddim     final ConcatSquashLinear (d_low -> 2) + DDIM update           diffusion.py:209, 524-528
//   integrate    SingleIntegrator.integrate_samples                            single_integrator.py:290-321
#pragma once
#include "common.hpp"
#include "gemm_f16x3.hpp"

namespace jmid {

// ------------------------------------------------------------------------------------------------ embed
struct EmbedArgs {
    const float* x;      // [M, 2]
    const float* W1;     // [d, 2]
    const float* b1;     // [d]
    const float* pe;     // [max_len, d]
    const float* hyp;    // [EA, hyp_ld]
    const float* thyp;   // [hyp_ld]
    float* X;            // [M, d] fp32, or nullptr when only the planes are wanted (split-fp16 mode)
    int M, d, hyp_ld, goff, boff;
    RowMap rmap;
    half_t* Xh;          // optional hi/lo planes of X (blocked panel layout) for the split-fp16 GEMMs
    half_t* Xl;
    unsigned char* Xl8;  // JMID_PREC_F16MX at d_model 512: the lo plane as bf8 bytes (gemm_ln2_mx.hpp::blk8_index) instead of Xl
};

// channels j..j+3 of token m (row t of the positional table): ConcatSquash(2 -> d) + PE, stored as fp32 and / or planes.
// In two pieces: what depends only on the (episode, agent) row and the step (weights, gate, bias: EmbedCols), and the rest.
struct EmbedCols {
    f32x4 w0, w1, b1, gate, bias;
};
__device__ __forceinline__ void embed_cols(const EmbedArgs& a, int j, const float* hrow, EmbedCols& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int cc = j + e;
        c.w0[e] = a.W1[2 * cc];
        c.w1[e] = a.W1[2 * cc + 1];
        c.b1[e] = a.b1[cc];
        c.gate[e] = sigmoidf_(hrow[a.goff + cc] + a.thyp[a.goff + cc]);
        c.bias[e] = hrow[a.boff + cc] + a.thyp[a.boff + cc];
    }
}
__device__ __forceinline__ void embed_store_cols(const EmbedArgs& a, int m, int j, int t, float x0, float x1, const EmbedCols& c) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lin = c.w0[e] * x0 + c.w1[e] * x1 + c.b1[e];
        o[e] = lin * c.gate[e] + c.bias[e] + a.pe[(size_t)t * a.d + j + e];
    }
    if (a.X) *reinterpret_cast<f32x4*>(a.X + (size_t)m * a.d + j) = o;
    if (a.Xh) {
        f16x4 vh, vl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            half_t hh, ll;
            split_f32(o[e], hh, ll);
            vh[e] = hh;
            vl[e] = ll;
        }
        const size_t ob = blk_index(m, j, a.d);
        *reinterpret_cast<f16x4*>(a.Xh + ob) = vh;
        if (a.Xl8) {
            const i32x2_e dl = __builtin_bit_cast(i32x2_e, vl);
            *reinterpret_cast<int*>(a.Xl8 + ((((size_t)(m >> 7) * (a.d >> 5) + (j >> 5)) * 128 + (m & 127)) * 32 + (j & 31))) =
                bf8_of_f16x4(dl[0], dl[1]);
        } else {
            *reinterpret_cast<f16x4*>(a.Xl + ob) = vl;
        }
    }
}
__device__ __forceinline__ void embed_store(const EmbedArgs& a, int m, int j, int t, float x0, float x1, const float* hrow) {
    EmbedCols c;
    embed_cols(a, j, hrow, c);
    embed_store_cols(a, m, j, t, x0, x1, c);
}

// one thread per (token, 4 channels)
static __global__ __launch_bounds__(256) void embed_kernel(EmbedArgs a) {
    args_now(a);
    const int d4 = a.d >> 2;
    const long total = (long)a.M * d4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / d4), j = (int)(idx % d4) * 4;
        embed_store(a, m, j, a.rmap.t_of(m), a.x[2 * (size_t)m], a.x[2 * (size_t)m + 1],
                    a.hyp + (size_t)a.rmap.ea(m) * a.hyp_ld);
    }
}

// ------------------------------------------------------------------------------------------------ add + LayerNorm
// X[m,:] = LN(X[m,:] + Y[m,:]) * gamma + beta ; one wave per row, d <= 64*4*VPL
// PLANES: the residual stream X lives ONLY in its hi/lo planes (blocked panel layout, hi + lo carries x to
// ~1 fp32 ulp): they are read for the residual and rewritten, the fp32 X array is not touched - one fp32 stream less
// through HBM per LayerNorm.
template <int VPL, bool PLANES>  // float4 vectors per lane
__global__ __launch_bounds__(256) void add_ln_kernel(float* X, const float* Y, const float* gamma, const float* beta,
                                                     int M, int d, float eps, half_t* Xh, half_t* Xl) {
    args_now_each(X, Y, gamma, beta, M, d, eps, Xh, Xl);
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    float* xr = PLANES ? nullptr : X + (size_t)row * d;
    const float* yr = Y + (size_t)row * d;
    f32x4 v[VPL];
    float s = 0.f;
    // gamma / beta requested with the row (behind the statistics they are VPL more dependent round trips: the compiler cannot
    // move them above the stores of the vector before)
    f32x4 gmv[VPL], btv[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
            gmv[i] = *reinterpret_cast<const f32x4*>(gamma + c);
            btv[i] = *reinterpret_cast<const f32x4*>(beta + c);
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
            f32x4 a;
            if (PLANES) {
                const size_t ob = blk_index(row, c, d);
                const f16x4 ph = *reinterpret_cast<const f16x4*>(Xh + ob);
                const f16x4 pl = *reinterpret_cast<const f16x4*>(Xl + ob);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = (float)ph[e] + (float)pl[e];
            } else {
                a = *reinterpret_cast<const f32x4*>(xr + c);
            }
            f32x4 b = *reinterpret_cast<const f32x4*>(yr + c);
            v[i] = a + b;
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        } else {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = v[i][e] - mean;
                q += t * t;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
            const f32x4 gm = gmv[i], bt = btv[i];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * gm[e] + bt[e];
            if (!PLANES) *reinterpret_cast<f32x4*>(xr + c) = o;
            if (Xh) {
                f16x4 vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    half_t hh, ll;
                    split_f32(o[e], hh, ll);
                    vh[e] = hh;
                    vl[e] = ll;
                }
                const size_t ob = blk_index(row, c, d);
                *reinterpret_cast<f16x4*>(Xh + ob) = vh;
                *reinterpret_cast<f16x4*>(Xl + ob) = vl;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ final CSL + DDIM
struct OutArgs {
    const float* Y4;     // [M, dl]
    const float* Wo;     // [2, dl]
    const float* bo;     // [2]
    const float* hyp;    // [EA, hyp_ld]
    const float* thyp;   // [hyp_ld]
    float* x;            // [M, 2] in/out (DDIM) ; untouched when e_out != nullptr
    float* e_out;        // [M, 2] or nullptr: write e_theta instead of updating x
    int M, dl, hyp_ld, goff, boff;
    float c_e, c_x, n_x, n_e;
    RowMap rmap;
    // DDPM (diffusion.py:521-522): x <- c0*(x - c1*e) + sigma*z ; z == nullptr -> DDIM update above
    const float* z;      // [M, 2] normal draws of this step (nullptr for t == 1: zeros)
    int ddpm;
    float c0, c1, sigma;
};

// The three pieces of the output stage of one token, shared by out_ddim_kernel (one wave per token, lane 0 updates) and
// tail_f16x3_kernel (a wave works on several tokens at once, lane q updates token q): the same expressions, so the same bits.
//   out_dot     partial dot products of the gated Y4 row `y` with the two rows of the output layer + wave reduction
//   out_update  (one lane) final ConcatSquash gate / bias, then e_theta out or the DDIM / DDPM update of x in place
//   embed_row   (whole wave) the next step's embedding of the token: embed_kernel's arithmetic with `nxt.thyp`
__device__ __forceinline__ void out_dot(const OutArgs& a, const float* y, int lane, float& s0, float& s1) {
    s0 = 0.f;
    s1 = 0.f;
    for (int c = lane; c < a.dl; c += 64) {
        const float v = y[c];
        s0 += v * a.Wo[c];
        s1 += v * a.Wo[a.dl + c];
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
}
__device__ __forceinline__ void out_update(const OutArgs& a, int m, float s0, float s1, float& xn0, float& xn1) {
    const float* hrow = a.hyp + (size_t)a.rmap.ea(m) * a.hyp_ld;
    const float e0 = (s0 + a.bo[0]) * sigmoidf_(hrow[a.goff] + a.thyp[a.goff]) + hrow[a.boff] + a.thyp[a.boff];
    const float e1 = (s1 + a.bo[1]) * sigmoidf_(hrow[a.goff + 1] + a.thyp[a.goff + 1]) + hrow[a.boff + 1] +
                     a.thyp[a.boff + 1];
    if (a.e_out) {
        a.e_out[2 * (size_t)m] = e0;
        a.e_out[2 * (size_t)m + 1] = e1;
    } else {
        // x0 = (x - e*sqrt(1-abar_t))/sqrt(abar_t) ; x <- sqrt(abar_next)*x0 + sqrt(1-abar_next)*e
        const float x0 = a.x[2 * (size_t)m], x1 = a.x[2 * (size_t)m + 1];
        if (a.ddpm) {
            const float z0 = a.z ? a.z[2 * (size_t)m] : 0.f, z1 = a.z ? a.z[2 * (size_t)m + 1] : 0.f;
            a.x[2 * (size_t)m] = a.c0 * (x0 - a.c1 * e0) + a.sigma * z0;
            a.x[2 * (size_t)m + 1] = a.c0 * (x1 - a.c1 * e1) + a.sigma * z1;
        } else {
            const float p0 = (x0 - e0 * a.c_e) / a.c_x, p1 = (x1 - e1 * a.c_e) / a.c_x;
            a.x[2 * (size_t)m] = a.n_x * p0 + a.n_e * e0;
            a.x[2 * (size_t)m + 1] = a.n_x * p1 + a.n_e * e1;
        }
        xn0 = a.x[2 * (size_t)m];
        xn1 = a.x[2 * (size_t)m + 1];
    }
}
__device__ __forceinline__ void embed_row(const EmbedArgs& nxt, int m, int lane, float x0, float x1) {
    const int t = nxt.rmap.t_of(m);
    const float* hrow = nxt.hyp + (size_t)nxt.rmap.ea(m) * nxt.hyp_ld;
    for (int j = lane * 4; j < nxt.d; j += 256) embed_store(nxt, m, j, t, x0, x1, hrow);
}

template <bool EMBED_NEXT>
__device__ __forceinline__ void out_ddim_row(const OutArgs& a, const EmbedArgs& nxt, int m, int lane, const float* y) {
    float s0, s1;
    out_dot(a, y, lane, s0, s1);
    float xn0 = 0.f, xn1 = 0.f;      // the updated x of this token (lane 0)
    if (lane == 0) out_update(a, m, s0, s1, xn0, xn1);
    if (EMBED_NEXT) embed_row(nxt, m, lane, __shfl(xn0, 0, 64), __shfl(xn1, 0, 64));
}

// one wave per token.  EMBED_NEXT: one launch and one pass over x less per denoise step.
template <bool EMBED_NEXT>
__global__ __launch_bounds__(256) void out_ddim_kernel(OutArgs a, EmbedArgs nxt) {
    args_now_each(a, nxt);
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= a.M) return;
    out_ddim_row<EMBED_NEXT>(a, nxt, m, lane, a.Y4 + (size_t)m * a.dl);
}

// One wave per TRAJECTORY (the T consecutive tokens of one sample of one agent), or per piece of one, for d <= 512: the ConcatSquash gate and bias of
// the next embedding depend on the (episode, agent) row and the step only, so they - 512 sigmoids and 2 KB of hyper-net rows
// per token in the kernel above - are computed once per trajectory, and a wave's stores walk through adjacent rows of the
// blocked planes.  Same expressions per element (embed_cols / embed_store_cols): the same bits.  70 -> 4x us per 61 200-token
// launch (the kernel above was 2.3 % of an F16MX step).
// The body: tokens m0 ... m0 + tpw - 1 (inside one trajectory) by one wave; `y` = the Y4 row of token m0, rows y_ld floats apart
// (YP: a global pointer into a.Y4, or an LDS pointer into the tile a GEMM epilogue left there: gemm_small_out_kernel).
template <bool EMBED_NEXT, typename YP>
__device__ __forceinline__ void out_ddim_piece(const OutArgs& a, const EmbedArgs& nxt, int m0, int tpw, int lane, YP y, int y_ld) {
    const int t0 = a.rmap.t_of(m0);
    EmbedCols c[2];
    const int j0 = lane * 4, j1 = lane * 4 + 256;
    if (EMBED_NEXT) {
        const float* hrow = nxt.hyp + (size_t)nxt.rmap.ea(m0) * nxt.hyp_ld;
        if (j0 < nxt.d) embed_cols(nxt, j0, hrow, c[0]);
        if (j1 < nxt.d) embed_cols(nxt, j1, hrow, c[1]);
    }
    // Nothing in the per-token chain waits for memory: the Y4 rows of up to 12 tokens (two values per lane and token for
    // dl <= 128) and their x are requested at once, and the output layer's gate / bias (per (episode, agent) row and step, like
    // the embedding's) are computed once per wave.  Every lane carries the token's update (the reductions are wave-uniform);
    // lane 0 stores it.  out_update's expressions, term by term.
    const bool regs = a.dl <= 128;
    const int c0 = lane, c1 = lane + 64;
    const float wo00 = c0 < a.dl ? a.Wo[c0] : 0.f, wo10 = c0 < a.dl ? a.Wo[a.dl + c0] : 0.f;
    const float wo01 = c1 < a.dl ? a.Wo[c1] : 0.f, wo11 = c1 < a.dl ? a.Wo[a.dl + c1] : 0.f;
    const float* hrow_o = a.hyp + (size_t)a.rmap.ea(m0) * a.hyp_ld;
    const float g0 = sigmoidf_(hrow_o[a.goff] + a.thyp[a.goff]), g1 = sigmoidf_(hrow_o[a.goff + 1] + a.thyp[a.goff + 1]);
    const float hb0 = hrow_o[a.boff], hb1 = hrow_o[a.boff + 1], tb0 = a.thyp[a.boff], tb1 = a.thyp[a.boff + 1];
    const float bo0 = a.bo[0], bo1 = a.bo[1];
    for (int tb = 0; tb < tpw; tb += 12) {
        const int nb = tpw - tb < 12 ? tpw - tb : 12;
        float yv[12][2];
        if (regs) {
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const YP yu = y + (size_t)(tb + (u < nb ? u : 0)) * y_ld;
                yv[u][0] = c0 < a.dl ? yu[c0] : 0.f;
                yv[u][1] = c1 < a.dl ? yu[c1] : 0.f;
            }
        }
        // lane u holds x (and the DDPM draw) of token u of the block
        const int mu = m0 + tb + (lane < nb ? lane : 0);
        const float xl0 = a.e_out ? 0.f : a.x[2 * (size_t)mu], xl1 = a.e_out ? 0.f : a.x[2 * (size_t)mu + 1];
        const float zl0 = (a.ddpm && a.z) ? a.z[2 * (size_t)mu] : 0.f, zl1 = (a.ddpm && a.z) ? a.z[2 * (size_t)mu + 1] : 0.f;
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            if (u >= nb) break;
            const int m = m0 + tb + u, t = t0 + tb + u;
            float s0, s1;
            if (regs) {               // out_dot's sums in out_dot's order: c = lane, then lane + 64
                s0 = 0.f;
                s1 = 0.f;
                if (c0 < a.dl) { s0 += yv[u][0] * wo00; s1 += yv[u][0] * wo10; }
                if (c1 < a.dl) { s0 += yv[u][1] * wo01; s1 += yv[u][1] * wo11; }
                s0 = wave_sum(s0);
                s1 = wave_sum(s1);
            } else {
                out_dot(a, a.Y4 + (size_t)m * a.dl, lane, s0, s1);
            }
            const float e0 = (s0 + bo0) * g0 + hb0 + tb0;
            const float e1 = (s1 + bo1) * g1 + hb1 + tb1;
            float xn0 = 0.f, xn1 = 0.f;
            if (a.e_out) {
                if (lane == 0) {
                    a.e_out[2 * (size_t)m] = e0;
                    a.e_out[2 * (size_t)m + 1] = e1;
                }
            } else {
                const float x0 = __shfl(xl0, u, 64), x1 = __shfl(xl1, u, 64);
                if (a.ddpm) {
                    const float z0 = __shfl(zl0, u, 64), z1 = __shfl(zl1, u, 64);
                    xn0 = a.c0 * (x0 - a.c1 * e0) + a.sigma * z0;
                    xn1 = a.c0 * (x1 - a.c1 * e1) + a.sigma * z1;
                } else {
                    const float p0 = (x0 - e0 * a.c_e) / a.c_x, p1 = (x1 - e1 * a.c_e) / a.c_x;
                    xn0 = a.n_x * p0 + a.n_e * e0;
                    xn1 = a.n_x * p1 + a.n_e * e1;
                }
                if (lane == 0) {
                    a.x[2 * (size_t)m] = xn0;
                    a.x[2 * (size_t)m + 1] = xn1;
                }
            }
            if (EMBED_NEXT) {
                if (j0 < nxt.d) embed_store_cols(nxt, m, j0, t, xn0, xn1, c[0]);
                if (j1 < nxt.d) embed_store_cols(nxt, m, j1, t, xn0, xn1, c[1]);
            }
        }
    }
}

template <bool EMBED_NEXT>
__global__ __launch_bounds__(256) void out_ddim_traj_kernel(OutArgs a, EmbedArgs nxt, int tpw) {
    // tpw tokens per wave (a divisor of T: a whole trajectory, or a piece of one when there are few trajectories)
    const int lane = threadIdx.x & 63;
    const int piece = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int m0 = piece * tpw;
    if (m0 >= a.M) return;
    out_ddim_piece<EMBED_NEXT>(a, nxt, m0, tpw, lane, a.Y4 + (size_t)m0 * a.dl, a.dl);
}

// ------------------------------------------------------------------------------------------------ integrator
// vel [R, T, 2] (R = E*K*A rows, r = (e*K+s)*A + a) -> pos = cumsum_t(vel)*dt + p0[e, a]
static __global__ void integrate_kernel(const float* vel, const float* p0, float* pos, int R, int T, int A, int KA, float dt) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (row, c)
    if (idx >= R * 2) return;
    const int r = idx >> 1, c = idx & 1;
    const int e = r / KA, ag = r % A;
    const float base = p0[((size_t)e * A + ag) * 2 + c];
    float acc = 0.f;
    for (int t = 0; t < T; ++t) {
        acc += vel[((size_t)r * T + t) * 2 + c];
        pos[((size_t)r * T + t) * 2 + c] = acc * dt + base;
    }
}

// ------------------------------------------------------------------------------------------------ sweep metrics
// Per-episode displacement metrics of the sampled futures against a ground-truth future (the per-element ADE/FDE
// definition of MID/evaluation/evaluation.py:11-28, with the joint (scene-level) minimum over samples):
//   out[e] = { mean ADE over (sample, agent, t),  min over samples of the agent-mean ADE,
//              mean FDE over (sample, agent),     min over samples of the agent-mean FDE }
// pos [E, K, A, T, 2], gt [E, A, T, 2].  One workgroup per episode, one wave per sample (strided).
static __global__ __launch_bounds__(256) void episode_metrics_kernel(const float* pos, const float* gt, float* out, int K, int A,
                                                              int T) {
    __shared__ float s_ade[256], s_fde[256];
    const int e = blockIdx.x, tid = threadIdx.x;
    float sum_ade = 0.f, sum_fde = 0.f, min_ade = INFINITY, min_fde = INFINITY;
    for (int s = tid; s < K; s += blockDim.x) {
        float ade = 0.f, fde = 0.f;
        for (int a = 0; a < A; ++a) {
            const float* p = pos + (((size_t)e * K + s) * A + a) * T * 2;
            const float* g = gt + ((size_t)e * A + a) * T * 2;
            float acc = 0.f, last = 0.f;
            for (int t = 0; t < T; ++t) {
                const float dx = p[2 * t] - g[2 * t], dy = p[2 * t + 1] - g[2 * t + 1];
                last = sqrtf(dx * dx + dy * dy);
                acc += last;
            }
            ade += acc / (float)T;
            fde += last;
        }
        ade /= (float)A;
        fde /= (float)A;
        sum_ade += ade;
        sum_fde += fde;
        min_ade = fminf(min_ade, ade);
        min_fde = fminf(min_fde, fde);
    }
    // block reductions (sum, min) through LDS
    s_ade[tid] = sum_ade; s_fde[tid] = sum_fde;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { s_ade[tid] += s_ade[tid + o]; s_fde[tid] += s_fde[tid + o]; }
        __syncthreads();
    }
    const float tot_ade = s_ade[0], tot_fde = s_fde[0];
    __syncthreads();
    s_ade[tid] = min_ade; s_fde[tid] = min_fde;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { s_ade[tid] = fminf(s_ade[tid], s_ade[tid + o]); s_fde[tid] = fminf(s_fde[tid], s_fde[tid + o]); }
        __syncthreads();
    }
    if (tid == 0) {
        out[4 * e + 0] = tot_ade / (float)K;
        out[4 * e + 1] = s_ade[0];
        out[4 * e + 2] = tot_fde / (float)K;
        out[4 * e + 3] = s_fde[0];
    }
}

}  // namespace jmid
