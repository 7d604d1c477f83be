// Joint-KDE ranking of the K sampled futures of an episode and selection of the k most likely ones, batched over episodes.
//
// Reference: get_most_likely_samples (sicnav_diffusion/JMID/mid_sim_wrapper.py:14-169), which always takes its JOINT branch for
// this predictor (:20-21): per horizon step h a Gaussian KDE over the K samples in R^d, d = 2 A, bandwidth
// bw[h] = exp(linspace(ln .01, ln .1, H)) (:26-30);
//     cov  = centered^T centered / (K - 1)                      (:48-50)
//     P    = bw^-2 cov + 1e-6 I ;  L = cholesky(inverse(P))     (:61-66)
//     e_ij = -1/2 || (p_i - p_j) L^-1 / bw ||^2                 (:67-78)   (the quadratic form of (L^T L)^-1, as the reference has it)
//     ll_i = logsumexp_j(e_ij - Z),  Z = d/2 ln 2 pi + 1/2 * 2 sum ln L_ii + ln K ;  ll -= logsumexp_i(ll)     (:80-112)
//     total = sum_h ll ;  keep = argsort(total)[-k:] (ascending) ; logw = total[keep] - logsumexp(total[keep])   (:117-151)
// The reference runs this on the GPU when one exists (:26-30, 43-72) with torch's fp32 linalg.  Here the [d, d] algebra and the
// pairwise sums are fp64 (d <= 64, K <= 1024: the work is tiny and latency-bound, two launches per call), so the result is the
// exact value the reference's fp32 pipeline approximates: measured on the reference-generated fixtures the totals agree to
// 4e-6, far below the gaps of a decisive ranking.  Exact ties (samples isolated at every bandwidth: all likelihoods equal) are
// broken by sample index, ascending - torch.argsort's tie order is an artefact of its unstable sort and is not reproduced.
#pragma once
#include "common.hpp"

namespace jmid {

struct KdeArgs {
    const float* pos;     // [E, K, A, T, 2] integrated sample trajectories (jmid_denoise's pos_out layout)
    const float* bw;      // [T] bandwidth per horizon step (host-computed the reference's way), or null: computed here
    double* Y;            // [E * T][K][d] whitened points (only used when they do not fit in LDS)
    double* ll;           // [E * T][K] per-step log-likelihoods, normalised over the samples
    float* sel;           // [E, A, k, T, 2] kept samples, ascending likelihood
    float* logw;          // [E, A, k] their renormalised log-weights (the same row for every agent, :139-151)
    int E, A, K, T, k;
    int y_in_lds;
    int p_in_lds;         // the K x d points of the (episode, horizon step) staged in LDS as doubles (they are read ~2 d + 3 times each)
};

constexpr int KDE_THREADS = 256;
inline size_t kde_lds_bytes(int d, int K, bool y_in_lds, bool p_in_lds = false) {
    return sizeof(double) * (size_t(2) * d * d + d + KDE_THREADS + K + (y_in_lds ? size_t(K) * d : 0) + (p_in_lds ? size_t(K) * d : 0));
}

// in-place lower Cholesky factor of the SPD matrix M [d, d] (row-major; the strict upper triangle is left as it was)
__device__ __forceinline__ void kde_cholesky(double* M, int d, int tid) {
    for (int j = 0; j < d; ++j) {
        if (tid == 0) M[j * d + j] = sqrt(M[j * d + j]);
        __syncthreads();
        const double dj = M[j * d + j];
        for (int i = j + 1 + tid; i < d; i += KDE_THREADS) M[i * d + j] /= dj;
        __syncthreads();
        const int n = d - j - 1;                     // trailing update of the lower triangle, rows / columns j+1 ..
        for (int idx = tid; idx < n * n; idx += KDE_THREADS) {
            const int i = j + 1 + idx / n, c = j + 1 + idx % n;
            if (c <= i) M[i * d + c] -= M[i * d + j] * M[c * d + j];
        }
        __syncthreads();
    }
}
// X = G^-1 for lower-triangular G: column c by forward substitution, one thread per column (X lower, upper part zeroed)
__device__ __forceinline__ void kde_tri_inverse(const double* G, double* X, int d, int tid) {
    for (int c = tid; c < d; c += KDE_THREADS) {
        for (int i = 0; i < c; ++i) X[i * d + c] = 0.0;
        for (int i = c; i < d; ++i) {
            double s = i == c ? 1.0 : 0.0;
            for (int q = c; q < i; ++q) s -= G[i * d + q] * X[q * d + c];
            X[i * d + c] = s / G[i * d + i];
        }
    }
    __syncthreads();
}

// one workgroup per (episode, horizon step): ll[e, h, :]
static __global__ __launch_bounds__(KDE_THREADS) void kde_step_kernel(KdeArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char kde_lds_raw[];
    const int tid = threadIdx.x, blk = blockIdx.x;
    const int e = blk / g.T, h = blk - e * g.T;
    const int A = g.A, K = g.K, T = g.T, d = 2 * A;
    double* Pm = reinterpret_cast<double*>(kde_lds_raw);     // P, then inverse(P), then L
    double* Im = Pm + d * d;                                 // triangular inverses
    double* mean = Im + d * d;
    double* red = mean + d;
    double* llv = red + KDE_THREADS;
    double* Yp = g.y_in_lds ? llv + K : g.Y + (size_t)blk * K * d;
    double* Pp = g.p_in_lds ? llv + K + (g.y_in_lds ? K * d : 0) : nullptr;
    const float* pe = g.pos + (size_t)e * K * A * T * 2;
    // the points of this (episode, horizon step): staged once (one pass over global memory instead of a dependent load per use -
    // K = 100, d = 6: 91 -> ~20 us per launch at the reference's shipped operating point), or read in place when they do not fit
    if (Pp) {
        for (int idx = tid; idx < K * d; idx += KDE_THREADS) {
            const int s = idx / d, c = idx - s * d;
            Pp[idx] = (double)pe[(((size_t)s * A + (c >> 1)) * T + h) * 2 + (c & 1)];
        }
        __syncthreads();
    }
    auto pt = [&](int s, int c) { return Pp ? Pp[s * d + c] : (double)pe[(((size_t)s * A + (c >> 1)) * T + h) * 2 + (c & 1)]; };
    double bw;
    if (g.bw) bw = (double)g.bw[h];
    else bw = exp(log(0.01) + (T > 1 ? (double)h * (log(0.1) - log(0.01)) / (double)(T - 1) : 0.0));

    for (int c = tid; c < d; c += KDE_THREADS) {
        double s = 0.0;
        for (int q = 0; q < K; ++q) s += pt(q, c);
        mean[c] = s / (double)K;
    }
    __syncthreads();
    for (int idx = tid; idx < d * d; idx += KDE_THREADS) {
        const int i = idx / d, j = idx - i * d;
        double s = 0.0;
        const double mi = mean[i], mj = mean[j];
        for (int q = 0; q < K; ++q) s += (pt(q, i) - mi) * (pt(q, j) - mj);
        Pm[idx] = s / (double)(K - 1) / (bw * bw) + (i == j ? 1e-6 : 0.0);
    }
    __syncthreads();
    // inverse(P) through its Cholesky factor G: P^-1 = G^-T G^-1
    kde_cholesky(Pm, d, tid);
    kde_tri_inverse(Pm, Im, d, tid);
    for (int idx = tid; idx < d * d; idx += KDE_THREADS) {
        const int i = idx / d, j = idx - i * d;
        double s = 0.0;
        for (int q = (i > j ? i : j); q < d; ++q) s += Im[q * d + i] * Im[q * d + j];
        Pm[idx] = s;
    }
    __syncthreads();
    // L = cholesky(inverse(P)), L^-1, log det
    kde_cholesky(Pm, d, tid);
    kde_tri_inverse(Pm, Im, d, tid);
    double log_det = 0.0;
    for (int i = 0; i < d; ++i) log_det += log(Pm[i * d + i]);
    log_det *= 2.0;
    // whitened points  y_s = p_s L^-1 / bw
    for (int idx = tid; idx < K * d; idx += KDE_THREADS) {
        const int s = idx / d, j = idx - s * d;
        double y = 0.0;
        for (int i = j; i < d; ++i) y += pt(s, i) * Im[i * d + j];
        Yp[idx] = y / bw;
    }
    __syncthreads();
    const double Z = 0.5 * (double)d * log(2.0 * 3.14159265358979323846) + 0.5 * log_det + log((double)K);
    // pairwise sums: the K x K exponentials are the kernel's time (fp64 exp), so every sample gets TPS = 2^n <= 256 / K threads,
    // each summing a contiguous slice of j; the slices are added in j order (red[] is free until the normalisation below)
    int tps = 1;
    while (tps * 2 * K <= KDE_THREADS) tps *= 2;
    {
        const int i = tid / tps, sl = tid - i * tps;
        double acc = 0.0;                              // max_j e_ij = e_ii = 0: no shift needed
        if (tps > 1 && i < K) {
            const int j0 = (int)((long)sl * K / tps), j1 = (int)((long)(sl + 1) * K / tps);
            for (int j = j0; j < j1; ++j) {
                double q = 0.0;
                for (int c = 0; c < d; ++c) {
                    const double t = Yp[i * d + c] - Yp[j * d + c];
                    q += t * t;
                }
                acc += exp(-0.5 * q);
            }
        }
        if (tps > 1) {
            red[tid] = acc;
            __syncthreads();
            if (i < K && sl == 0) {
                double a = red[tid];
                for (int u = 1; u < tps; ++u) a += red[tid + u];
                llv[i] = log(a) - Z;
            }
        } else {
            for (int ii = tid; ii < K; ii += KDE_THREADS) {      // K > 128: one thread per sample, several samples per thread
                double a = 0.0;
                for (int j = 0; j < K; ++j) {
                    double q = 0.0;
                    for (int c = 0; c < d; ++c) {
                        const double t = Yp[ii * d + c] - Yp[j * d + c];
                        q += t * t;
                    }
                    a += exp(-0.5 * q);
                }
                llv[ii] = log(a) - Z;
            }
        }
    }
    __syncthreads();
    // normalise over the samples: ll -= logsumexp(ll)
    double m = -INFINITY;
    for (int i = tid; i < K; i += KDE_THREADS) m = fmax(m, llv[i]);
    red[tid] = m;
    __syncthreads();
    for (int o = KDE_THREADS / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] = fmax(red[tid], red[tid + o]);
        __syncthreads();
    }
    m = red[0];
    __syncthreads();
    double s = 0.0;
    for (int i = tid; i < K; i += KDE_THREADS) s += exp(llv[i] - m);
    red[tid] = s;
    __syncthreads();
    for (int o = KDE_THREADS / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const double lse = m + log(red[0]);
    for (int i = tid; i < K; i += KDE_THREADS) g.ll[(size_t)blk * K + i] = llv[i] - lse;
}

// one workgroup per episode: totals over the horizon, stable ascending rank, the last k, their log-weights, the gather
static __global__ __launch_bounds__(KDE_THREADS) void kde_select_kernel(KdeArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char kde_lds_raw[];
    const int tid = threadIdx.x, e = blockIdx.x;
    const int A = g.A, K = g.K, T = g.T, k = g.k;
    double* tot = reinterpret_cast<double*>(kde_lds_raw);     // [K]
    double* red = tot + K;                                    // [KDE_THREADS]
    int* keep = reinterpret_cast<int*>(red + KDE_THREADS);    // [k]
    for (int i = tid; i < K; i += KDE_THREADS) {
        double s = 0.0;
        for (int h = 0; h < T; ++h) s += g.ll[((size_t)e * T + h) * K + i];
        // a NaN total (non-finite positions, a covariance whose Cholesky factor fails) ranks lowest: the order below stays
        // total, every rank is taken exactly once and keep[] is fully written (the reference returns NaNs in that case)
        tot[i] = s == s ? s : -INFINITY;
    }
    __syncthreads();
    for (int i = tid; i < K; i += KDE_THREADS) {
        const double ti = tot[i];
        int rank = 0;
        for (int j = 0; j < K; ++j) {
            const double tj = tot[j];
            rank += (tj < ti || (tj == ti && j < i)) ? 1 : 0;
        }
        if (rank >= K - k) keep[rank - (K - k)] = i;
    }
    __syncthreads();
    const double m = tot[keep[k - 1]];                        // the largest kept total
    double s = 0.0;
    for (int q = tid; q < k; q += KDE_THREADS) s += exp(tot[keep[q]] - m);
    red[tid] = s;
    __syncthreads();
    for (int o = KDE_THREADS / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const double lse = m + log(red[0]);
    for (int idx = tid; idx < A * k; idx += KDE_THREADS) g.logw[(size_t)e * A * k + idx] = (float)(tot[keep[idx % k]] - lse);
    const float* pe = g.pos + (size_t)e * K * A * T * 2;
    float* se = g.sel + (size_t)e * A * k * T * 2;
    for (int idx = tid; idx < A * k * T * 2; idx += KDE_THREADS) {
        const int r = idx % (T * 2), q = (idx / (T * 2)) % k, a = idx / (T * 2 * k);
        se[idx] = pe[((size_t)keep[q] * A + a) * T * 2 + r];
    }
}

// do the whitened points of one (episode, horizon step) fit in LDS next to the [d, d] algebra?  (otherwise KdeArgs::Y)
inline bool kde_y_in_lds(int A, int K) { return kde_lds_bytes(2 * A, K, true) <= 96 * 1024; }

inline hipError_t launch_kde(const KdeArgs& g0, hipStream_t st) {
    KdeArgs g = g0;
    const int d = 2 * g.A;
    g.y_in_lds = kde_y_in_lds(g.A, g.K);
    g.p_in_lds = kde_lds_bytes(d, g.K, g.y_in_lds != 0, true) <= 150 * 1024;
    const size_t lds = kde_lds_bytes(d, g.K, g.y_in_lds != 0, g.p_in_lds != 0);
    static DevSeen seen;
    if (auto once_ = first_use_on_device(seen))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kde_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kde_step_kernel, dim3(g.E * g.T), dim3(KDE_THREADS), lds, st, g);
    const size_t lds2 = sizeof(double) * (g.K + KDE_THREADS) + sizeof(int) * g.k;
    hipLaunchKernelGGL(kde_select_kernel, dim3(g.E), dim3(KDE_THREADS), lds2, st, g);
    return hipGetLastError();
}

}  // namespace jmid
