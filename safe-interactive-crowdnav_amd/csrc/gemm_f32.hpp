// Exact-fp32 "NT" GEMM on the CDNA4 matrix cores:  C[M,N] = A[M,K] . W[N,K]^T  (+ epilogue)
//
// v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, bit-identical to an fmaf chain).  Both operands are
// K-contiguous (activations [tokens, features], nn.Linear weights [out, in]), so one ds_read_b128 of four
// consecutive k feeds four MFMAs: lanes 0-31 carry k = kb+e, lanes 32-63 carry k = kb+4+e (e = 0..3).
//
// Block = 4 waves (2x2); each wave owns (WM*32) x (WN*32) of C.  BK = 32 floats, double-buffered LDS,
// rows padded to 36 floats so that a 16-lane ds_read_b128 group touches 16 distinct 16-B slots.
// The next K-tile is fetched into registers before the MFMAs of the current one and written to the other
// LDS buffer afterwards (one barrier per K-tile).
#pragma once
#include "common.hpp"

namespace jmid {

enum GemmEpi { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_CSL = 2 };

struct GemmArgs {
    const float* A;      // [M, K], row stride lda
    const float* W;      // [N, K], row stride ldw
    const float* bias;   // [N] or nullptr
    float* C;            // [M, N], row stride ldc
    int M, N, K, lda, ldw, ldc;
    // EPI_CSL: C = (acc + bias) * sigmoid(hyp[ea(m), goff+n] + thyp[goff+n]) + hyp[ea(m), boff+n] + thyp[boff+n]
    const float* hyp;    // [EA, hyp_ld]
    const float* thyp;   // [hyp_ld] (row of the current step)
    int hyp_ld, goff, boff;
    RowMap rmap;
};

constexpr int GEMM_BK = 32;
constexpr int GEMM_LDS_LD = 36;

template <int WM, int WN>
constexpr size_t gemm_f32_lds_bytes() {
    return size_t(2) * (64 * WM + 64 * WN) * GEMM_LDS_LD * sizeof(float);
}

// Epilogue (lane holds col = l31, rows frag_row(reg, hi)).  FULL = block tile entirely inside [M, N]: unpredicated,
// back-to-back stores.  Per-column scalars are loaded once and pinned by an empty asm so that hipcc does not re-wait
// vmcnt(0) (which also counts the stores on CDNA4) before every store.
template <int WM, int WN, int EPI, bool FULL>
__device__ __forceinline__ void gemm_f32_epilogue(const GemmArgs& g, f32x16 (&acc)[WM][WN], int m0, int n0, int wr,
                                                  int wc, int l31, int hi) {
    float bv[WN], tg[WN], tb[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        int n = n0 + wc * WN * 32 + j * 32 + l31;
        n = (FULL || n < g.N) ? n : g.N - 1;
        bv[j] = g.bias ? g.bias[n] : 0.f;
        tg[j] = 0.f;
        tb[j] = 0.f;
        if (EPI == EPI_CSL) {
            tg[j] = g.thyp[g.goff + n];
            tb[j] = g.thyp[g.boff + n];
        }
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) asm volatile("" : "+v"(bv[j]), "+v"(tg[j]), "+v"(tb[j]));
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = n0 + wc * WN * 32 + j * 32 + l31;
        if (!FULL && n >= g.N) continue;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * WM * 32 + i * 32 + frag_row(r, hi);
                if (!FULL && m >= g.M) continue;
                float v = acc[i][j][r] + bv[j];
                if (EPI == EPI_BIAS_RELU) v = v > 0.f ? v : 0.f;
                if (EPI == EPI_CSL) {
                    const float* hrow = g.hyp + (size_t)g.rmap.ea(m) * g.hyp_ld;
                    v = fmaf(v, sigmoidf_(hrow[g.goff + n] + tg[j]), hrow[g.boff + n] + tb[j]);
                }
                g.C[(size_t)m * g.ldc + n] = v;
            }
        }
    }
}

template <int WM, int WN, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int LD = GEMM_LDS_LD;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;                   // [2][BM][LD]
    float* Bs = lds + 2 * BM * LD;     // [2][BN][LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wid >> 1, wc = wid & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // global -> register staging map: thread handles float4 column c4 of rows (tid/8 + 32*i)
    const int ld_row = tid >> 3, ld_c4 = tid & 7;
    constexpr int NA = BM / 32, NB = BN / 32;
    const float* aptr[NA];
    const float* bptr[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int r = m0 + ld_row + 32 * i;
        r = r < g.M ? r : g.M - 1;
        aptr[i] = g.A + (size_t)r * g.lda + ld_c4 * 4;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        int r = n0 + ld_row + 32 * i;
        r = r < g.N ? r : g.N - 1;
        bptr[i] = g.W + (size_t)r * g.ldw + ld_c4 * 4;
    }

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[NA], rb[NB];
    const int nk = g.K / GEMM_BK;

    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(aptr[i] + kt * GEMM_BK);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bptr[i] + kt * GEMM_BK);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
            *reinterpret_cast<f32x4*>(&As[(buf * BM + ld_row + 32 * i) * LD + ld_c4 * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i)
            *reinterpret_cast<f32x4*>(&Bs[(buf * BN + ld_row + 32 * i) * LD + ld_c4 * 4]) = rb[i];
    };

    gload(0);
    lstore(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const float* Ab = As + (buf * BM + wr * WM * 32 + l31) * LD + 4 * hi;
        const float* Bb = Bs + (buf * BN + wc * WN * 32 + l31) * LD + 4 * hi;
#pragma unroll
        for (int kk = 0; kk < GEMM_BK / 8; ++kk) {
            f32x4 a[WM], b[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LD + kk * 8);
#pragma unroll
            for (int j = 0; j < WN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LD + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    if (m0 + BM <= g.M && n0 + BN <= g.N)
        gemm_f32_epilogue<WM, WN, EPI, true>(g, acc, m0, n0, wr, wc, l31, hi);
    else
        gemm_f32_epilogue<WM, WN, EPI, false>(g, acc, m0, n0, wr, wc, l31, hi);
}

template <int WM, int WN, int EPI>
inline hipError_t launch_gemm_f32_cfg(const GemmArgs& g, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM);
    size_t lds = gemm_f32_lds_bytes<WM, WN>();
    static DevSeen attr_seen;
    if (auto once_ = first_use_on_device(attr_seen)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<WM, WN, EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, EPI>), grid, dim3(256), lds, st, g);
    return hipGetLastError();
}

template <int EPI>
inline hipError_t launch_gemm_f32(const GemmArgs& g, hipStream_t st) {
    // 128x128 tiles when they still give >= ~2 blocks per CU, else 64x64 (small single-scene batches)
    const long big = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    if (big >= 512) return launch_gemm_f32_cfg<2, 2, EPI>(g, st);
    return launch_gemm_f32_cfg<1, 1, EPI>(g, st);
}

}  // namespace jmid
