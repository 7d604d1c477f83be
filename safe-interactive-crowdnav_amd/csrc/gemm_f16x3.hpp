// "NT" GEMM with fp32-class accuracy at fp16 MFMA rate:  C[M,N] = A[M,K] . W[N,K]^T  (+ epilogue)
//
// Every fp32 operand x is carried as two fp16 planes  hi = fp16(x),  lo = fp16((x - hi) * 2^11)
// (~22 significand bits; the 2^11 scale keeps the residual in the fp16 normal range).  A product is three
// v_mfma_f32_32x32x16_f16 with fp32 accumulation in two accumulators:
//     main += Ahi.Whi          corr += Ahi.Wlo + Alo.Whi          C = main + corr * 2^-11
// (the lo.lo term is 2^-22 relative and dropped).  Peak is 1/3 of the dense fp16 MFMA rate = 833 TFLOP/s,
// 5.3x the exact-fp32 MFMA path.
//
// Operand fragments are 8 consecutive k per lane (lanes 0-31: k 0-7, lanes 32-63: k 8-15 of each 16-wide step);
// A and W use the same per-lane k assignment, which is all the instruction requires.
// Block = 4 waves (2x2), wave tile (WM*32)x(WN*32), BK = 32, double-buffered LDS with 80-byte rows
// (16 consecutive rows land on 16 distinct 16-B slots -> conflict-free ds_read_b128).
//
// The epilogue can emit the result directly in the form the consumer wants: fp32, hi/lo planes, or the packed
// Q / K / V^T planes of the attention kernel (V is stored key-contiguous so that the PV product needs no
// transpose on the way into the MFMA).
#pragma once
#include "common.hpp"
#include "gemm_f32.hpp"

namespace jmid {

typedef _Float16 half_t;
constexpr float kLoScale = 2048.0f;
constexpr float kLoInv = 1.0f / 2048.0f;
constexpr float kHalfMax = 60000.0f;

__device__ __forceinline__ void split_f32(float v, half_t& hi, half_t& lo) {
    hi = (half_t)v;
    lo = (half_t)((v - (float)hi) * kLoScale);
}

enum GemmOut { OUT_F32 = 0, OUT_SPLIT = 1, OUT_QKV = 2 };

struct GemmHArgs {
    const half_t *Ahi, *Alo;  // [M, K], row stride lda (elements)
    const half_t *Whi, *Wlo;  // [N, K], row stride ldw
    const float* bias;
    int M, N, K, lda, ldw;
    float* C;                 // OUT_F32: [M, N] row stride ldc
    half_t *Chi, *Clo;        // OUT_SPLIT: planes [M, N] row stride ldc ; OUT_QKV: Q planes [M, d]
    int ldc;
    half_t *Khi, *Klo;        // OUT_QKV: K planes [M, d]
    half_t *Vthi, *Vtlo;      // OUT_QKV: V^T planes [nseq][nhead][hd][Spad]
    int d, hd, S, Spad;
    const float* hyp;         // EPI_CSL (see gemm_f32.hpp)
    const float* thyp;
    int hyp_ld, goff, boff;
    RowMap rmap;
    int* range_flag;          // set to 1 when an emitted fp16 operand would leave the fp16 range
};

constexpr int GEMMH_BK = 32;
constexpr int GEMMH_LD = 40;  // halfs per LDS row (80 bytes)

template <int WM, int WN>
constexpr size_t gemm_h_lds_bytes() {
    return size_t(2) /*buffers*/ * 2 /*planes*/ * (64 * WM + 64 * WN) * GEMMH_LD * sizeof(half_t);
}

template <int WM, int WN, int EPI, int OUT>
__global__ __launch_bounds__(256, 2) void gemm_f16x3_kernel(GemmHArgs g) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int LD = GEMMH_LD;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(lds_raw);
    // per buffer: Ahi [BM][LD], Alo [BM][LD], Whi [BN][LD], Wlo [BN][LD]
    constexpr int BUF = 2 * (BM + BN) * LD;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wid >> 1, wc = wid & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // staging: 16-byte chunks, 4 per row; chunk id = tid + 256*i -> row = id>>2, c = id&3
    constexpr int NA = BM / 64, NB = BN / 64;
    const int s_row = tid >> 2, s_c = tid & 3;
    const half_t *pah[NA], *pal[NA], *pwh[NB], *pwl[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int r = m0 + s_row + 64 * i;
        r = r < g.M ? r : g.M - 1;
        pah[i] = g.Ahi + (size_t)r * g.lda + s_c * 8;
        pal[i] = g.Alo + (size_t)r * g.lda + s_c * 8;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        int r = n0 + s_row + 64 * i;
        r = r < g.N ? r : g.N - 1;
        pwh[i] = g.Whi + (size_t)r * g.ldw + s_c * 8;
        pwl[i] = g.Wlo + (size_t)r * g.ldw + s_c * 8;
    }
    f16x8 rah[NA], ral[NA], rwh[NB], rwl[NB];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            rah[i] = *reinterpret_cast<const f16x8*>(pah[i] + kt * GEMMH_BK);
            ral[i] = *reinterpret_cast<const f16x8*>(pal[i] + kt * GEMMH_BK);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            rwh[i] = *reinterpret_cast<const f16x8*>(pwh[i] + kt * GEMMH_BK);
            rwl[i] = *reinterpret_cast<const f16x8*>(pwl[i] + kt * GEMMH_BK);
        }
    };
    auto lstore = [&](int buf) {
        half_t* b = lds + buf * BUF;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int off = (s_row + 64 * i) * LD + s_c * 8;
            *reinterpret_cast<f16x8*>(b + off) = rah[i];
            *reinterpret_cast<f16x8*>(b + BM * LD + off) = ral[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int off = (s_row + 64 * i) * LD + s_c * 8;
            *reinterpret_cast<f16x8*>(b + 2 * BM * LD + off) = rwh[i];
            *reinterpret_cast<f16x8*>(b + 2 * BM * LD + BN * LD + off) = rwl[i];
        }
    };

    f32x16 accm[WM][WN], accc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                accm[i][j][r] = 0.f;
                accc[i][j][r] = 0.f;
            }

    const int nk = g.K / GEMMH_BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const half_t* b = lds + buf * BUF;
        const half_t* Ah = b + (wr * WM * 32 + l31) * LD + 8 * hi;
        const half_t* Al = Ah + BM * LD;
        const half_t* Wh = b + 2 * BM * LD + (wc * WN * 32 + l31) * LD + 8 * hi;
        const half_t* Wl = Wh + BN * LD;
#pragma unroll
        for (int ks = 0; ks < GEMMH_BK / 16; ++ks) {
            f16x8 ah[WM], al[WM], wh[WN], wl[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(Ah + i * 32 * LD + ks * 16);
                al[i] = *reinterpret_cast<const f16x8*>(Al + i * 32 * LD + ks * 16);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                wh[j] = *reinterpret_cast<const f16x8*>(Wh + j * 32 * LD + ks * 16);
                wl[j] = *reinterpret_cast<const f16x8*>(Wl + j * 32 * LD + ks * 16);
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], accm[i][j], 0, 0, 0);
                    accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl[j], accc[i][j], 0, 0, 0);
                    accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh[j], accc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds col = l31, rows frag_row(reg, hi)
    bool overflow = false;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = n0 + wc * WN * 32 + j * 32 + l31;
        if (n >= g.N) continue;
        const float bv = g.bias ? g.bias[n] : 0.f;
        float tg = 0.f, tb = 0.f;
        if (EPI == EPI_CSL) {
            tg = g.thyp[g.goff + n];
            tb = g.thyp[g.boff + n];
        }
        // OUT_QKV: which of Q / K / V this column belongs to
        int part = 0, nn = n, vh = 0, vc = 0;
        if (OUT == OUT_QKV) {
            part = n / g.d;
            nn = n - part * g.d;
            vh = nn / g.hd;
            vc = nn - vh * g.hd;
        }
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * WM * 32 + i * 32 + frag_row(r, hi);
                if (m >= g.M) continue;
                float v = accm[i][j][r] + accc[i][j][r] * kLoInv + bv;
                if (EPI == EPI_BIAS_RELU) v = v > 0.f ? v : 0.f;
                if (EPI == EPI_CSL) {
                    const float* hrow = g.hyp + (size_t)g.rmap.ea(m) * g.hyp_ld;
                    v = v * sigmoidf_(hrow[g.goff + n] + tg) + (hrow[g.boff + n] + tb);
                }
                if (OUT == OUT_F32) {
                    g.C[(size_t)m * g.ldc + n] = v;
                } else {
                    half_t h, l;
                    split_f32(v, h, l);
                    overflow |= !(fabsf(v) <= kHalfMax);
                    if (OUT == OUT_SPLIT) {
                        g.Chi[(size_t)m * g.ldc + n] = h;
                        g.Clo[(size_t)m * g.ldc + n] = l;
                    } else {
                        if (part == 0) {
                            g.Chi[(size_t)m * g.d + nn] = h;
                            g.Clo[(size_t)m * g.d + nn] = l;
                        } else if (part == 1) {
                            g.Khi[(size_t)m * g.d + nn] = h;
                            g.Klo[(size_t)m * g.d + nn] = l;
                        } else {
                            const int seq = m / g.S, key = m - seq * g.S;
                            const size_t o = (((size_t)seq * (g.d / g.hd) + vh) * g.hd + vc) * g.Spad + key;
                            g.Vthi[o] = h;
                            g.Vtlo[o] = l;
                        }
                    }
                }
            }
        }
    }
    if (OUT != OUT_F32 && overflow) atomicOr(g.range_flag, 1);
}

template <int WM, int WN, int EPI, int OUT>
inline hipError_t launch_gemm_h_cfg(const GemmHArgs& g, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM);
    size_t lds = gemm_h_lds_bytes<WM, WN>();
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_kernel<WM, WN, EPI, OUT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f16x3_kernel<WM, WN, EPI, OUT>), grid, dim3(256), lds, st, g);
    return hipGetLastError();
}

template <int EPI, int OUT>
inline hipError_t launch_gemm_h(const GemmHArgs& g, hipStream_t st) {
    const long big = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    if (big >= 512) return launch_gemm_h_cfg<2, 2, EPI, OUT>(g, st);
    return launch_gemm_h_cfg<1, 1, EPI, OUT>(g, st);
}

// fp32 -> hi/lo planes (weights at load time, activations produced by fp32-only kernels)
__global__ void split_planes_kernel(const float* in, half_t* hi, half_t* lo, size_t n, int* range_flag) {
    bool overflow = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = in[i];
        half_t h, l;
        split_f32(v, h, l);
        overflow |= !(fabsf(v) <= kHalfMax);
        hi[i] = h;
        lo[i] = l;
    }
    if (overflow && range_flag) atomicOr(range_flag, 1);
}

// hi/lo planes -> fp32 (diagnostics)
__global__ void merge_planes_kernel(const half_t* hi, const half_t* lo, float* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (float)hi[i] + (float)lo[i] * kLoInv;
}

// packed fp32 QKV [M, 3d] -> Q/K planes [M, d] + V^T planes [nseq][nhead][hd][Spad] (diagnostics; the pipeline
// gets these straight from the QKV GEMM epilogue)
__global__ void qkv_to_planes_kernel(const float* qkv, half_t* qh, half_t* ql, half_t* kh, half_t* kl, half_t* vth,
                                     half_t* vtl, size_t M, int d, int hd, int S, int Spad) {
    const size_t n = M * d;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / d;
        const int c = (int)(i % d);
        half_t h, l;
        split_f32(qkv[m * 3 * d + c], h, l);
        qh[i] = h; ql[i] = l;
        split_f32(qkv[m * 3 * d + d + c], h, l);
        kh[i] = h; kl[i] = l;
        split_f32(qkv[m * 3 * d + 2 * d + c], h, l);
        const size_t seq = m / S, key = m % S;
        const int head = c / hd, vc = c % hd;
        const size_t o = ((seq * (d / hd) + head) * hd + vc) * Spad + key;
        vth[o] = h; vtl[o] = l;
    }
}

}  // namespace jmid
