// "NT" GEMM with fp32-class accuracy at fp16 MFMA rate:  C[M,N] = A[M,K] . W[N,K]^T  (+ epilogue)
//
// Every fp32 operand x is carried as two fp16 planes  hi = fp16(x),  lo = fp16(x - hi)  (~22 significand bits down
// to 2^-25 absolute: v_mfma_f32_32x32x16_f16 takes fp16 subnormals exactly, tools/mfma_denorm.hip).  Weights are
// multiplied by 2^8 before the split so that their residuals stay in the fp16 normal range (a weight of 0.04 would
// otherwise keep only ~19 bits); the epilogue multiplies by 2^-8, which is exact.  A product is three MFMAs into
// ONE fp32 accumulator:      acc += Ahi.Whi ; acc += Ahi.Wlo ; acc += Alo.Whi
// (the lo.lo term is 2^-22 relative and dropped).  Operand representation error 8e-8 relative on a K = 512 product
// (fp32 inputs: 6e-8).  Peak is 1/3 of the dense fp16 MFMA rate = 833 TFLOP/s, 5.3x the exact-fp32 MFMA path.
// JMID_PREC_F16X2 (template parameter X2 of every kernel here): the Alo.Whi term is left out - the activation enters
// as fp16, the weights stay exact - and the A_lo images are neither copied into LDS nor written by the epilogues
// whose consumer is such a GEMM.  Two MFMAs per product, 1250 TFLOP/s peak; parity in docs/NOTEBOOK.md section 3.
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
#include <type_traits>
#include "common.hpp"
#include "gemm_f32.hpp"

namespace jmid {

typedef _Float16 half_t;
constexpr float kWScale = 256.0f;           // weights are split as W * 2^8 ...
constexpr float kWInv = 1.0f / 256.0f;      // ... and every GEMM epilogue scales the accumulator back (exact)
constexpr float kHalfMax = 60000.0f;

__device__ __forceinline__ void split_f32(float v, half_t& hi, half_t& lo) {
    // v must be an opaque fp32 value here: when v = a * b (or fma) hipcc fuses ONE of the two uses of fp16(v) into
    // v_fma_mixlo_f16 (single rounding from the exact product) and converts the other from the rounded fp32 - the
    // stored hi and the hi inside lo then differ by one fp16 ulp at near-ties (seen as 2^-12 errors in 1 of 25 000
    // attention outputs)
    asm("" : "+v"(v));
    hi = (half_t)v;
    lo = (half_t)(v - (float)hi);
}
__device__ __forceinline__ void split_f32_unscaled(float v, half_t& hi, half_t& lo) { split_f32(v, hi, lo); }
// bf8 (e5m2) image of one fp16 value: its top byte after rounding to nearest.  The carry of the rounding turns magnitudes from
// 61440 (0x7B80) on into 0x7C = Inf: every producer of a split plane flags |v| > kHalfMax = 60000 (0x7B53) as JMID_ERANGE, so an
// operand that reaches this conversion is below the carry.
__device__ __forceinline__ unsigned char bf8_of_f16(half_t v) {
    return (unsigned char)((__builtin_bit_cast(unsigned short, v) + 0x80u) >> 8);
}

// hi / lo planes of FOUR fp32 values, two per conversion instruction (v_cvt_pk_f16_f32 rounds to nearest even like the scalar
// conversion: the same bits as four split_f32 calls) - 16 instead of ~28 VALU instructions, range check included.  The check
// accumulates the largest |hi| as 15-bit patterns (packed 16-bit max): a value beyond kHalfMax, an Inf or a NaN ends above
// kHalfMaxBits.  (Epilogues that split value by value spend more VALU time on this than on their stores.)
typedef int i32x2_s __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2_s __attribute__((ext_vector_type(2)));
constexpr unsigned kHalfMaxBits = 0x7B53;      // fp16(60000)
struct Split4 {
    i32x2_s hi, lo;      // packed fp16 pairs: {v0, v1}, {v2, v3}
};
__device__ __forceinline__ Split4 split_f32x4(float a, float b, float c, float d, unsigned& amax16) {
    asm("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));      // opaque values, as in split_f32
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const f16x2 h01 = __builtin_convertvector(f32x2_{a, b}, f16x2), h23 = __builtin_convertvector(f32x2_{c, d}, f16x2);
    const f16x2 l01 = __builtin_convertvector(f32x2_{a - (float)h01[0], b - (float)h01[1]}, f16x2);
    const f16x2 l23 = __builtin_convertvector(f32x2_{c - (float)h23[0], d - (float)h23[1]}, f16x2);
    Split4 r;
    r.hi = i32x2_s{__builtin_bit_cast(int, h01), __builtin_bit_cast(int, h23)};
    r.lo = i32x2_s{__builtin_bit_cast(int, l01), __builtin_bit_cast(int, l23)};
    u16x2_s m = __builtin_bit_cast(u16x2_s, amax16);
    m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2_s, (unsigned)r.hi[0] & 0x7fff7fffu));
    m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2_s, (unsigned)r.hi[1] & 0x7fff7fffu));
    amax16 = __builtin_bit_cast(unsigned, m);
    return r;
}
__device__ __forceinline__ bool split_range_exceeded(unsigned amax16) {
    return (amax16 & 0xffffu) > kHalfMaxBits || (amax16 >> 16) > kHalfMaxBits;
}

// epilogue stores of streamed outputs (the staged Q / K / V^T rows of the in_proj GEMM: written once, read by another kernel) carry
// the non-temporal hint, so that they do not push the launch's operand panels out of the XCD's L2: a 51-episode f16mx call 113.7 ->
// 112.8 ms, f16x3 unchanged, same bits (tools/ab_builds.py; -DJMID_NO_NT_STORES for the A/B)
template <typename T>
__device__ __forceinline__ void store_stream(T* p, const T& v) {
#ifdef JMID_NO_NT_STORES
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}

enum GemmOut { OUT_F32 = 0, OUT_SPLIT = 1, OUT_QKV = 2, OUT_LNX = 4 };     // OUT_LNX (gemm_small.hpp only): + residual + LayerNorm, every workgroup normalising its own
                                                                        // 64 columns after exchanging the row statistics with the seven others of its row tile


struct GemmHArgs {
    const half_t *Ahi, *Alo;  // [M, K] in the blocked panel layout (common.hpp::blk_index), rows padded to 128
    const half_t *Whi, *Wlo;  // [N, K] blocked
    const float* bias;
    int M, N, K;
    float* C;                 // OUT_F32: [M, N] row stride ldc
    half_t *Chi, *Clo;        // OUT_SPLIT: planes [M, N] blocked (operand of the next GEMM) ; OUT_QKV: Q planes [M, d] row-major
    int ldc;
    half_t *Khi, *Klo;        // OUT_QKV: K planes [M, d]
    half_t *Vthi, *Vtlo;      // OUT_QKV: V planes [M, d] (row-major; transposed afterwards by v_transpose_kernel), or
                              // with vt_direct the final V^T planes [nseq][nhead][hd][Spad] (see v_transpose_kernel)
    int d, hd, S, Spad;
    int vt_direct;            // OUT_QKV, needs S % 4 == 0: the epilogue writes V^T itself, 4 keys (8 bytes) per store
    float qscale;             // OUT_QKV: Q is stored pre-multiplied by log2(e)/sqrt(head_dim)
    const float* hyp;         // EPI_CSL (see gemm_f32.hpp)
    const float* thyp;
    int hyp_ld, goff, boff;
    RowMap rmap;
    int* range_flag;          // set to 1 when an emitted fp16 operand would leave the fp16 range
    int x2;                   // JMID_PREC_F16X2: two-term product A_hi x (W_hi + W_lo)
    const unsigned char* W8;  // JMID_PREC_F16MX: bf8(W_lo) in MFMA-fragment order (w8_image_kernel), or null
    unsigned char *K8h, *K8l; // OUT_QKV, JMID_PREC_F16MX with head_dim 128: bf8 images of K_hi / K_lo, [M, d] bytes each, written
                              // INSTEAD of the fp16 K_lo plane (attn_f16x3_dma_kernel<.., MX>); null: K_lo as fp16
    unsigned char* Q8l;       // with them: bf8 image of Q_lo, [M, d] bytes, INSTEAD of the fp16 Q_lo plane (all that kernel wants of Q_lo)
    // OUT_LNX (gemm_small.hpp, N = 512): X <- LayerNorm(X + A . W^T + bias) * gamma + beta
    const float *ln_gamma, *ln_beta;
    half_t *ln_xh, *ln_xl;    // residual stream planes (blocked); F16MX at d_model 512: ln_xl8 instead of ln_xl
    unsigned char* ln_xl8;    // bf8 image of the lo plane (gemm_ln2_mx.hpp::blk8_index), or null
    float ln_eps;
    int ln_no_lo;             // the last LayerNorm of the net: nobody reads its lo plane (F16MX)
    // the row-statistics exchange of the workgroups of a row tile
    unsigned long long* ln_xchg;   // [2 kinds][ceil(M / 64)][8 column tiles][64 rows] granules {fp32 partial, launch tag}, zeroed once per call
    unsigned ln_epoch;             // this launch's tag: never 0, never repeated within a call
    int ln_polls;                  // poll budget of a wait for a partner workgroup (0: gemm_small.hpp's SM_LNX_POLLS; tests shrink it)
    int ln_withhold;               // diagnostics flavour: block 7 of row tile 0 never publishes its statistics (tests: the give-up path)
    // OUT_LNX of the attention out-projection after a split-KV attention launch: the merge of the partial outputs (attn_combine_kernel's
    // arithmetic) happens in THIS launch - every workgroup merges its 64 rows x 64 columns of the A operand, publishes a flag, waits
    // for the seven others of its row tile and only then starts its K loop
    const float* cmb_O;            // [cmb_ns][cmb_Mtot][d] fp32 partial outputs, or null: A is ready
    const float* cmb_ML;           // [cmb_ns][cmb_Mtot][nhead][2] their (max, sum)
    int cmb_ns, cmb_nhead;
    unsigned cmb_Mtot;
};

constexpr int GEMMH_BK = 32;
constexpr int GEMMH_LD = 40;  // halfs per LDS row (80 bytes)

template <int WM, int WN>
constexpr size_t gemm_h_lds_bytes() {
    return size_t(2) /*buffers*/ * 2 /*planes*/ * (64 * WM + 64 * WN) * GEMMH_LD * sizeof(half_t);
}

// three passes over the wave's tiles so that consecutive MFMAs into the same accumulator are WM*WN instructions
// apart (a back-to-back dependent pair stalls for the MFMA latency)
// SWAP: the W fragments go first - the accumulators hold the TRANSPOSED tile (a lane then owns one token row per 32-row block and
// four consecutive columns per register group: rowwise epilogues below).  Same products, same order.
template <int WM, int WN, bool X2, bool SWAP = false>
__device__ __forceinline__ void mfma3(const f16x8 (&ah)[WM], const f16x8 (&al)[WM], const f16x8 (&wh)[WN],
                                      const f16x8 (&wl)[WN], f32x16 (&acc)[WM][WN]) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
            acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], ah[i], acc[i][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
            acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], ah[i], acc[i][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl[j], acc[i][j], 0, 0, 0);
    if (X2) return;   // JMID_PREC_F16X2: the activation's lo plane stays out of the product
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
            acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], al[i], acc[i][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh[j], acc[i][j], 0, 0, 0);
}

// Epilogue.  FULL = the whole block tile is inside [M, N] (block-uniform): no per-element predication, so the
// stores issue back to back.  The per-column scalars (bias, time part of the hyper nets) are loaded once and pinned
// with an empty asm: otherwise hipcc re-waits `vmcnt(0)` before every use inside the store loop, and since stores
// count on vmcnt too (CDNA4) every store would wait for the previous one to complete.
template <int WM, int WN, int EPI, int OUT, bool X2, bool FULL, bool K8 = false>
__device__ __forceinline__ void gemm_h_epilogue_impl(const GemmHArgs& g, f32x16 (&accm)[WM][WN],
                                                     int m0, int n0, int wr, int wc, int l31, int hi) {
    bool overflow = false;
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
        // OUT_QKV: which of Q / K / V this column belongs to
        int part = 0, nn = n;
        if (OUT == OUT_QKV) {
            part = n / g.d;
            nn = n - part * g.d;
        }
        if (OUT == OUT_QKV && part == 2 && g.vt_direct) {
            // V^T straight from the accumulators: registers 4q..4q+3 of a lane are 4 consecutive tokens (= keys) of
            // column nn, i.e. one 8-byte granule of V^T row (head, nn % hd).  Granules never straddle a sequence
            // (S % 4 == 0).  Granule order inside a 16-key group: common.hpp::vt_key_pos (same as v_transpose_kernel).
            const int head = nn / g.hd, vc = nn - head * g.hd, nh = g.d / g.hd;
#pragma unroll
            for (int i = 0; i < WM; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = m0 + wr * WM * 32 + i * 32 + 8 * q + 4 * hi;
                    if (!FULL && m >= g.M) continue;
                    f16x4 vh, vl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = fmaf(accm[i][j][4 * q + e], kWInv, bv[j]);
                        half_t hh, ll;
                        split_f32_unscaled(v, hh, ll);
                        overflow |= !(fabsf(v) <= kHalfMax);
                        vh[e] = hh;
                        vl[e] = ll;
                    }
                    const int seq = m / g.S, key = m - seq * g.S;
                    const size_t o = (((size_t)seq * nh + head) * g.hd + vc) * g.Spad + vt_key_pos(key);
                    *reinterpret_cast<f16x4*>(g.Vthi + o) = vh;
                    if (!X2) *reinterpret_cast<f16x4*>(g.Vtlo + o) = vl;   // F16X2: P.V takes V_hi only
                }
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * WM * 32 + i * 32 + frag_row(r, hi);
                if (!FULL && m >= g.M) continue;
                float v = fmaf(accm[i][j][r], kWInv, bv[j]);
                if (EPI == EPI_BIAS_RELU) v = v > 0.f ? v : 0.f;
                if (EPI == EPI_CSL) {
                    const float* hrow = g.hyp + (size_t)g.rmap.ea(m) * g.hyp_ld;
                    v = fmaf(v, sigmoidf_(hrow[g.goff + n] + tg[j]), hrow[g.boff + n] + tb[j]);
                }
                if (OUT == OUT_F32) {
                    g.C[(size_t)m * g.ldc + n] = v;
                } else {
                    half_t h, l;
                    if (OUT == OUT_QKV) {
                        if (part == 0) v *= g.qscale;
                        split_f32_unscaled(v, h, l);
                    } else {
                        split_f32(v, h, l);
                    }
                    overflow |= !(fabsf(v) <= kHalfMax);
                    if (OUT == OUT_SPLIT) {
                        const size_t o = blk_index(m, n, g.N);
                        g.Chi[o] = h;
                        if (!X2) g.Clo[o] = l;   // F16X2: the consumer GEMM takes A_hi only
                    } else {
                        if (part == 0) {
                            g.Chi[(size_t)m * g.d + nn] = h;
                            if (K8 && g.Q8l)
                                g.Q8l[(size_t)m * g.d + nn] = bf8_of_f16(l);
                            else
                                g.Clo[(size_t)m * g.d + nn] = l;
                        } else if (part == 1) {
                            g.Khi[(size_t)m * g.d + nn] = h;
                            if (K8 && g.K8h) {      // K8: compile-time, only the F16MX kernels (the byte stores cost the others registers)
                                g.K8h[(size_t)m * g.d + nn] = bf8_of_f16(h);
                                g.K8l[(size_t)m * g.d + nn] = bf8_of_f16(l);
                            } else {
                                g.Klo[(size_t)m * g.d + nn] = l;
                            }
                        } else {   // V row-major planes; v_transpose_kernel makes them key-contiguous
                            g.Vthi[(size_t)m * g.d + nn] = h;
                            if (!X2) g.Vtlo[(size_t)m * g.d + nn] = l;
                        }
                    }
                }
            }
        }
    }
    if (OUT != OUT_F32 && overflow) atomicOr(g.range_flag, 1);
}

template <int WM, int WN, int EPI, int OUT, bool X2 = false, bool K8 = false>
__device__ __forceinline__ void gemm_h_epilogue(const GemmHArgs& g, f32x16 (&accm)[WM][WN],
                                                int m0, int n0, int wr, int wc, int l31, int hi, int bm = 64 * WM,
                                                int bn = 64 * WN) {
    if (m0 + bm <= g.M && n0 + bn <= g.N)
        gemm_h_epilogue_impl<WM, WN, EPI, OUT, X2, true, K8>(g, accm, m0, n0, wr, wc, l31, hi);
    else
        gemm_h_epilogue_impl<WM, WN, EPI, OUT, X2, false, K8>(g, accm, m0, n0, wr, wc, l31, hi);
}

// Row-wise epilogue (ConcatSquash; also plain bias + ReLU) for accumulators of the TRANSPOSED product (W fragments as the first MFMA operand): a lane holds ONE
// token row per 32-row block and 4 consecutive columns per register group, so the (episode, agent) row of the hyper buffer -
// three integer divisions - is found once per row instead of once per element, gate / bias / time vectors come as 16-byte
// loads, and the result leaves as 16-byte (fp32) or 8-byte (fp16 planes) stores.  Per element the arithmetic of
// gemm_h_epilogue_impl<.., EPI_CSL, ..>: the same bits.
template <int WM, int WN, int EPI, int OUT, bool X2>
__device__ __forceinline__ void csl_swapped_epilogue(const GemmHArgs& g, f32x16 (&acc)[WM][WN], int mw0, int nw0, int l31, int hi) {
    bool overflow = false;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int m = mw0 + i * 32 + l31;
        if (m >= g.M) continue;
        const float* hrow = EPI == EPI_CSL ? g.hyp + (size_t)g.rmap.ea(m) * g.hyp_ld : nullptr;
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = nw0 + j * 32 + 8 * q + 4 * hi;
                if (c0 >= g.N) continue;          // N is a multiple of 4 (d_mid, d_low of the net; 128 in the F16MX kernels)
                const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + c0);
                f32x4 tg, tb, hg, hb;
                if (EPI == EPI_CSL) {
                    tg = *reinterpret_cast<const f32x4*>(g.thyp + g.goff + c0);
                    tb = *reinterpret_cast<const f32x4*>(g.thyp + g.boff + c0);
                    hg = *reinterpret_cast<const f32x4*>(hrow + g.goff + c0);
                    hb = *reinterpret_cast<const f32x4*>(hrow + g.boff + c0);
                }
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = fmaf(acc[i][j][4 * q + e], kWInv, bv[e]);
                    if (EPI == EPI_BIAS_RELU) v = v > 0.f ? v : 0.f;
                    if (EPI == EPI_CSL) v = fmaf(v, sigmoidf_(hg[e] + tg[e]), hb[e] + tb[e]);
                    o[e] = v;
                }
                if (OUT == OUT_F32) {
                    *reinterpret_cast<f32x4*>(g.C + (size_t)m * g.ldc + c0) = o;
                } else {
                    f16x4 vh, vl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        half_t hh, ll;
                        split_f32(o[e], hh, ll);
                        overflow |= !(fabsf(o[e]) <= kHalfMax);
                        vh[e] = hh;
                        vl[e] = ll;
                    }
                    const size_t ob = blk_index(m, c0, g.N);
                    *reinterpret_cast<f16x4*>(g.Chi + ob) = vh;
                    if (!X2) *reinterpret_cast<f16x4*>(g.Clo + ob) = vl;
                }
            }
    }
    if (OUT != OUT_F32 && overflow) atomicOr(g.range_flag, 1);
}

// the ConcatSquash GEMMs (EPI_CSL into fp32 or planes) of every kernel below run transposed with the row-wise epilogue
template <int EPI, int OUT>
constexpr bool csl_rowwise() { return EPI == EPI_CSL && (OUT == OUT_F32 || OUT == OUT_SPLIT); }


template <int WM, int WN, int EPI, int OUT, bool X2 = false>
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
    for (int i = 0; i < NA; ++i) {   // planes are padded to 128 rows: no clamping needed
        const size_t o = blk_index(m0 + s_row + 64 * i, s_c * 8, g.K);
        pah[i] = g.Ahi + o;
        pal[i] = g.Alo + o;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const size_t o = blk_index(n0 + s_row + 64 * i, s_c * 8, g.K);
        pwh[i] = g.Whi + o;
        pwl[i] = g.Wlo + o;
    }
    f16x8 rah[NA], ral[NA], rwh[NB], rwl[NB];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            rah[i] = *reinterpret_cast<const f16x8*>(pah[i] + (size_t)kt * 4096);
            ral[i] = *reinterpret_cast<const f16x8*>(pal[i] + (size_t)kt * 4096);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            rwh[i] = *reinterpret_cast<const f16x8*>(pwh[i] + (size_t)kt * 4096);
            rwl[i] = *reinterpret_cast<const f16x8*>(pwl[i] + (size_t)kt * 4096);
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

    f32x16 accm[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                accm[i][j][r] = 0.f;
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
            mfma3<WM, WN, X2, csl_rowwise<EPI, OUT>()>(ah, al, wh, wl, accm);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    if constexpr (csl_rowwise<EPI, OUT>()) {
        csl_swapped_epilogue<WM, WN, EPI, OUT, X2>(g, accm, m0 + wr * WM * 32, n0 + wc * WN * 32, l31, hi);
        return;
    }
    gemm_h_epilogue<WM, WN, EPI, OUT, X2>(g, accm, m0, n0, wr, wc, l31, hi);
}

template <int WM, int WN, int EPI, int OUT, bool X2 = false>
inline hipError_t launch_gemm_h_cfg(const GemmHArgs& g, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM);
    size_t lds = gemm_h_lds_bytes<WM, WN>();
    static DevSeen attr_seen;
    if (auto once_ = first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_kernel<WM, WN, EPI, OUT, X2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((gemm_f16x3_kernel<WM, WN, EPI, OUT, X2>), grid, dim3(256), lds, st, g);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant (global_load_lds_dwordx4): operand tiles go HBM/L2 -> LDS without passing through VGPRs, into a
// 4-stage ring with three K-tiles in flight and ONE barrier per K-tile (counted vmcnt, raw s_barrier).
// 128x128 tile, BK = 32, 4 waves, 128 KB of LDS (one workgroup per CU).
// A DMA writes LDS lane-linearly (wave base + lane*16), so rows are unpadded 64-byte lines and the bank-conflict
// swizzle lives on the SOURCE address: 16-byte chunk c of row r is stored at chunk  c ^ ((r>>2)&3).
// blockIdx is remapped so that each XCD (block b runs on XCD b%8) walks a contiguous range of tiles and the A
// row-panel shared by consecutive N-tiles stays in that XCD's L2.
constexpr int DMA_STAGES = 4;
constexpr int DMA_PLANE = 128 * 32;                       // halfs per plane per stage (8 KB)
constexpr int DMA_STAGE = 4 * DMA_PLANE;                  // Ahi, Alo, Whi, Wlo
constexpr size_t DMA_LDS_BYTES = size_t(DMA_STAGES) * DMA_STAGE * sizeof(half_t);

template <int EPI, int OUT, bool X2 = false>
__global__ __launch_bounds__(256, 1) void gemm_f16x3_dma_kernel(GemmHArgs g, int ntm, int ntn) {
    constexpr int WM = 2, WN = 2, BM = 128, BN = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wid >> 1, wc = wid & 1;
    // XCD-aware tile order (bijective for any grid size)
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    // L2-aware order inside the XCD's range: N-tiles in groups of NG, all M-tiles of a group before the next group.
    // The 32 CUs of an XCD then work on (8 M-tiles) x (NG N-tiles): the W group (<= 1 MB) stays in the 4 MB L2 and
    // every A row-panel is fetched from Infinity Cache / HBM once per group instead of once per N-tile.
    const int NG = (ntn % 4 == 0) ? 4 : (ntn % 3 == 0) ? 3 : (ntn % 2 == 0) ? 2 : 1;
    const int per_group = ntm * NG;
    const int grp = swz / per_group, rem = swz - grp * per_group;
    const int tm = rem / NG, tn = grp * NG + (rem - tm * NG);
    const int m0 = tm * BM, n0 = tn * BN;

    // DMA sources: the operand planes are stored in the blocked panel layout, so K-tile kt of this block's A (W)
    // panel is one contiguous 8 KB image per plane; round i (0..7) = plane i>>1, half i&1, and thread t copies the
    // 16 bytes at offset (i&1)*4 KB + t*16 of that image - 1 KB of contiguous full cache lines per wave-instruction.
    const int nk = g.K / GEMMH_BK;
    const half_t* src[4];
    src[0] = g.Ahi + (size_t)tm * nk * 4096 + tid * 8;
    src[1] = g.Alo + (size_t)tm * nk * 4096 + tid * 8;
    src[2] = g.Whi + (size_t)tn * nk * 4096 + tid * 8;
    src[3] = g.Wlo + (size_t)tn * nk * 4096 + tid * 8;
    auto issue = [&](int kt) {
        half_t* st = lds + (kt & (DMA_STAGES - 1)) * DMA_STAGE + wid * 512;   // wave-uniform base (+ lane*16 B by HW)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src[i >> 1] + (size_t)kt * 4096 + (i & 1) * 2048),
                (__attribute__((address_space(3))) void*)(st + i * 2048), 16, 0, 0);
    };

    f32x16 accm[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                accm[i][j][r] = 0.f;
            }

    issue(0);
    if (nk > 1) issue(1);
    if (nk > 2) issue(2);
    // fragment read offsets (halfs) inside a plane: row*32 + (chunk ^ swz)*8
    int offA[WM][2], offW[WN][2];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int row = wr * 64 + i * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offA[i][ks] = row * 32 + (((ks * 2 + hi) ^ ((row >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int row = wc * 64 + j * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offW[j][ks] = row * 32 + (((ks * 2 + hi) ^ ((row >> 2) & 3)) * 8);
    }

    for (int kt = 0; kt < nk; ++kt) {
        // this wave's DMAs of tile kt have landed once at most (tiles still wanted in flight) * 8 remain outstanding
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();             // every wave's part of tile kt landed; stage (kt-1)&3 is free
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 3 < nk) issue(kt + 3);
        const half_t* st = lds + (kt & (DMA_STAGES - 1)) * DMA_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah[WM], al[WM], wh[WN], wl[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(st + offA[i][ks]);
                al[i] = *reinterpret_cast<const f16x8*>(st + DMA_PLANE + offA[i][ks]);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                wh[j] = *reinterpret_cast<const f16x8*>(st + 2 * DMA_PLANE + offW[j][ks]);
                wl[j] = *reinterpret_cast<const f16x8*>(st + 3 * DMA_PLANE + offW[j][ks]);
            }
            mfma3<WM, WN, X2, csl_rowwise<EPI, OUT>()>(ah, al, wh, wl, accm);
        }
    }
    if constexpr (csl_rowwise<EPI, OUT>()) {
        csl_swapped_epilogue<WM, WN, EPI, OUT, X2>(g, accm, m0 + wr * WM * 32, n0 + wc * WN * 32, l31, hi);
        return;
    }
    gemm_h_epilogue<WM, WN, EPI, OUT, X2>(g, accm, m0, n0, wr, wc, l31, hi);
}

template <int EPI, int OUT, bool X2 = false>
inline hipError_t launch_gemm_h_dma(const GemmHArgs& g, hipStream_t st) {
    const int ntm = (g.M + 127) / 128, ntn = (g.N + 127) / 128;
    static DevSeen attr_seen;
    if (auto once_ = first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_dma_kernel<EPI, OUT, X2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)DMA_LDS_BYTES);
    }
    hipLaunchKernelGGL((gemm_f16x3_dma_kernel<EPI, OUT, X2>), dim3(ntm * ntn), dim3(256), DMA_LDS_BYTES, st, g, ntm, ntn);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// 256x128 LDS-DMA variant: 8 waves (4 along M x 2 along N, two per SIMD so that one wave's MFMAs cover the other's
// LDS-read latency and barrier skew), 48 KB stages, 3-stage ring (two K-tiles in flight), 144 KB of LDS.
// 25 % fewer operand bytes per FLOP than the 128x128 tile.  Each DMA round copies one whole 8 KB panel image.
constexpr int DMA2_STAGES = 3;
constexpr int DMA2_STAGE = 6 * DMA_PLANE;                 // Ahi(2 images), Alo(2), Whi, Wlo
constexpr size_t DMA2_LDS_BYTES = size_t(DMA2_STAGES) * DMA2_STAGE * sizeof(half_t);

template <int EPI, int OUT, bool X2 = false>
__global__ __launch_bounds__(512, 2) void gemm_f16x3_dma256_kernel(GemmHArgs g, int ntm, int ntn, int ng_req, int abl) {
    constexpr int WM = 2, WN = 2, BM = 256, BN = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wid >> 1, wc = wid & 1;                 // wr 0..3, wc 0..1
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const int NG = (ng_req > 0 && ntn % ng_req == 0) ? ng_req
                   : (ntn % 4 == 0) ? 4 : (ntn % 3 == 0) ? 3 : (ntn % 2 == 0) ? 2 : 1;
    const int per_group = ntm * NG;
    const int grp = swz / per_group, rem = swz - grp * per_group;
    const int tm = rem / NG, tn = grp * NG + (rem - tm * NG);
    const int m0 = tm * BM, n0 = tn * BN;

    const int nk = g.K / GEMMH_BK;
    const int nrb = (g.M + 127) / 128;                     // allocated 128-row panels of A
    const int rb0 = 2 * tm, rb1 = (2 * tm + 1 < nrb) ? 2 * tm + 1 : nrb - 1;
    const half_t* src[6];
    src[0] = g.Ahi + (size_t)rb0 * nk * 4096 + tid * 8;
    src[1] = g.Ahi + (size_t)rb1 * nk * 4096 + tid * 8;
    src[2] = g.Alo + (size_t)rb0 * nk * 4096 + tid * 8;
    src[3] = g.Alo + (size_t)rb1 * nk * 4096 + tid * 8;
    src[4] = g.Whi + (size_t)tn * nk * 4096 + tid * 8;
    src[5] = g.Wlo + (size_t)tn * nk * 4096 + tid * 8;
    auto issue = [&](int kt, int stage) {
        half_t* st = lds + stage * DMA2_STAGE + wid * 512;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (X2 && (i == 2 || i == 3)) continue;   // F16X2: no A_lo images (4 instructions per tile: vmcnt 4 below)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)kt * 4096),
                                             (__attribute__((address_space(3))) void*)(st + i * 4096), 16, 0, 0);
        }
    };

    f32x16 accm[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                accm[i][j][r] = 0.f;
            }
    // fragment read offsets (halfs).  A rows 0-127 live in image 0, rows 128-255 in image 1; lo planes 2 images later.
    int offA[WM][2], offW[WN][2];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int row = wr * 64 + i * 32 + l31;            // 0..255
        const int r = row & 127;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            offA[i][ks] = (row >> 7) * DMA_PLANE + r * 32 + (((ks * 2 + hi) ^ ((r >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int row = wc * 64 + j * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offW[j][ks] = row * 32 + (((ks * 2 + hi) ^ ((row >> 2) & 3)) * 8);
    }

    // abl: timing ablations (diagnostics, results wrong): 1 = no fragment reads / MFMA, 2 = fragment reads but no MFMA,
    //      4 = no DMA after the first two tiles
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (!(kt + 1 < nk && !(abl & 4))) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (X2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk && !(abl & 4)) issue(kt + 2, stage == 0 ? 2 : stage - 1);   // (stage + 2) % 3
        const half_t* st = lds + stage * DMA2_STAGE;
        if (!(abl & 1))
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah[WM], al[WM], wh[WN], wl[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(st + offA[i][ks]);
                if (!X2) al[i] = *reinterpret_cast<const f16x8*>(st + 2 * DMA_PLANE + offA[i][ks]);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                wh[j] = *reinterpret_cast<const f16x8*>(st + 4 * DMA_PLANE + offW[j][ks]);
                wl[j] = *reinterpret_cast<const f16x8*>(st + 5 * DMA_PLANE + offW[j][ks]);
            }
            if (abl & 2) {
#pragma unroll
                for (int i = 0; i < WM; ++i) asm volatile("" ::"v"(ah[i]), "v"(wh[i]), "v"(wl[i]));
            } else {
                mfma3<WM, WN, X2, csl_rowwise<EPI, OUT>()>(ah, al, wh, wl, accm);
            }
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
    if constexpr (csl_rowwise<EPI, OUT>()) {
        csl_swapped_epilogue<WM, WN, EPI, OUT, X2>(g, accm, m0 + wr * WM * 32, n0 + wc * WN * 32, l31, hi);
        return;
    }
    gemm_h_epilogue<WM, WN, EPI, OUT, X2>(g, accm, m0, n0, wr, wc, l31, hi, BM, BN);
}

template <int EPI, int OUT, bool X2 = false>
inline hipError_t launch_gemm_h_dma256(const GemmHArgs& g, hipStream_t st) {
    const int ntm = (g.M + 255) / 256, ntn = (g.N + 127) / 128;
    static DevSeen attr_seen;
    if (auto once_ = first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_dma256_kernel<EPI, OUT, X2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)DMA2_LDS_BYTES);
    }
    hipLaunchKernelGGL((gemm_f16x3_dma256_kernel<EPI, OUT, X2>), dim3(ntm * ntn), dim3(512), DMA2_LDS_BYTES, st, g, ntm, ntn,
                       tune().gemm_ng, gemm_abl_bits());
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// V^T through LDS (256x256 kernel, OUT_QKV, head_dim 128).  The direct V^T epilogue above writes 8 bytes per lane: one
// wave instruction touches 32 V^T rows with 16 contiguous bytes each, and a 128-byte line is assembled from eight such
// pieces - the V^T stores alone cost 40 us (F16X2) / 87 us (F16X3) of a 61 200-token QKV launch (a build without them:
// 172.5 -> 166.4 / 215.7 -> 202.6 ms per 51-episode call).  Here a wave's 64 keys x 128 head-dim tile is transposed in
// the wave's own 16 KB of the (finished) operand ring - rows = head dim, 128 contiguous bytes = 64 keys in the
// vt_key_pos order, 16-byte chunks XOR-swizzled by the row - and written with 16 bytes per lane: eight lanes cover the
// 128 contiguous bytes of one V^T row.  Same values as the direct path (same arithmetic per element).
// Needs the wave's 64 tokens inside one sequence, starting at a multiple of 16 keys; otherwise returns false.
template <bool X2>
__device__ __forceinline__ bool vt_staged_store(const GemmHArgs& g, f32x16 (&acc)[2][4], int mrow0, int ncol0, half_t* wlds,
                                                int l31, int hi, int lane) {
    const int seq = mrow0 / g.S, key0 = mrow0 - seq * g.S;
    if (mrow0 + 64 > g.M || key0 + 64 > g.S || (key0 & 15) != 0) return false;
    const int nn0 = ncol0 - 2 * g.d, head = nn0 / g.hd, nh = g.d / g.hd;
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = g.bias ? g.bias[ncol0 + j * 32 + l31] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bv[j]));
    // The values once: bias, hi / lo split four at a time into packed fp16 pairs.  In two halves of 64 V^T rows (j = 2 jh, 2 jh + 1):
    // the hi pairs go straight into the wave's LDS tile, the lo pairs of the half (32 integer registers) wait for its second
    // plane pass - no more registers than the accumulators they replace.  (Packed pairs must not be parked in FLOAT registers:
    // measured wrong - a pair can be a denormal / NaN pattern as an fp32.)
    unsigned amax16 = 0;
    half_t* dst_base[2] = {g.Vthi, g.Vtlo};
    auto lds_at = [&](int i, int j, int q) {
        const int vc = j * 32 + l31;
        const int pos = vt_key_pos(i * 32 + 8 * q + 4 * hi);      // first of 4 consecutive stored positions
        return reinterpret_cast<f16x4*>(wlds + vc * 64 + ((((pos >> 3) ^ (vc & 7)) << 3) | (pos & 4)));
    };
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
        i32x2_s lo_pk[2][2][4];      // [j - 2 jh][i][q]
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = 2 * jh + jj;
                    const Split4 sp = split_f32x4(fmaf(acc[i][j][4 * q + 0], kWInv, bv[j]), fmaf(acc[i][j][4 * q + 1], kWInv, bv[j]),
                                                  fmaf(acc[i][j][4 * q + 2], kWInv, bv[j]), fmaf(acc[i][j][4 * q + 3], kWInv, bv[j]), amax16);
                    *lds_at(i, j, q) = __builtin_bit_cast(f16x4, sp.hi);
                    lo_pk[jj][i][q] = sp.lo;
                }
#pragma unroll
        for (int plane = 0; plane < (X2 ? 1 : 2); ++plane) {
            if (plane == 1) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) *lds_at(i, 2 * jh + jj, q) = __builtin_bit_cast(f16x4, lo_pk[jj][i][q]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            half_t* dst = dst_base[plane] + (((size_t)seq * nh + head) * g.hd) * g.Spad + key0;
#pragma unroll
            for (int t = 8 * jh; t < 8 * jh + 8; ++t) {
                const int row = t * 8 + (lane >> 3), c = lane & 7;
                const f16x8 v8 = *reinterpret_cast<const f16x8*>(wlds + row * 64 + ((c ^ (row & 7)) << 3));
                store_stream(reinterpret_cast<f16x8*>(dst + (size_t)row * g.Spad + c * 8), v8);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
    const bool overflow = split_range_exceeded(amax16);
    if (overflow) atomicOr(g.range_flag, 1);
    return true;
}

// Q / K planes of a wave's 64-token x 128-column tile through LDS, for accumulators of the TRANSPOSED product (W fragments as the
// first MFMA operand): registers 4q..4q+3 of a lane are then 4 consecutive COLUMNS of one token, so a plane goes into the wave's
// 16 KB of LDS with 8-byte writes (16-byte units XOR-swizzled by the row: conflict-free both ways) and out again as 16 bytes
// per lane = 256 contiguous bytes per token row: 16 + 16 store instructions per wave instead of 256 two-byte ones.
// Same arithmetic as gemm_h_epilogue (bias, Q pre-scaled by qscale, unscaled hi/lo split).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x4_e __attribute__((ext_vector_type(4)));
typedef int i32x2_e __attribute__((ext_vector_type(2)));
// bf8 (e5m2) images of the four fp16 values in two dwords: their top bytes after rounding to nearest (= bf8_of_f16x4 below)
__device__ __forceinline__ int bf8_of_f16x4_e(int d0, int d1) {
    return (int)__builtin_amdgcn_perm((unsigned)(d1 + 0x00800080), (unsigned)(d0 + 0x00800080), 0x07050301u);
}
template <bool K8IMG = false>     // K8IMG: the kernel variant that can write bf8 K images (attn_mx = 1); compiled out of the default one
__device__ __forceinline__ void qk_staged_store(const GemmHArgs& g, f32x16 (&acc)[2][4], int mw0, int nw0, half_t* wlds, int l31,
                                                int hi, int lane) {
    const int part = nw0 / g.d, nn0 = nw0 - part * g.d;
    half_t* dst_base[2] = {part == 0 ? g.Chi : g.Khi, part == 0 ? g.Clo : g.Klo};
    const float qs = part == 0 ? g.qscale : 1.0f;
    // The values once: bias, Q scale, hi / lo split four at a time into packed fp16 pairs.  In two halves of 32 token rows (i): the
    // hi pairs go straight into the wave's LDS tile, the lo pairs of the half (32 integer registers) wait for its second plane
    // pass - no more registers than the accumulators they replace.  Rows past M hold whatever the padding held: they stay out
    // of the range check.
    unsigned amax16 = 0;
    const bool k8 = K8IMG && part == 1 && g.K8h != nullptr;     // K tile in F16MX: plane 1 is the two bf8 images instead of fp16 K_lo
    const bool q8 = K8IMG && part == 0 && g.Q8l != nullptr;     // Q tile: the bf8 image of Q_lo instead of the fp16 plane
    auto lds_at = [&](int i, int j, int q) {
        const int row = i * 32 + l31;
        const int c = (8 * j + 2 * q + hi) ^ ((row & 15) << 1);       // 8-byte chunk of the row, swizzled in 16-byte units
        return reinterpret_cast<f16x4*>(wlds + row * 128 + c * 4);
    };
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        i32x2_s lo_pk[4][4];      // [j][q]
        unsigned am = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 bv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = *reinterpret_cast<const f32x4*>(g.bias + nw0 + j * 32 + 8 * q + 4 * hi);
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(bv[q]));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaf(acc[i][j][4 * q + e], kWInv, bv[q][e]);
                    if (part == 0) v[e] *= qs;
                }
                const Split4 sp = split_f32x4(v[0], v[1], v[2], v[3], am);
                *lds_at(i, j, q) = __builtin_bit_cast(f16x4, sp.hi);
                lo_pk[j][q] = sp.lo;
            }
        }
        if (mw0 + i * 32 + l31 < g.M) {
            const u16x2_s m = __builtin_elementwise_max(__builtin_bit_cast(u16x2_s, amax16), __builtin_bit_cast(u16x2_s, am));
            amax16 = __builtin_bit_cast(unsigned, m);
        }
#pragma unroll
        for (int plane = 0; plane < 2; ++plane) {
            if (plane == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) *lds_at(i, j, q) = __builtin_bit_cast(f16x4, lo_pk[j][q]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            half_t* dst = dst_base[plane] + (size_t)mw0 * g.d + nn0;
            // K tile with bf8 images wanted (attn_mx = 1): they are made from the rows on their way out - 8 bytes per lane next to
            // the 16 of the fp16 plane; the fp16 K_lo plane itself is not written then
            const bool img = k8 || (q8 && plane == 1);
            unsigned char* dst8 = img ? (k8 ? (plane == 0 ? g.K8h : g.K8l) : g.Q8l) + (size_t)mw0 * g.d + nn0 : nullptr;
#pragma unroll
            for (int t = 8 * i; t < 8 * i + 8; ++t) {
                const int row = t * 4 + (lane >> 4), u = lane & 15;
                const f16x8 v8 = *reinterpret_cast<const f16x8*>(wlds + row * 128 + ((u ^ (row & 15)) << 3));
                if (mw0 + row < g.M) {
                    if (!((k8 || q8) && plane == 1)) store_stream(reinterpret_cast<f16x8*>(dst + (size_t)row * g.d + u * 8), v8);
                    if (img) {
                        const i32x4_e dw = __builtin_bit_cast(i32x4_e, v8);
                        i32x2_e b8;
                        b8[0] = bf8_of_f16x4_e(dw[0], dw[1]);
                        b8[1] = bf8_of_f16x4_e(dw[2], dw[3]);
                        store_stream(reinterpret_cast<i32x2_e*>(dst8 + (size_t)row * g.d + u * 8), b8);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
    const bool overflow = split_range_exceeded(amax16);
    if (overflow) atomicOr(g.range_flag, 1);
}

// linear1 (bias + ReLU, ONE fp16 plane: the F16X2 / F16MX consumer takes A_hi only) the same way: the wave's 64-token x 128-column tile
// of the TRANSPOSED product goes into its 16 KB of LDS with 8-byte writes and leaves as 16 bytes per lane - a wave-instruction then
// writes 4 token rows x 4 panels x 64 bytes, i.e. eight whole 128-byte lines of the blocked plane (two consecutive rows of a panel
// share a line), where the element-wise epilogue issued 128 two-byte store instructions per lane that each touched two half lines.
// Same arithmetic as gemm_h_epilogue_impl<.., EPI_BIAS_RELU, OUT_SPLIT, X2 = true>: the same bits.
__device__ __forceinline__ void h1_staged_store(const GemmHArgs& g, f32x16 (&acc)[2][4], int mw0, int nw0, half_t* wlds, int l31, int hi,
                                                int lane) {
    unsigned amax16 = 0;
    auto lds_at = [&](int i, int j, int q) {
        const int row = i * 32 + l31;
        const int c = (8 * j + 2 * q + hi) ^ ((row & 15) << 1);       // 8-byte chunk of the row, swizzled in 16-byte units
        return reinterpret_cast<f16x4*>(wlds + row * 128 + c * 4);
    };
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        unsigned am = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 bv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = *reinterpret_cast<const f32x4*>(g.bias + nw0 + j * 32 + 8 * q + 4 * hi);
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(bv[q]));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaf(acc[i][j][4 * q + e], kWInv, bv[q][e]);
                    v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                const Split4 sp = split_f32x4(v[0], v[1], v[2], v[3], am);
                *lds_at(i, j, q) = __builtin_bit_cast(f16x4, sp.hi);
            }
        }
        if (mw0 + i * 32 + l31 < g.M) {      // rows past M hold whatever the padding held: they stay out of the range check
            const u16x2_s m = __builtin_elementwise_max(__builtin_bit_cast(u16x2_s, amax16), __builtin_bit_cast(u16x2_s, am));
            amax16 = __builtin_bit_cast(unsigned, m);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // lane = (token row 4 t + lane / 16, 16-byte unit u = lane % 16 of its 256 bytes): panel nw0 / 32 + u / 4, chunk u % 4 of the panel
    // row, swizzled as blk_index does.  The tile starts at row 0 or 64 of its 128-row block, so bits 2-3 of panel row 4 t + lane / 16
    // are t & 3: the stored chunk is (u % 4) ^ (t & 3) - four lane offsets, and 4 rows = 256 bytes per step of t as an immediate
    const int row0 = lane >> 4, u = lane & 15;
    half_t* const dst = g.Chi + ((size_t)(mw0 >> 7) * (g.N >> 5) + (nw0 >> 5) + (u >> 2)) * 4096 + (size_t)((mw0 & 127) + row0) * 32;
    const int cb[4] = {(u & 3) << 3, ((u & 3) ^ 1) << 3, ((u & 3) ^ 2) << 3, ((u & 3) ^ 3) << 3};
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int row = t * 4 + row0;
        const f16x8 v8 = *reinterpret_cast<const f16x8*>(wlds + row * 128 + ((u ^ (row & 15)) << 3));
        if (mw0 + row < g.M) store_stream(reinterpret_cast<f16x8*>(dst + t * 128 + cb[t & 3]), v8);
    }
    if (split_range_exceeded(amax16)) atomicOr(g.range_flag, 1);
}

// ---------------------------------------------------------------------------------------------------------------
// 256x256 LDS-DMA variant (N a multiple of 256: in_proj 1536, linear1 1024): 8 waves (4 along M x 2 along N), wave tile
// 64 x 128 - possible since the product needs ONE accumulator set (128 VGPRs).  A third fewer operand bytes per FLOP
// through L2 -> LDS and a quarter fewer fragment reads per MFMA than the 256x128 tile.  64 KB stages, 2-stage ring
// (the next tile's 8 DMA instructions are issued right after the barrier):
// one K-tile of look-ahead is 48 MFMAs per wave, the same cover time as two tiles of the 256x128 kernel.
constexpr int DMA3_STAGE = 8 * DMA_PLANE;                 // Ahi(2 images), Alo(2), Whi(2), Wlo(2)
constexpr size_t DMA3_LDS_BYTES = size_t(2) * DMA3_STAGE * sizeof(half_t);

template <int EPI, int OUT, bool X2 = false>
__global__ __launch_bounds__(512, 2) void gemm_f16x3_dma256x256_kernel(GemmHArgs g, int ntm, int ntn, int burst, int stage_vt) {
    constexpr int WM = 2, WN = 4, BM = 256, BN = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wid >> 1, wc = wid & 1;                 // wr 0..3 (64 rows each), wc 0..1 (128 columns each)
    // XCD-contiguous tile ranges, N fastest: the N-tiles of an M-tile run together and share its A panels in L2
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const int tm = swz / ntn, tn = swz - tm * ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = g.K / GEMMH_BK;
    const int nrb = (g.M + 127) / 128;
    const int rb0 = 2 * tm, rb1 = (2 * tm + 1 < nrb) ? 2 * tm + 1 : nrb - 1;
    // copies: wave-uniform base (pinned in scalar registers, through an integer) + ONE 32-bit lane offset, tid * 16 bytes - eight
    // per-thread 64-bit pointers cost 16 registers, a 64-bit vector add per copy and a v_readfirstlane pair for its LDS destination
    const half_t* src[8];
    src[0] = g.Ahi + (size_t)rb0 * nk * 4096;
    src[1] = g.Ahi + (size_t)rb1 * nk * 4096;
    src[2] = g.Alo + (size_t)rb0 * nk * 4096;
    src[3] = g.Alo + (size_t)rb1 * nk * 4096;
    src[4] = g.Whi + (size_t)(2 * tn) * nk * 4096;
    src[5] = g.Whi + (size_t)(2 * tn + 1) * nk * 4096;
    src[6] = g.Wlo + (size_t)(2 * tn) * nk * 4096;
    src[7] = g.Wlo + (size_t)(2 * tn + 1) * nk * 4096;
    unsigned lane_off = (unsigned)tid * 16u;      // (re-pinned once per K tile: hoisted out of the loop as a 64-bit pair it defeats the scalar-base form)
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);
    auto issue_one = [&](int kt, int i, int stage) {
        half_t* st = lds + stage * DMA3_STAGE + wid_s * 512;
        if ((i == 2 || i == 3) && X2) return;   // F16X2: the A lo images stay out of LDS
        const unsigned long long u = pin_uniform(reinterpret_cast<unsigned long long>(src[i] + (size_t)kt * 4096));
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) char*>(u) + lane_off,
                                         (__attribute__((address_space(3))) void*)(st + i * 4096), 16, 0, 0);
    };
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int offA[WM][2], offW[WN][2];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int row = wr * 64 + i * 32 + l31, r = row & 127;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            offA[i][ks] = (row >> 7) * DMA_PLANE + r * 32 + (((ks * 2 + hi) ^ ((r >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int row = wc * 128 + j * 32 + l31, r = row & 127;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            offW[j][ks] = (row >> 7) * DMA_PLANE + r * 32 + (((ks * 2 + hi) ^ ((r >> 2) & 3)) * 8);
    }
    // block-uniform: a Q or K tile of the QKV GEMM (columns below 2 d) is computed TRANSPOSED - W fragments as the first MFMA
    // operand, so that a lane holds four consecutive columns of a token - and leaves through LDS in full rows (qk_staged_store;
    // same values: a product does not care which operand it came in as)
    if (OUT == OUT_QKV && (stage_vt & 2) && n0 < 2 * g.d && g.d % 128 == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) issue_one(0, i, 0);
        // one K tile with its ring stage as a constant (two tiles per trip below): every fragment read is lane offset + immediate
        auto ktile = [&](const int kt, auto stg_c) {
            constexpr int STG = decltype(stg_c)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+v"(lane_off));
            if (kt + 1 < nk) {
#pragma unroll
                for (int i = 0; i < 8; ++i) issue_one(kt + 1, i, 1 - STG);
            }
            const half_t* st = lds + STG * DMA3_STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 ah[WM], al[WM], wh[WN], wl[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8*>(st + offA[i][ks]);
                    if (!X2) al[i] = *reinterpret_cast<const f16x8*>(st + 2 * DMA_PLANE + offA[i][ks]);
                }
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    wh[j] = *reinterpret_cast<const f16x8*>(st + 4 * DMA_PLANE + offW[j][ks]);
                    wl[j] = *reinterpret_cast<const f16x8*>(st + 6 * DMA_PLANE + offW[j][ks]);
                }
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], ah[i], acc[i][j], 0, 0, 0);
                if (!X2)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], al[i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        {
            int kt = 0;
            for (; kt + 1 < nk; kt += 2) {
                ktile(kt, std::integral_constant<int, 0>{});
                ktile(kt + 1, std::integral_constant<int, 1>{});
            }
            if (kt < nk) ktile(kt, std::integral_constant<int, 0>{});
        }
        __syncthreads();      // everybody is done with the operand rings (all DMAs landed: vmcnt(0) in the last K-tile)
        qk_staged_store(g, acc, m0 + wr * 64, n0 + wc * 128, lds + wid * 8192, l31, hi, lane);
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_one(0, i, 0);
    auto ktile2 = [&](const int kt, auto stg_c) {
        constexpr int STG = decltype(stg_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(lane_off));
        const bool more = kt + 1 < nk && !burst;
        if (burst && kt + 1 < nk) {
#pragma unroll
            for (int i = 0; i < 8; ++i) issue_one(kt + 1, i, 1 - STG);
        }
        const half_t* st = lds + STG * DMA3_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah[WM], al[WM], wh[WN], wl[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(st + offA[i][ks]);
                if (!X2) al[i] = *reinterpret_cast<const f16x8*>(st + 2 * DMA_PLANE + offA[i][ks]);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                wh[j] = *reinterpret_cast<const f16x8*>(st + 4 * DMA_PLANE + offW[j][ks]);
                wl[j] = *reinterpret_cast<const f16x8*>(st + 6 * DMA_PLANE + offW[j][ks]);
            }
            // burst == 0 (diagnostics, gemm_abl bit 3): the 8 DMA instructions of the next K-tile go out behind the MFMA
            // groups instead of right after the barrier - measured 1 % slower here, unlike in the attention kernel
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc[i][j], 0, 0, 0);
            if (more) {
                issue_one(kt + 1, 4 * ks + 0, 1 - STG);
                issue_one(kt + 1, 4 * ks + 1, 1 - STG);
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl[j], acc[i][j], 0, 0, 0);
            if (more) {
                issue_one(kt + 1, 4 * ks + 2, 1 - STG);
                issue_one(kt + 1, 4 * ks + 3, 1 - STG);
            }
            if (!X2)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            ktile2(kt, std::integral_constant<int, 0>{});
            ktile2(kt + 1, std::integral_constant<int, 1>{});
        }
        if (kt < nk) ktile2(kt, std::integral_constant<int, 0>{});
    }
    if (OUT == OUT_QKV) {
        // a V tile (block-uniform: n0 is a multiple of 256 and d_model of 128): V^T goes out through LDS in full rows
        if (g.vt_direct && g.hd == 128 && n0 >= 2 * g.d && (stage_vt & 1)) {
            __syncthreads();      // everybody is done with the operand rings (all DMAs landed: vmcnt(0) in the last K-tile)
            if (vt_staged_store<X2>(g, acc, m0 + wr * 64, n0 + wc * 128, lds + wid * 8192, l31, hi, lane)) return;
        }
    }
    gemm_h_epilogue<WM, WN, EPI, OUT, X2>(g, acc, m0, n0, wr, wc, l31, hi, BM, BN);
}

template <int EPI, int OUT, bool X2 = false>
inline hipError_t launch_gemm_h_dma256x256(const GemmHArgs& g, hipStream_t st) {
    const int ntm = (g.M + 255) / 256, ntn = g.N / 256;
    static DevSeen attr_seen;
    if (auto once_ = first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_dma256x256_kernel<EPI, OUT, X2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)DMA3_LDS_BYTES);
    }
    const int vs = tune().vt_stage;     // 0 / 1: V^T and Q / K through LDS, 2: neither, 3: V^T only
    hipLaunchKernelGGL((gemm_f16x3_dma256x256_kernel<EPI, OUT, X2>), dim3(ntm * ntn), dim3(512), DMA3_LDS_BYTES, st, g, ntm, ntn, (gemm_abl_bits() & 8) ? 0 : 1,
                       vs == 2 ? 0 : (vs == 3 ? 1 : 3));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// JMID_PREC_F16MX: the F16X2 product  A_hi . (W_hi + W_lo)  with the correction term on the block-scaled fp8 matrix path:
//     acc += A_hi . W_hi                 four v_mfma_f32_32x32x16_f16 per k64 and output tile, as before
//     acc += bf8(A_hi) . bf8(W_lo)       ONE v_mfma_f32_32x32x64_f8f6f4 per k64 and output tile (the time of two fp16 steps)
// 1.5 instead of 2 MFMA passes per product.  The term is 2^-11 of the product, so 2-3 significand bits are plenty:
//   * bf8 (e5m2) has the exponent field of fp16: bf8(A_hi) is the top byte of every fp16 value, rounded to nearest by adding
//     0x80 below it - built from the A_hi fragments the wave already holds (one v_add + half a v_perm per dword), no extra
//     operand plane and no extra LDS traffic;
//   * bf8(W_lo) is made at weight-load time (w8_image_kernel), already in the order the instruction wants it, half the bytes of
//     the fp16 W_lo image it replaces in L2 -> LDS.  W_lo of the 2^8-scaled weights lies in bf8's range (subnormals to 2^-16).
//   * NO block scales: with literal zero scale operands the compiler emits the plain v_mfma_f32_32x32x64_f8f6f4.  The scaled
//     form is a PAIR (v_mfma_ld_scale_b32 + MFMA), and a wave of another kernel on the same SIMD (out_ddim_kernel next to the
//     166-VGPR 64-row GEMM + LayerNorm kernel) made such pairs compute with a wrong scale now and then: tile-wide 1-ulp
//     differences from run to run (tools/concurrency_probe9.hip, docs/NOTEBOOK.md section 3).
// The instruction sums over its 64 k in any order as long as A and W agree: byte p of lane (row, h) is k = 16 (p / 8) + 8 h + p % 8
// of the k64 block for both, i.e. exactly the fp16 fragments' assignment.
// W8 layout: [K / 64][N / 32][2 pieces][64 lanes][16 bytes]: the 32-column blocks of a workgroup tile are contiguous per k64
// block, a wave reads its block with two linear ds_read_b128.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// fp32 [N, K] -> bf8 (e5m2) image of W_lo = W * scale - fp16(W * scale): the top byte of fp16(W_lo) after rounding to nearest
static __global__ void w8_image_kernel(const float* W, unsigned char* out, int N, int K, float scale) {
    const size_t n = (size_t)N * K;
    const int nb32 = N / 32;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / K), k = (int)(i % K);
        float v = W[i] * scale;
        asm("" : "+v"(v));
        const half_t h = (half_t)v;
        const half_t lo = (half_t)(v - (float)h);
        const int kb = k >> 6, kk = k & 63, ksg = kk >> 4, hh = (kk >> 3) & 1, e = kk & 7;
        const int p = ksg * 8 + e, lane = (r & 31) + 32 * hh;
        const size_t o = ((((size_t)kb * nb32 + (r >> 5)) * 2 + (p >> 4)) * 64 + lane) * 16 + (p & 15);
        out[o] = (unsigned char)((__builtin_bit_cast(unsigned short, lo) + 0x80u) >> 8);
    }
}

// bf8 (e5m2) images of the four fp16 values in two dwords: their top bytes after rounding to nearest
__device__ __forceinline__ int bf8_of_f16x4(int d0, int d1) {
    return (int)__builtin_amdgcn_perm((unsigned)(d1 + 0x00800080), (unsigned)(d0 + 0x00800080), 0x07050301u);
}

// ONE kernel for every tile shape of the mode (the shapes of the F16X2 kernels above, chosen by the same rules), so that an
// episode's result does not depend on which shape its batch got: every accumulator sees, per k64 block, the four fp16 steps in
// k order and then the fp8 instruction - bit-identical across shapes.
//   WR x WC waves, wave tile (32 WM) x (32 WN):  64 x 64 (2x2 waves of 32 x 32, small M),  128 x 128 (2x2 of 64 x 64),
//   256 x 128 (4x2 of 64 x 64; the ConcatSquash epilogue fits next to 64 accumulators),  256 x 256 (4x2 of 64 x 128).
// LDS: NS stages of [A_hi tile][W_hi tile] of one k32 step (without the lo images there is room for a second / third tile of
// look-ahead), then two buffers for the fp8 image of a k64 block, which travels with the ODD k32 tile of its block (the one after
// which it is used).  The DMA instructions younger than tile kt's at the top of tile kt are a compile-time constant per tile parity.
template <int WR, int WC, int WM, int WN, int NS>
struct MxCfg {
    static constexpr int NT = 64 * WR * WC, BM = 32 * WM * WR, BN = 32 * WN * WC;
    static constexpr int A_HALFS = BM * 32, W_HALFS = BN * 32, STAGE = A_HALFS + W_HALFS;     // halfs per k32 stage
    static constexpr int W8_BYTES = BN * 64;                                                  // fp8 image of a k64 block
    static constexpr int PIECE = NT * 8;                                                      // halfs per DMA instruction
    static constexpr int NA = A_HALFS / PIECE, NW = W_HALFS / PIECE, N8 = W8_BYTES / (NT * 16);
    static constexpr size_t W8_OFF = size_t(NS) * STAGE * sizeof(half_t);
    static constexpr size_t RING_BYTES = W8_OFF + 2 * W8_BYTES;
    static constexpr size_t VT_BYTES = (WM == 2 && WN == 4) ? size_t(WR * WC) * 16384 : 0;     // vt / qk staged stores
    static constexpr size_t LDS_BYTES = RING_BYTES > VT_BYTES ? RING_BYTES : VT_BYTES;
    static_assert(NA >= 1 && NW >= 1 && N8 >= 1 && A_HALFS % PIECE == 0 && W_HALFS % PIECE == 0, "tile / workgroup mismatch");
};
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int EPI, int OUT, int WR, int WC, int WM, int WN, int NS, bool K8IMG = false>
__global__ __launch_bounds__(64 * WR * WC, (WR * WC == 4 && WM == 2 && NS > 2) ? 1 : 2) void gemm_mx_kernel(GemmHArgs g, int ntm, int ntn, int stage_vt) {
    using C = MxCfg<WR, WC, WM, WN, NS>;
    constexpr int BM = C::BM, BN = C::BN, L = NS - 1;          // L tiles of look-ahead
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wid / WC, wc = wid % WC;
    // XCD-contiguous tile ranges, N fastest: the N-tiles of an M-tile run together and share its A panels in L2
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    // (experiment knob "gemm_pn": the tile sequence cut into column groups of gw N-tiles, each walked M-major - gemm_small.hpp's order)
    const int gw = (stage_vt >> 8) ? (stage_vt >> 8) : ntn;
    const int per_g = ntm * gw, cg = swz / per_g, rem_g = swz - cg * per_g;
    const int tm = rem_g / gw, tn = cg * gw + (rem_g - tm * gw);
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = g.K / GEMMH_BK;                     // even: K is a multiple of 64
    const int nrb = (g.M + 127) / 128;
    // DMA sources: piece p of the A (W) tile is C::PIECE halfs of the blocked panel image(s) of this tile's rows
    // wave-uniform bases (scalar registers) + ONE 32-bit lane offset for every copy: a thread's 16 bytes of a piece are at tid * 16 in
    // each of them (per-thread 64-bit pointers cost a 64-bit vector add and a v_readfirstlane pair per copy: 26 of the K loop's ~100
    // vector instructions per k64 block)
    const half_t* srcA[C::NA];
    const half_t* srcW[C::NW];
#pragma unroll
    for (int p = 0; p < C::NA; ++p) {
        const int h0 = (m0 & 127) * 32 + p * C::PIECE;                   // halfs from the start of the tile's first panel image
        int rb = m0 / 128 + h0 / 4096;
        rb = rb < nrb ? rb : nrb - 1;
        srcA[p] = g.Ahi + (size_t)rb * nk * 4096 + h0 % 4096;
    }
#pragma unroll
    for (int p = 0; p < C::NW; ++p) {
        const int h0 = (n0 & 127) * 32 + p * C::PIECE;
        srcW[p] = g.Whi + (size_t)(n0 / 128 + h0 / 4096) * nk * 4096 + h0 % 4096;
    }
    const unsigned char* src8 = g.W8 + (size_t)(n0 / 32) * 2048;
    unsigned lane_off = (unsigned)tid * 16u;      // (re-pinned once per K tile in top(): zero-extended and hoisted out of the loop as a 64-bit pair it defeats the scalar-base form of the copies)
    const size_t w8_kstride = (size_t)(g.N / 32) * 2048;
    unsigned char* lds8 = lds_raw + C::W8_OFF;
    // (the uniform part is pinned in scalar registers - through an integer, a pointer that passes an asm operand comes back generic:
    //  left alone, hipcc hoists base + lane offset out of the K loop as a 64-bit vector and adds the tile stride to THAT per copy)
    auto dma16 = [&](const void* s, void* d) {
        const unsigned long long u = pin_uniform(reinterpret_cast<unsigned long long>(s));
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) char*>(u) + lane_off,
                                         (__attribute__((address_space(3))) void*)d, 16, 0, 0);
    };
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);      // (the LDS destination of a copy is a scalar: M0)
    auto issue = [&](int kt, int stg) {                // NA + NW wave-instructions, + N8 for an odd tile
        half_t* st = lds + stg * C::STAGE + wid_s * 512;
#pragma unroll
        for (int p = 0; p < C::NA; ++p) dma16(srcA[p] + (size_t)kt * 4096, st + p * C::PIECE);
#pragma unroll
        for (int p = 0; p < C::NW; ++p) dma16(srcW[p] + (size_t)kt * 4096, st + C::A_HALFS + p * C::PIECE);
        if (kt & 1) {                                  // the fp8 image of k64 block kt / 2, into buffer (kt / 2) & 1 (last read L + 1 tiles ago)
            const unsigned char* s8 = src8 + (size_t)(kt >> 1) * w8_kstride;
            unsigned char* d8 = lds8 + ((kt >> 1) & 1) * C::W8_BYTES + wid_s * 1024;
#pragma unroll
            for (int q = 0; q < C::N8; ++q) dma16(s8 + q * C::NT * 16, d8 + q * C::NT * 16);
        }
    };
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int offA[WM][2], offW[WN][2];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int row = wr * WM * 32 + i * 32 + l31, r = row & 127;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            offA[i][ks] = (row >> 7) * DMA_PLANE + r * 32 + (((ks * 2 + hi) ^ ((r >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int row = wc * WN * 32 + j * 32 + l31, r = row & 127;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            offW[j][ks] = C::A_HALFS + (row >> 7) * DMA_PLANE + r * 32 + (((ks * 2 + hi) ^ ((r >> 2) & 3)) * 8);
    }
    i32x8 a8[WM];
    // one k32 tile: the fp16 product, and the bf8 image of its A fragments into half HALF of the fp8 operand.
    // SWAP: the W fragments go first, the accumulators hold the transposed tile (qk_staged_store)
    auto tile = [&](const half_t* st, auto half_c, auto swap_c) {
        constexpr int HALF = decltype(half_c)::value;
        constexpr bool SWAP = decltype(swap_c)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah[WM], wh[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) ah[i] = *reinterpret_cast<const f16x8*>(st + offA[i][ks]);
#pragma unroll
            for (int j = 0; j < WN; ++j) wh[j] = *reinterpret_cast<const f16x8*>(st + offW[j][ks]);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], ah[i], acc[i][j], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const i32x4 d = __builtin_bit_cast(i32x4, ah[i]);
                a8[i][HALF * 4 + ks * 2 + 0] = bf8_of_f16x4(d[0], d[1]);
                a8[i][HALF * 4 + ks * 2 + 1] = bf8_of_f16x4(d[2], d[3]);
            }
        }
    };
    // tile kt has landed for everybody; the stage of tile kt - 1 is free again.  Younger than tile kt's DMA group at this point:
    // the groups of tiles kt + 1 .. kt + L - 1, of which the odd ones carry an fp8 image
    auto top = [&](int kt, int stg_next, auto par_c) {
        constexpr int P = decltype(par_c)::value;
        constexpr int n_odd = P == 0 ? L / 2 : (L - 1) / 2;
        if (kt + L - 1 < nk) wait_vmcnt<(L - 1) * (C::NA + C::NW) + n_odd * C::N8>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(lane_off));
        if (kt + L < nk) issue(kt + L, stg_next);
    };
    auto kloop = [&](auto swap_c) {
        constexpr bool SWAP = decltype(swap_c)::value;
#pragma unroll
        for (int t = 0; t < L; ++t)
            if (t < nk) issue(t, t);
        int stg = 0;                                       // stage of tile kt
        for (int kt = 0; kt < nk; kt += 2) {
            const int s1 = stg + 1 == NS ? 0 : stg + 1, s2 = s1 + 1 == NS ? 0 : s1 + 1;
            top(kt, stg == 0 ? NS - 1 : stg - 1, std::integral_constant<int, 0>{});      // (stg + L) % NS
            tile(lds + stg * C::STAGE, std::integral_constant<int, 0>{}, swap_c);
            __builtin_amdgcn_sched_barrier(0);
            top(kt + 1, stg, std::integral_constant<int, 1>{});                          // (s1 + L) % NS
            tile(lds + s1 * C::STAGE, std::integral_constant<int, 1>{}, swap_c);
            {
                const unsigned char* wb = lds8 + ((kt >> 1) & 1) * C::W8_BYTES;
                i32x8 w8[WN];
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const unsigned char* p = wb + (size_t)((wc * WN + j) * 2) * 1024 + lane * 16;
                    const i32x4 lo = *reinterpret_cast<const i32x4*>(p);
                    const i32x4 up = *reinterpret_cast<const i32x4*>(p + 1024);
                    w8[j] = i32x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
                }
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)      // both operands bf8 (format selector 1); literal zero scale operands select the UNSCALED instruction
                        acc[i][j] = SWAP ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8[j], a8[i], acc[i][j], 1, 1, 0, 0, 0, 0)
                                         : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], w8[j], acc[i][j], 1, 1, 0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            stg = s2;
        }
    };
    if constexpr (OUT == OUT_QKV && WM == 2 && WN == 4) {
        // block-uniform: a Q or K tile (columns below 2 d) is computed transposed and leaves through LDS in full rows
        // (with bf8 K images always: the generic epilogue of this shape is compiled without the byte stores)
        if (((stage_vt & 2) || (K8IMG && g.K8h && (n0 >= g.d || g.Q8l))) && n0 < 2 * g.d && g.d % 128 == 0) {
            kloop(std::true_type{});
            __syncthreads();      // everybody is done with the operand rings (all DMAs landed: vmcnt(0) in the last K-tile)
            qk_staged_store<K8IMG>(g, acc, m0 + wr * 64, n0 + wc * 128, lds + wid * 8192, l31, hi, lane);
            return;
        }
    }
    if constexpr ((EPI == EPI_CSL && (OUT == OUT_F32 || OUT == OUT_SPLIT)) || (EPI == EPI_BIAS_RELU && OUT == OUT_SPLIT)) {
        // the tail GEMMs (ConcatSquash): transposed product, row-wise epilogue.  (linear1 the same way - csl_swap = 3 - is 1 %
        // slower per call: its column-wise 2-byte stores are cheaper than row-wise 8-byte ones, and it has no per-row work to save)
        if (stage_vt & (EPI == EPI_CSL ? 4 : 8)) {
            kloop(std::true_type{});
            csl_swapped_epilogue<WM, WN, EPI, OUT, true>(g, acc, m0 + wr * WM * 32, n0 + wc * WN * 32, l31, hi);
            return;
        }
    }
    if constexpr (EPI == EPI_BIAS_RELU && OUT == OUT_SPLIT && WM == 2 && WN == 4) {
        // linear1 in the 64 x 128 wave tile: transposed product, the tile out through LDS in whole lines (block-uniform)
        if (stage_vt & 16) {
            kloop(std::true_type{});
            __syncthreads();      // everybody is done with the operand rings
            h1_staged_store(g, acc, m0 + wr * 64, n0 + wc * 128, lds + wid * 8192, l31, hi, lane);
            return;
        }
    }
    kloop(std::false_type{});
    if constexpr (OUT == OUT_QKV && WM == 2 && WN == 4) {
        if (g.vt_direct && g.hd == 128 && n0 >= 2 * g.d && (stage_vt & 1)) {
            __syncthreads();
            if (vt_staged_store<true>(g, acc, m0 + wr * 64, n0 + wc * 128, lds + wid * 8192, l31, hi, lane)) return;
        }
    }
    gemm_h_epilogue<WM, WN, EPI, OUT, true, OUT == OUT_QKV && !(WM == 2 && WN == 4)>(g, acc, m0, n0, wr, wc, l31, hi, BM, BN);
}

template <int EPI, int OUT, int WR, int WC, int WM, int WN, int NS, bool K8IMG = false>
inline hipError_t launch_gemm_mx_cfg(const GemmHArgs& g, hipStream_t st) {
    using C = MxCfg<WR, WC, WM, WN, NS>;
    if constexpr (OUT == OUT_QKV && WM == 2 && WN == 4 && !K8IMG)      // bf8 K images wanted (attn_mx = 1): the variant that can write them
        if (g.K8h) return launch_gemm_mx_cfg<EPI, OUT, WR, WC, WM, WN, NS, true>(g, st);
    const int ntm = (g.M + C::BM - 1) / C::BM, ntn = g.N / C::BN;
    static DevSeen attr_seen;
    if (auto once_ = first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mx_kernel<EPI, OUT, WR, WC, WM, WN, NS, K8IMG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
    }
    const int vs = tune().vt_stage;     // 0 / 1: V^T and Q / K through LDS, 2: neither, 3: V^T only; bit 2: row-wise ConcatSquash epilogue
    hipLaunchKernelGGL((gemm_mx_kernel<EPI, OUT, WR, WC, WM, WN, NS, K8IMG>), dim3(ntm * ntn), dim3(C::NT), C::LDS_BYTES, st, g, ntm, ntn,
                       (vs == 2 ? 0 : (vs == 3 ? 1 : 3)) | (tune().csl_swap == 2 ? 0 : 4) | (tune().csl_swap == 3 ? 8 : 0) | (tune().h1_stage == 2 ? 0 : 16) |
                           ((tune().gemm_pn > 1 && ntn % tune().gemm_pn == 0 ? ntn / tune().gemm_pn : 0) << 8));
    return hipGetLastError();
}
template <int EPI, int OUT> inline hipError_t launch_gemm_mx_64(const GemmHArgs& g, hipStream_t st) { return launch_gemm_mx_cfg<EPI, OUT, 2, 2, 1, 1, 4>(g, st); }
template <int EPI, int OUT> inline hipError_t launch_gemm_mx_128(const GemmHArgs& g, hipStream_t st) { return launch_gemm_mx_cfg<EPI, OUT, 2, 2, 2, 2, 4>(g, st); }
template <int EPI, int OUT> inline hipError_t launch_gemm_mx_256x128(const GemmHArgs& g, hipStream_t st) { return launch_gemm_mx_cfg<EPI, OUT, 4, 2, 2, 2, 3>(g, st); }
// 256 x 256: a TWO-stage ring - the K loop's body is two tiles, so every tile's stage is a constant and its fragment reads are lane offset +
// immediate (29 vector address instructions per k64 block less than with three stages, whose extra tile of look-ahead measured neutral in
// round 2); a 51-episode call 113.4 -> 112.8 ms, 256 episodes 573 -> 560 ms.  Diagnostics variant 8: the three-stage ring, for the A/B.
template <int EPI, int OUT> inline hipError_t launch_gemm_mx_256x256(const GemmHArgs& g, hipStream_t st) { return launch_gemm_mx_cfg<EPI, OUT, 4, 2, 2, 4, 2>(g, st); }
template <int EPI, int OUT> inline hipError_t launch_gemm_mx_256x256_ns3(const GemmHArgs& g, hipStream_t st) { return launch_gemm_mx_cfg<EPI, OUT, 4, 2, 2, 4, 3>(g, st); }
// 128 x 256 with four waves (the 256 x 256 shape's wave tile), a two-stage ring: 80 KB of LDS and <= 256 registers, so TWO workgroups
// share a CU - one's epilogue (stores through LDS, no MFMA) under the other's K loop
template <int EPI, int OUT> inline hipError_t launch_gemm_mx_128x256(const GemmHArgs& g, hipStream_t st) { return launch_gemm_mx_cfg<EPI, OUT, 2, 2, 2, 4, 2>(g, st); }

// the F16X2 shape rules (launch_gemm_h_mode below) for the fp8-correction kernels
template <int EPI, int OUT>
inline hipError_t launch_gemm_mx(const GemmHArgs& g, hipStream_t st) {
    const long big = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    const int v = tune().gemm_h_variant;
    // the 256 x 128 shape with the ConcatSquash epilogue into fp32 (concat4: N = 128, never enough tiles for it) spills: not instantiated
    constexpr bool has_256x128 = !(EPI == EPI_CSL && OUT == OUT_F32);
    if (v == 3) return launch_gemm_mx_128<EPI, OUT>(g, st);
    if constexpr (has_256x128)
        if (v == 4) return launch_gemm_mx_256x128<EPI, OUT>(g, st);
    if (v == 5) return launch_gemm_mx_64<EPI, OUT>(g, st);
    if constexpr (EPI != EPI_CSL) {
        if (v == 6 && g.N % 256 == 0) return launch_gemm_mx_256x256<EPI, OUT>(g, st);
#ifdef JMID_DIAGNOSTICS      // measured: 51 episodes in ONE chunk 119.95 -> 117.82 ms, as two chunks in flight (the default plan) 114.62 -> 115.63
        if (v == 7 && g.N % 256 == 0) return launch_gemm_mx_128x256<EPI, OUT>(g, st);
        if (v == 8 && g.N % 256 == 0) return launch_gemm_mx_256x256_ns3<EPI, OUT>(g, st);
#endif
    }
    if (big < 256) return launch_gemm_mx_64<EPI, OUT>(g, st);
    const long nb256 = (long)((g.M + 255) / 256) * ((g.N + 127) / 128);
    auto eff = [](long nb) { return (double)nb / (double)(((nb + 255) / 256) * 256); };
    if constexpr (EPI != EPI_CSL)
        if (g.N % 256 == 0) {
            // also below one workgroup per CU, from 7168 rows: two chunks are in flight, and the larger tile moves a third
            // fewer operand bytes per FLOP (2 x 12 episodes per call 60.6 vs 67.6 ms, 2 x 8: 45.1 vs 47.4, 2 x 6 equal,
            // 2 x 5: 35.8 vs 33.4; tools/single_scene_sweep.py gemm_h_variant=0,6 f16mx E)
            const long nbq = (long)((g.M + 255) / 256) * (g.N / 256);
            if ((nbq >= 256 && 1.2 * eff(nbq) >= eff(nb256)) || (nbq < 256 && g.M >= 7168)) return launch_gemm_mx_256x256<EPI, OUT>(g, st);
        }
    if constexpr (has_256x128)
        if (nb256 >= 256 && 1.2 * eff(nb256) >= eff(big)) return launch_gemm_mx_256x128<EPI, OUT>(g, st);
    return launch_gemm_mx_128<EPI, OUT>(g, st);
}

// ---------------------------------------------------------------------------------------------------------------
// 64x64 LDS-DMA variant for small M (one scene: M = 1200 tokens): the K loop of a small tile is pure latency, so the
// 4-stage ring (three 16 KB K-tiles in flight, 64 KB of LDS, two workgroups per CU) matters more here than anywhere.
// A 64-row half of a 128-row panel image is a contiguous 4 KB piece: one DMA round per plane.
constexpr int DMA64_PLANE = 64 * 32;
constexpr int DMA64_STAGE = 4 * DMA64_PLANE;
constexpr size_t DMA64_LDS_BYTES = size_t(4) * DMA64_STAGE * sizeof(half_t);

template <int EPI, int OUT, bool X2 = false>
__global__ __launch_bounds__(256, 2) void gemm_f16x3_dma64_kernel(GemmHArgs g, int ntm, int ntn) {
    constexpr int WM = 1, WN = 1, BM = 64, BN = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wid >> 1, wc = wid & 1;
    // consecutive workgroups share the A row-panel (N fastest); XCD-contiguous ranges
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const int tm = swz / ntn, tn = swz - tm * ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = g.K / GEMMH_BK;
    const half_t* src[4];
    src[0] = g.Ahi + ((size_t)(tm >> 1) * nk) * 4096 + (tm & 1) * 2048 + tid * 8;
    src[1] = g.Alo + ((size_t)(tm >> 1) * nk) * 4096 + (tm & 1) * 2048 + tid * 8;
    src[2] = g.Whi + ((size_t)(tn >> 1) * nk) * 4096 + (tn & 1) * 2048 + tid * 8;
    src[3] = g.Wlo + ((size_t)(tn >> 1) * nk) * 4096 + (tn & 1) * 2048 + tid * 8;
    auto issue = [&](int kt) {
        half_t* st = lds + (kt & 3) * DMA64_STAGE + wid * 512;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)kt * 4096),
                                             (__attribute__((address_space(3))) void*)(st + i * DMA64_PLANE), 16, 0, 0);
    };
    f32x16 accm[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        accm[0][0][r] = 0.f;
    }
    const int rowA = wr * 32 + l31, rowW = wc * 32 + l31;
    int offA[2], offW[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        offA[ks] = rowA * 32 + (((ks * 2 + hi) ^ ((rowA >> 2) & 3)) * 8);
        offW[ks] = rowW * 32 + (((ks * 2 + hi) ^ ((rowW >> 2) & 3)) * 8);
    }
    issue(0);
    if (nk > 1) issue(1);
    if (nk > 2) issue(2);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 3 < nk) issue(kt + 3);
        const half_t* st = lds + (kt & 3) * DMA64_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah[1], al[1], wh[1], wl[1];
            ah[0] = *reinterpret_cast<const f16x8*>(st + offA[ks]);
            al[0] = *reinterpret_cast<const f16x8*>(st + DMA64_PLANE + offA[ks]);
            wh[0] = *reinterpret_cast<const f16x8*>(st + 2 * DMA64_PLANE + offW[ks]);
            wl[0] = *reinterpret_cast<const f16x8*>(st + 3 * DMA64_PLANE + offW[ks]);
            mfma3<1, 1, X2, csl_rowwise<EPI, OUT>()>(ah, al, wh, wl, accm);
        }
    }
    if constexpr (csl_rowwise<EPI, OUT>()) {
        csl_swapped_epilogue<WM, WN, EPI, OUT, X2>(g, accm, m0 + wr * WM * 32, n0 + wc * WN * 32, l31, hi);
        return;
    }
    gemm_h_epilogue<WM, WN, EPI, OUT, X2>(g, accm, m0, n0, wr, wc, l31, hi, BM, BN);
}

template <int EPI, int OUT, bool X2 = false>
inline hipError_t launch_gemm_h_dma64(const GemmHArgs& g, hipStream_t st) {
    const int ntm = (g.M + 63) / 64, ntn = (g.N + 63) / 64;
    static DevSeen attr_seen;
    if (auto once_ = first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_dma64_kernel<EPI, OUT, X2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)DMA64_LDS_BYTES);
    }
    hipLaunchKernelGGL((gemm_f16x3_dma64_kernel<EPI, OUT, X2>), dim3(ntm * ntn), dim3(256), DMA64_LDS_BYTES, st, g, ntm, ntn);
    return hipGetLastError();
}

template <int EPI, int OUT, bool X2>
inline hipError_t launch_gemm_h_mode(const GemmHArgs& g, hipStream_t st) {
    if constexpr (X2)      // JMID_PREC_F16MX: every shape has its fp8-correction kernel (N a multiple of 128, K of 64: all of the net's GEMMs)
        if (g.W8 && g.K % 64 == 0 && g.N % 128 == 0 && tune().gemm_h_variant != 1 && tune().gemm_h_variant != 2)
            return launch_gemm_mx<EPI, OUT>(g, st);
    const long big = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    // 0 auto, 1 = 64x64 register-staged, 2 = 128x128 register-staged, 3 = 128x128 LDS-DMA, 4 = 256x128 LDS-DMA,
    // 5 = 64x64 LDS-DMA, 6 = 256x256 LDS-DMA (N % 256 == 0)
    const int v = tune().gemm_h_variant;
    if (v == 1) return launch_gemm_h_cfg<1, 1, EPI, OUT, X2>(g, st);
    if (v == 2) return launch_gemm_h_cfg<2, 2, EPI, OUT, X2>(g, st);
    if (v == 3) return launch_gemm_h_dma<EPI, OUT, X2>(g, st);
    if (v == 4) return launch_gemm_h_dma256<EPI, OUT, X2>(g, st);
    if (v == 5) return launch_gemm_h_dma64<EPI, OUT, X2>(g, st);
    // (the ConcatSquash epilogue next to 128 accumulators spills: that shape is not instantiated for it)
    if constexpr (EPI != EPI_CSL)
        if (v == 6 && g.N % 256 == 0) return launch_gemm_h_dma256x256<EPI, OUT, X2>(g, st);
    if (big < 256) return launch_gemm_h_dma64<EPI, OUT, X2>(g, st);
    // auto: 256x128 unless the coarser grid quantises badly onto the 256 CUs (one workgroup per CU)
    const long nb256 = (long)((g.M + 255) / 256) * ((g.N + 127) / 128);
    auto eff = [](long nb) { return (double)nb / (double)(((nb + 255) / 256) * 256); };
    // 256x256 when N allows it (in_proj, linear1) and the grid still fills the chip; the ConcatSquash epilogue needs
    // too many registers next to the 128 accumulators
    if constexpr (EPI != EPI_CSL) if (g.N % 256 == 0) {
        // (below one workgroup per CU too from 12288 rows - two chunks are in flight: 2 x 12 episodes per call 76.0 vs 79.7 ms
        // in F16X2, 102.2 vs 107.5 in F16X3; 2 x 8: within 1 %)
        const long nbq = (long)((g.M + 255) / 256) * (g.N / 256);
        if ((nbq >= 256 && 1.2 * eff(nbq) >= eff(nb256)) || (nbq < 256 && g.M >= 12288)) return launch_gemm_h_dma256x256<EPI, OUT, X2>(g, st);
    }
    if (nb256 >= 256 && 1.2 * eff(nb256) >= eff(big)) return launch_gemm_h_dma256<EPI, OUT, X2>(g, st);
    return launch_gemm_h_dma<EPI, OUT, X2>(g, st);
}

// launches of at most one workgroup per CU: gemm_small.hpp (defined after the LayerNorm headers it builds on)
template <int EPI, int OUT>
inline hipError_t launch_gemm_small(const GemmHArgs& g, int wc, hipStream_t st);
inline int small_gemm_shape(const GemmHArgs& g);

// the arithmetic mode is a template parameter of every kernel (a run-time flag in the K loops cost F16X3 4 %)
template <int EPI, int OUT>
inline hipError_t launch_gemm_h(const GemmHArgs& g, hipStream_t st) {
    if (const int wc = small_gemm_shape(g)) return launch_gemm_small<EPI, OUT>(g, wc, st);      // at most one workgroup per CU
    return g.x2 ? launch_gemm_h_mode<EPI, OUT, true>(g, st) : launch_gemm_h_mode<EPI, OUT, false>(g, st);
}

// fp32 -> hi/lo planes (weights at load time, activations produced by fp32-only kernels)
static __global__ void split_planes_kernel(const float* in, half_t* hi, half_t* lo, size_t n, int* range_flag) {
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

// fp32 row-major [rows, K] -> hi/lo planes in the blocked panel layout (weights at load time, diagnostics)
static __global__ void split_planes_blocked_kernel(const float* in, half_t* hi, half_t* lo, int rows, int K, int* range_flag,
                                            float scale) {   // scale = kWScale for weights, 1 for activations
    bool overflow = false;
    const size_t n = (size_t)rows * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / K), k = (int)(i % K);
        const float v = in[i] * scale;
        half_t h, l;
        split_f32(v, h, l);
        overflow |= !(fabsf(v) <= kHalfMax);
        const size_t o = blk_index(r, k, K);
        hi[o] = h;
        lo[o] = l;
    }
    if (overflow && range_flag) atomicOr(range_flag, 1);
}

// V planes [nseq*S, d] (token-major, as the QKV GEMM emits them) -> V^T planes [nseq][nhead][hd][Spad]
// (key-contiguous: the k-operand layout of the PV product).  64x64 tiles through LDS; both planes per block.
// Key order inside every 16-key group: common.hpp::vt_key_pos.   grid = (ceil(S/64), d/64, nseq)
static __global__ __launch_bounds__(256) void v_transpose_kernel(const half_t* vh, const half_t* vl, half_t* vth, half_t* vtl,
                                                          int S, int Spad, int d, int hd) {
    __shared__ half_t tile[2][64][64 + 8];
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * 64, c0 = blockIdx.y * 64, seq = blockIdx.z;
    const size_t tok0 = (size_t)seq * S;
    // load: 64 keys x 8 chunks of 8 columns, two planes
    for (int id = tid; id < 1024; id += 256) {
        const int p = id >> 9, r = (id >> 3) & 63, c = id & 7;
        const int key = k0 + r;
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (key < S) v = *reinterpret_cast<const f16x8*>((p ? vl : vh) + (tok0 + key) * d + c0 + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[p][r][c * 8 + e] = v[e];
    }
    __syncthreads();
    // store: 64 columns x 8 chunks of 8 keys
    for (int id = tid; id < 1024; id += 256) {
        const int p = id >> 9, col = (id >> 3) & 63, kc = id & 7;
        const int key = k0 + kc * 8;
        if (key >= Spad) continue;
        const int cg = c0 + col, head = cg / hd, vc = cg - head * hd;
        f16x8 v;   // stored position kc*8 + e holds key vt_key_pos(kc*8 + e) of this 64-key tile
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[p][vt_key_pos(kc * 8 + e)][col];
        const size_t o = (((size_t)seq * (d / hd) + head) * hd + vc) * Spad + key;
        *reinterpret_cast<f16x8*>((p ? vtl : vth) + o) = v;
    }
}

// blocked hi/lo planes [rows, K] -> fp32 row-major (diagnostics)
static __global__ void merge_planes_kernel(const half_t* hi, const half_t* lo, float* out, int rows, int K) {
    const size_t n = (size_t)rows * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t o = blk_index((int)(i / K), (int)(i % K), K);
        out[i] = (float)hi[o] + (float)lo[o];
    }
}

// packed fp32 QKV [M, 3d] -> Q/K planes [M, d] + V^T planes [nseq][nhead][hd][Spad] (diagnostics; the pipeline
// gets these straight from the QKV GEMM epilogue)
static __global__ void qkv_to_planes_kernel(const float* qkv, half_t* qh, half_t* ql, half_t* kh, half_t* kl, half_t* vth,
                                     half_t* vtl, size_t M, int d, int hd, int S, int Spad, float qscale) {
    const size_t n = M * d;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / d;
        const int c = (int)(i % d);
        half_t h, l;
        split_f32_unscaled(qkv[m * 3 * d + c] * qscale, h, l);
        qh[i] = h; ql[i] = l;
        split_f32_unscaled(qkv[m * 3 * d + d + c], h, l);
        kh[i] = h; kl[i] = l;
        split_f32_unscaled(qkv[m * 3 * d + 2 * d + c], h, l);
        const size_t seq = m / S, key = m % S;
        const int head = c / hd, vc = c % hd;
        const size_t o = ((seq * (d / hd) + head) * hd + vc) * Spad + (size_t)vt_key_pos((int)key);
        vth[o] = h; vtl[o] = l;
    }
}

}  // namespace jmid
