// Flash attention with fp32-class accuracy on the fp16 matrix cores (hi/lo split operands, see gemm_f16x3.hpp).
//
// Same transposed formulation as attn_f32.hpp (a lane owns one query row, softmax state is lane-local):
//   S^T = K . Q^T      3 MFMAs per 16-deep step:  Khi.Qhi + Khi.Qlo + Klo.Qhi
//   O^T = V^T . P^T    P is split to hi/lo in registers straight from the S^T accumulators; V^T comes
//                      key-contiguous from the QKV GEMM epilogue (key order inside 16-key groups:
//                      common.hpp::vt_key_pos), so no transpose is needed on the way in.
// The lo planes are lo = fp16(x - hi): v_mfma_f32_32x32x16_f16 takes fp16 subnormal inputs exactly (probed on gfx950,
// tools/mfma_denorm.hip), so hi.hi + hi.lo + lo.hi all accumulate into ONE fp32 accumulator (the pair carries x to
// max(2^-22 |x|, 2^-25) absolute, which is what the O(1) Q/K/V/P values need).  Q arrives pre-multiplied by log2(e)/sqrt(head_dim) from the QKV GEMM
// epilogue, so the scores are already in log2 units: softmax is one v_exp_f32 per element.
//
// Inputs : Qhi/Qlo, Khi/Klo [nseq*S, d] planes;  Vthi/Vtlo [nseq][nhead][hd][Spad] planes (all lo unscaled).
// Output : hi/lo planes [nseq*S, d] in the blocked panel layout (operand of the attention out-projection GEMM).
#pragma once
#include "common.hpp"
#include <algorithm>

#include "gemm_f16x3.hpp"

namespace jmid {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// two fp32 -> packed fp16 pair (round toward zero; the residual goes to the lo plane, so the mode does not matter)
__device__ __forceinline__ unsigned int pk_f16(float a, float b) {
    return __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(a, b));
}
// the same, rounded to nearest even (v_cvt_pk_f16_f32)
__device__ __forceinline__ unsigned int pk_f16_rne(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_{a, b}, f16x2));
}
// acc + lo(pair) + hi(pair) of a packed fp16 pair, in fp32: ONE v_dot2_f32_f16 against {1, 1}
__device__ __forceinline__ float dot2_ones(unsigned int pair, float acc) {
    typedef __fp16 fp16x2_ __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(fp16x2_, pair), __builtin_bit_cast(fp16x2_, 0x3C003C00u), acc, false);
}
// p[0..7] -> hi fragment and unscaled lo fragment (8 fp16 each)
__device__ __forceinline__ void split8(const float* p, f16x8& hi, f16x8& lo) {
    u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const auto hp = __builtin_amdgcn_cvt_pkrtz(p[2 * i], p[2 * i + 1]);
        h[i] = __builtin_bit_cast(unsigned int, hp);
        l[i] = pk_f16(p[2 * i] - (float)hp[0], p[2 * i + 1] - (float)hp[1]);
    }
    hi = __builtin_bit_cast(f16x8, h);
    lo = __builtin_bit_cast(f16x8, l);
}

// The reference maximum of the online softmax is LAZY: it moves to the row's new tile maximum only when that exceeds it by more than
// ATT_LAZY_TAU (log2 units), so P = 2^(s - m) can reach 2^8 - exact in fp32, far inside fp16 once rounded, and the row sum carries the
// same factor: O / l is the same softmax.  The strict form max(m, tile max) rescales the O accumulators whenever ANY of a wave's 32
// rows finds a new maximum - with 38 key tiles that is more than every second tile; the lazy one after the first tile only.  (For the
// one-wave-per-SIMD kernel of attn_q64.hpp, whose O lives in the accumulation registers, a rescale is 192 instructions per block.)
constexpr float ATT_LAZY_TAU = 8.0f;
__device__ __forceinline__ float att_lazy_max(float m_run, float tile_max) { return tile_max > m_run + ATT_LAZY_TAU ? tile_max : m_run; }

struct AttnHArgs {
    const half_t *Qhi, *Qlo, *Khi, *Klo, *Vthi, *Vtlo;
    half_t *Ohi, *Olo;
    int S, Spad, d, nhead;
    float scale;       // unused by the kernels (Q is pre-scaled); kept for the diagnostics path
    int* range_flag;
    // split-KV (few sequences, e.g. one scene): the key range is divided over nsplit workgroups per q-tile; each
    // writes an un-normalised partial O plus its (max, sum) and attn_combine_kernel merges them
    int nsplit;
    float* Opart;      // [nsplit][nseq*S][d] fp32
    float* MLpart;     // [nsplit][nseq*S][nhead][2]
    int x2;            // JMID_PREC_F16X2: V enters P.V as its hi plane only, and so does P in the head_dim-128 DMA kernel (the logits keep all terms)
    // JMID_PREC_F16MX (head_dim 128): bf8 images of K_hi and K_lo, [nseq*S, d] bytes each (written by the QKV GEMM instead
    // of the fp16 K_lo plane): the two correction terms of the logits run as bf8 x bf8 MFMAs.  Null: the fp16 terms.
    const unsigned char *K8h, *K8l;
    const unsigned char* Q8l;    // with them: bf8 image of Q_lo in the Q_lo plane's place (null: made here from the fp16 plane)
    // reciprocals of the three divisors of the workgroup index (q-tiles, splits, heads), filled in by launch_attn_f16x3: the
    // kernel's index arithmetic was 625 instructions before its first copy went out (nine run-time integer divisions, two of
    // them 64-bit: 1.8 us of a one-scene wave's 10 us).  0 = not filled in (probes that launch the kernel directly): divide.
    unsigned mq = 0, ms = 0, mh = 0;
    int nseq = 0;
    int skip_combine = 0;        // split-KV launches: the partial outputs are merged by the consumer (gemm_small.hpp, lnx_combine), not by attn_combine_kernel
    int prio = 0;                // static s_setprio 1 for half of the workgroups (two share a CU, one wave of each per SIMD): 0 none, 1 = odd
                                 // workgroups of an XCD's sequence, 2 = every second group of 32 of that sequence (the second to land on each CU)
};


template <int HD>
__global__ __launch_bounds__(256, 1) void attn_f16x3_kernel(AttnHArgs a) {
    constexpr int KT = 32;                    // keys per tile
    constexpr int KLD = HD + 8;               // halfs per K row in LDS (row = 2*HD + 16 bytes)
    constexpr int VLD = KT + 4;               // halfs per V^T row in LDS (72 bytes)
    constexpr int NT = (HD + 31) / 32;        // 32-wide tiles of the head dim in O^T
    constexpr int NKS = HD / 16;              // 16-deep steps of the QK^T contraction
    constexpr int VROWS = NT * 32;
    __shared__ __attribute__((aligned(16))) half_t Ksh[KT * KLD], Ksl[KT * KLD];
    __shared__ __attribute__((aligned(16))) half_t Vsh[VROWS * VLD], Vsl[VROWS * VLD];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y, seq = blockIdx.z;
    const int S = a.S, d = a.d;
    const size_t tok0 = (size_t)seq * S;
    const int q = (blockIdx.x * 4 + wid) * 32 + l31;
    const int qc = q < S ? q : S - 1;

    // Q fragments (B operand of S^T): 8 consecutive d per lane and step; the softmax scale is applied to the
    // fp32 scores, not to the fp16 operands
    f16x8 qh[NKS], ql[NKS];
    {
        const size_t o = (tok0 + qc) * d + h * HD + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            qh[ks] = *reinterpret_cast<const f16x8*>(a.Qhi + o + 16 * ks);
            ql[ks] = *reinterpret_cast<const f16x8*>(a.Qlo + o + 16 * ks);
        }
    }
    f32x16 ot[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[n][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // zero the padding rows of the V^T tile once (HD < 32)
    if (HD % 32 != 0) {
        for (int i = tid; i < VROWS * VLD; i += 256) {
            Vsh[i] = (half_t)0.f;
            Vsl[i] = (half_t)0.f;
        }
    }
    const half_t* kh_g = a.Khi + tok0 * d + h * HD;
    const half_t* kl_g = a.Klo + tok0 * d + h * HD;
    const size_t vt0 = ((size_t)seq * a.nhead + h) * HD * a.Spad;
    constexpr int KCH = HD / 8;               // 16-byte chunks per K row
    const int ntiles = (S + KT - 1) / KT;

    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();                      // previous tile fully consumed (also orders the padding zero-fill)
        // ---- stage K tile [KT][HD] (rows past S zero-filled)
        for (int idx = tid; idx < KT * KCH; idx += 256) {
            const int row = idx / KCH, c = idx % KCH;
            const int key = kt * KT + row;
            f16x8 vh = {0, 0, 0, 0, 0, 0, 0, 0}, vl = {0, 0, 0, 0, 0, 0, 0, 0};
            if (key < S) {
                vh = *reinterpret_cast<const f16x8*>(kh_g + (size_t)key * d + c * 8);
                vl = *reinterpret_cast<const f16x8*>(kl_g + (size_t)key * d + c * 8);
            }
            *reinterpret_cast<f16x8*>(&Ksh[row * KLD + c * 8]) = vh;
            *reinterpret_cast<f16x8*>(&Ksl[row * KLD + c * 8]) = vl;
        }
        // ---- stage V^T tile [HD][KT]: 8-byte pieces (4 keys); keys past Spad zero-filled
        for (int idx = tid; idx < HD * (KT / 4); idx += 256) {
            const int row = idx / (KT / 4), c = idx % (KT / 4);
            const int key = kt * KT + c * 4;
            f16x4 vh = {0, 0, 0, 0}, vl = {0, 0, 0, 0};
            if (key < a.Spad) {
                vh = *reinterpret_cast<const f16x4*>(a.Vthi + vt0 + (size_t)row * a.Spad + key);
                vl = *reinterpret_cast<const f16x4*>(a.Vtlo + vt0 + (size_t)row * a.Spad + key);
            }
            *reinterpret_cast<f16x4*>(&Vsh[row * VLD + c * 4]) = vh;
            *reinterpret_cast<f16x4*>(&Vsl[row * VLD + c * 4]) = vl;
        }
        __syncthreads();

        // ---- S^T = K . Q^T  (rows = keys, cols = queries)
        f32x16 sm;
#pragma unroll
        for (int r = 0; r < 16; ++r) sm[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const f16x8 kh = *reinterpret_cast<const f16x8*>(&Ksh[l31 * KLD + 16 * ks + 8 * hi]);
            const f16x8 kl = *reinterpret_cast<const f16x8*>(&Ksl[l31 * KLD + 16 * ks + 8 * hi]);
            sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], sm, 0, 0, 0);
            sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], sm, 0, 0, 0);
            sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], sm, 0, 0, 0);
        }
        // ---- online softmax (fp32) over the 32 keys of this tile
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = sm[r];
            const int key = kt * KT + frag_row(r, hi);
            s = key < S ? s : -INFINITY;
            sm[r] = s;
            tmax = fmaxf(tmax, s);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = att_lazy_max(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sm[r] = __builtin_amdgcn_exp2f(sm[r] - m_new);
            psum += sm[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = fmaf(l_run, alpha, psum);
        m_run = m_new;
        // P fragments (B operand of O^T): register r = 8*mf + j holds key frag_row(r, hi)
        f16x8 ph[2], pl[2];
        {
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[r] = sm[r];
            split8(pv, ph[0], pl[0]);
            split8(pv + 8, ph[1], pl[1]);
        }
        // ---- O^T = alpha * O^T + V^T . P^T
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[n][r] *= alpha;
            const int vrow = n * 32 + l31;
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                // keys of this lane-half for MFMA mf: {16mf + 4hi + 0..3, 16mf + 8 + 4hi + 0..3} = stored positions
                // 16mf + 8hi .. +7 (common.hpp::vt_key_pos)
                const int c0 = vrow * VLD + 16 * mf + 8 * hi;
                const f16x4 vh0 = *reinterpret_cast<const f16x4*>(&Vsh[c0]);
                const f16x4 vh1 = *reinterpret_cast<const f16x4*>(&Vsh[c0 + 4]);
                const f16x4 vl0 = *reinterpret_cast<const f16x4*>(&Vsl[c0]);
                const f16x4 vl1 = *reinterpret_cast<const f16x4*>(&Vsl[c0 + 4]);
                const f16x8 vh = {vh0[0], vh0[1], vh0[2], vh0[3], vh1[0], vh1[1], vh1[2], vh1[3]};
                const f16x8 vl = {vl0[0], vl0[1], vl0[2], vl0[3], vl1[0], vl1[1], vl1[2], vl1[3]};
                ot[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[mf], ot[n], 0, 0, 0);
                ot[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[mf], ot[n], 0, 0, 0);
                if (!a.x2) ot[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[mf], ot[n], 0, 0, 0);
            }
        }
    }

    // ---- normalise, split and store: register r of tile n is head-dim n*32 + frag_row(r, hi)
    if (q < S) {
        const float inv = 1.0f / l_run;
        const int orow = (int)tok0 + q;
        bool overflow = false;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int c0 = n * 32 + 8 * r4 + 4 * hi;
                if (HD % 32 == 0 || c0 < HD) {
                    f16x4 vh, vl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = ot[n][4 * r4 + e] * inv;
                        half_t hh, ll;
                        split_f32(v, hh, ll);
                        overflow |= !(fabsf(v) <= kHalfMax);
                        vh[e] = hh;
                        vl[e] = ll;
                    }
                    const size_t ob = blk_index(orow, h * HD + c0, d);
                    *reinterpret_cast<f16x4*>(a.Ohi + ob) = vh;
                    if (!a.x2) *reinterpret_cast<f16x4*>(a.Olo + ob) = vl;   // F16X2: out_proj reads O_hi only
                }
            }
        }
        if (overflow) atomicOr(a.range_flag, 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant for head_dim 128: K and V^T tiles are copied HBM/L2 -> LDS by global_load_lds_dwordx4 into a
// 2-stage ring, the copy of tile t+1 overlapping the MFMAs of tile t (one raw s_barrier per tile, counted vmcnt).
// LDS rows are unpadded; bank-conflict swizzles live on the DMA source address:
//   K  tile [32 keys][16 chunks of 16 B]: chunk c of row r stored at c ^ (r & 15)      (conflict-free b128 reads)
//   V^T tile [128 d][4 chunks of 16 B]  : chunk c of row r stored at c ^ ((r>>2) & 3); a chunk is exactly the 8 keys
//                                         one lane-half feeds to a PV MFMA (common.hpp::vt_key_pos) -> one b128 read
constexpr int ATT_KPLANE = 32 * 128;                        // halfs per K plane per stage (8 KB)
constexpr int ATT_VPLANE = 128 * 32;                        // halfs per V^T plane per stage (8 KB)
constexpr int ATT_STAGE = 2 * ATT_KPLANE + 2 * ATT_VPLANE;  // Kh, Kl, Vh, Vl = 32 KB
constexpr size_t ATT_DMA_LDS = size_t(2) * ATT_STAGE * sizeof(half_t);

// TRACE (tools/attn_trace.hip only): every wave accumulates s_memtime deltas per phase of the key-tile loop into trace[].
#define ATT_STAMP(i)                                                  \
    if (TRACE) {                                                      \
        __builtin_amdgcn_sched_barrier(0);                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        tacc[i] += now_ - tprev;                                      \
        tprev = now_;                                                 \
        __builtin_amdgcn_sched_barrier(0);                            \
    }
// MX (JMID_PREC_F16MX): S^T = K_hi . Q_hi in fp16 as before, and the two correction terms as ONE bf8 x bf8
// v_mfma_f32_32x32x64_f8f6f4 each per 64-deep half of the head: bf8(K_hi) . bf8(Q_lo) + bf8(K_lo) . bf8(Q_hi) - 16 instead of
// 24 fp16-MFMA times per key tile.  The terms are 2^-12 of the logit; with 2-bit significands they are good to ~2^-15 of it,
// which moves the attention output by 2^-18.7 (flat softmax) ... 2^-14 (logit spread 4 nats, a key at p = 0.95) relative -
// below the 2^-12.7 of the mode's GEMMs.  bf8(K) comes from the QKV GEMM epilogue (two 4 KB images per key tile in the
// place of the 8 KB fp16 K_lo image: same DMA bytes, same fragment-read bytes, no extra VALU work per tile); bf8(Q) is made
// once per wave from the Q planes.  k order of the fp8 instruction: byte p of lane (row, h) is head dim 64 blk + 32 h + p.
// P1 (the mode's default): P.V with ONE fp16 plane of P, rounded to nearest - the same single rounding the mode gives every other
// activation (V itself is one plane): 8 instead of 16 MFMAs per key tile and no lo half of the split.  ADE against exact fp32
// 1.164e-5 m with or without the P_lo term.
// (PIPE, a software-pipelined key-tile loop - the softmax of tile t in the shadow of the QK^T MFMAs of tile t + 1 - measured 1.7 %
// slower in round 2 and was removed in round 3; the template parameter is kept so that kernel names stay comparable across profiles.)
// SM (round 6) - the softmax of a key tile with the vector instructions that are not exponentials, conversions or the row sum taken out of
// the common path (the SQ's counters put 120 vector instructions beside 19 matrix instructions per wave and key tile, and the wave pair of
// a SIMD is bound by their issue slots - docs/NOTEBOOK.md section 10):
//   * the reference maximum m enters the logits through the ACCUMULATOR: the first matrix instruction of a tile takes C = -m (sixteen
//     registers that change only when m moves), so s - m leaves the matrix pipe and the sixteen subtractions are gone;
//   * no tile maximum on the common path: p = 2^(s - m) is formed optimistically and the ROW SUM of the tile says whether m was too
//     low (sum > ATT_SM_THR = 2^14, or not finite: some p is about to leave the fp16 range) - only then (and in a wave's first tile) the wave
//     takes the slow path: tile maximum, m moves there, O and l rescaled, p formed again.  m is any earlier tile maximum of the row, as
//     with the lazy maximum of SM = 0; O / l is the same softmax;
//   * P1 (F16X2 / F16MX): the row sum is taken from the PACKED fp16 P that multiplies V (v_dot2_f32_f16 against {1, 1}: eight
//     instructions instead of sixteen, and l normalises exactly the P that was used).
// SM = 0: the round 2-5 form (tile maximum, lazy reference maximum, subtraction, fp32 row sum); diagnostics A/B ("attn_sm" = 2).
constexpr float ATT_SM_THR = 16384.0f;      // the largest P stays below 2^14 (fp16 holds 2^16); every P keeps fp16's relative precision whatever its size
template <bool TRACE, bool X2 = false, bool MX = false, bool PIPE = false, bool P1 = false, bool PF = false, int SM = 1>
__global__ __launch_bounds__(256, 2) void attn_f16x3_dma_kernel(AttnHArgs a, int nqt, int abl, unsigned long long* trace) {
    constexpr int HD = 128, KT = 32, NT = 4, NKS = 8;
    args_now_each(a, nqt, abl, trace);
#ifndef JMID_ABLATIONS
    if (!TRACE) abl = 0;      // (the launch helpers pass 0 then: said here, every `abl` test below folds away)
#endif
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0, tstart = 0;
    unsigned long long rt0 = 0;
    if (TRACE) {
        tstart = tprev = __builtin_amdgcn_s_memtime();
        rt0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char att_lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(att_lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // XCD-aware order: the q-tiles of one (sequence, head) share K/V, keep them on one XCD's L2
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    if (a.prio) {
        if (a.prio == 1 ? ((b >> 3) & 1) : ((b >> 8) & 1)) __builtin_amdgcn_s_setprio(1);
    }
    const int sh0 = fast_div(swz, nqt, a.mq), qt = swz - sh0 * nqt;
    const int sh = fast_div(sh0, a.nsplit, a.ms), split = sh0 - sh * a.nsplit;
    const int seq = fast_div(sh, a.nhead, a.mh), h = sh - seq * a.nhead;
    const int S = a.S, d = a.d;
    const size_t tok0 = (size_t)seq * S;
    const int q = (qt * 4 + wid) * 32 + l31;
    const int qc = q < S ? q : S - 1;
    // all 32 queries of this wave lie past the end of the sequence (last q-tile): it keeps copying its share of every
    // K / V^T tile and meets the barriers, but computes nothing
    const bool wave_idle = (qt * 4 + wid) * 32 >= S;

    // Q operands: the raw loads go out here, their conversions (MX: bf8 images) wait until the first key tile's copies have been
    // issued (q_finish below) - the two memory round trips of a workgroup's start overlap instead of adding up (one scene: the
    // prologue was 31 % of a wave's life, tools/attn_small_trace.hip)
    f16x8 qh[NKS], ql[MX ? 1 : NKS];
    i32x8 q8h[2], q8l[2];            // MX: bf8 images of this lane's Q_hi / Q_lo row, 32 head dims per 64-deep block
    i32x4 q8raw[2][4], q8lraw[2][4];
    {
        const size_t o = (tok0 + qc) * d + h * HD + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            qh[ks] = *reinterpret_cast<const f16x8*>(a.Qhi + o + 16 * ks);
            if (!MX) ql[ks] = *reinterpret_cast<const f16x8*>(a.Qlo + o + 16 * ks);
        }
        if (MX) {
            const size_t o8 = (tok0 + qc) * d + h * HD + 32 * hi;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    q8raw[blk][c] = __builtin_bit_cast(i32x4, *reinterpret_cast<const f16x8*>(a.Qhi + o8 + 64 * blk + 8 * c));
            if (a.Q8l) {      // the QKV GEMM wrote the image (wave-uniform)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    q8lraw[blk][0] = *reinterpret_cast<const i32x4*>(a.Q8l + o8 + 64 * blk);
                    q8lraw[blk][1] = *reinterpret_cast<const i32x4*>(a.Q8l + o8 + 64 * blk + 16);
                }
            } else {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        q8lraw[blk][c] = __builtin_bit_cast(i32x4, *reinterpret_cast<const f16x8*>(a.Qlo + o8 + 64 * blk + 8 * c));
            }
        }
    }
    auto q_finish = [&]() {
        if (MX) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    q8h[blk][2 * c] = bf8_of_f16x4(q8raw[blk][c][0], q8raw[blk][c][1]);
                    q8h[blk][2 * c + 1] = bf8_of_f16x4(q8raw[blk][c][2], q8raw[blk][c][3]);
                }
            if (a.Q8l) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const i32x4 l0 = q8lraw[blk][0], l1 = q8lraw[blk][1];
                    q8l[blk] = i32x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                }
            } else {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        q8l[blk][2 * c] = bf8_of_f16x4(q8lraw[blk][c][0], q8lraw[blk][c][1]);
                        q8l[blk][2 * c + 1] = bf8_of_f16x4(q8lraw[blk][c][2], q8lraw[blk][c][3]);
                    }
            }
        }
        // the Q loads are ordinary VMEM loads, older than the first tile's copies: pinned here, before the key-tile loop, so that
        // the loop's waits concern the DMA ring only
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qh[ks]));
        if (!MX) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(ql[ks]));
        } else {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) asm volatile("" : "+v"(q8h[blk]), "+v"(q8l[blk]));
        }
    };

    f32x16 ot[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[n][r] = 0.f;
    float m_run = SM ? 0.f : -INFINITY, l_run = 0.f;   // running max in log2 units (Q is pre-scaled by log2(e)/sqrt(hd))
    f32x16 negm;                             // SM: -m_run in all sixteen registers, the C operand of a tile's first matrix instruction
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;

    // DMA sources.  K rounds 0-3: plane = i>>1, row = 16*(i&1) + tid/16, stored chunk tid&15.
    //               V rounds 4-7: plane = (i-4)>>1, row = 64*(i&1) + tid/4, stored chunk tid&3.
    const half_t* kh_g = a.Khi + tok0 * d + h * HD;
    const half_t* kl_g = a.Klo + tok0 * d + h * HD;
    const size_t vt0 = ((size_t)seq * a.nhead + h) * HD * a.Spad;
    const int k_row = tid >> 4, k_c = (tid & 15) ^ (k_row & 15);             // rows 0-15 (+16 for odd rounds)
    const int v_row = tid >> 2, v_c = (tid & 3) ^ ((v_row >> 2) & 3);       // rows 0-63 (+64 for odd rounds)
    const int last_vchunk = a.Spad / 8 - 1;
    // one of the 8 DMA wave-instructions of key tile kt (i < 4: K planes, else V^T planes): source = wave-uniform base (scalar
    // registers: plane pointer + kt * tile stride) + a per-thread 32-bit offset - no vector address arithmetic per copy (the
    // per-copy form cost ~110 VALU instructions per key tile, more than the softmax).  The offsets come in two sets, computed
    // once: the regular one, and the one of the sequence's LAST tile, whose rows past S / chunks past Spad are clamped to valid
    // memory (the keys are masked afterwards); `use_last_offsets` switches sets when that tile's copies are due.
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);
    const char* const kh_b = reinterpret_cast<const char*>(kh_g);
    const char* const kl_b = reinterpret_cast<const char*>(kl_g);
    const char* const k8h_b = reinterpret_cast<const char*>(a.K8h) + (tok0 * d + h * HD);
    const char* const k8l_b = reinterpret_cast<const char*>(a.K8l) + (tok0 * d + h * HD);
    const char* const vth_b = reinterpret_cast<const char*>(a.Vthi + vt0);
    const char* const vtl_b = reinterpret_cast<const char*>(a.Vtlo + vt0);
    const int last_tile = (S + KT - 1) / KT - 1;
    const int rows_last = S - last_tile * KT - 1;                 // highest valid row of the last tile
    const int chunks_last = last_vchunk - last_tile * 4;          // highest valid 16-byte chunk of its V^T rows
    auto rowc = [&](int r) { return r < rows_last ? r : rows_last; };
    unsigned offK16[2] = {(unsigned)(k_row * d + k_c * 8) * 2u, (unsigned)((16 + k_row) * d + k_c * 8) * 2u};   // fp16 K planes
    unsigned offK8 = (unsigned)((tid >> 3) * d) + (unsigned)(((tid & 7) ^ (((tid >> 3) >> 1) & 7)) << 4);          // bf8 images
    unsigned offV = (unsigned)(v_row * a.Spad + v_c * 8) * 2u;                                                   // V^T planes, rows 0-63 (+ 64)
    const unsigned offK16_last[2] = {(unsigned)(rowc(k_row) * d + k_c * 8) * 2u, (unsigned)(rowc(16 + k_row) * d + k_c * 8) * 2u};
    const unsigned offK8_last = (unsigned)(rowc(tid >> 3) * d) + (unsigned)(((tid & 7) ^ (((tid >> 3) >> 1) & 7)) << 4);
    const unsigned offV_last = (unsigned)(v_row * a.Spad + (v_c < chunks_last ? v_c : chunks_last) * 8) * 2u;
    auto use_last_offsets = [&]() {
        offK16[0] = offK16_last[0];
        offK16[1] = offK16_last[1];
        offK8 = offK8_last;
        offV = offV_last;
    };
    auto issue_one = [&](int kt, int i, int stage) {
        if (i >= 6 && X2) return;   // F16X2: the V^T lo plane is neither written by the QKV epilogue nor read here
        half_t* st = lds + stage * ATT_STAGE + wid_s * 512;
        const char* src;
        half_t* dst;
        if (MX && (i == 2 || i == 3)) {
            // a whole bf8 K image: 32 keys x 128 bytes; thread = (key row tid / 8, stored 16-byte chunk tid % 8), which
            // holds source chunk (tid % 8) ^ ((row >> 1) & 7): conflict-free ds_read_b128 of a lane-half's two chunks
            src = (i == 2 ? k8h_b : k8l_b) + (size_t)kt * (KT * d) + offK8;
            dst = st + i * 2048;
        } else if (i < 4) {
            src = ((i >> 1) ? kl_b : kh_b) + (size_t)kt * (KT * d) * 2 + offK16[i & 1];
            dst = st + i * 2048;
        } else {
            const int j = i - 4;
            src = ((j >> 1) ? vtl_b : vth_b) + (size_t)(64 * (j & 1)) * a.Spad * 2 + (size_t)kt * 64 + offV;
            dst = st + 2 * ATT_KPLANE + j * 2048;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int i = 0; i < 8; ++i) issue_one(kt, i, stage);
    };
    // fragment read offsets (halfs), kept to a handful of registers:
    //   K : row l31, chunk (2ks+hi) ^ (l31&15)                        -> kbase + (((2ks+hi) ^ kx) << 3)
    //   V : row n*32+l31, chunk (2mf+hi) ^ ((row>>2)&3) (same for every n)  -> vbase[mf] + n*1024
    const int kbase = l31 * 128, kx = l31 & 15;
    int vbase[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) vbase[mf] = l31 * 32 + (((2 * mf + hi) ^ ((l31 >> 2) & 3)) * 8);

    const int ntiles_all = (S + KT - 1) / KT;
    const bool fd = a.ms != 0 || a.nsplit == 1;
    const int kt_begin = fd ? fast_div(split * ntiles_all, a.nsplit, a.ms) : (int)((long)split * ntiles_all / a.nsplit);
    const int ntiles = fd ? fast_div((split + 1) * ntiles_all, a.nsplit, a.ms)
                          : (int)((long)(split + 1) * ntiles_all / a.nsplit);   // exclusive end of this split's key tiles
    {
    if (kt_begin == last_tile) use_last_offsets();
    ATT_STAMP(7)   // arguments, tile arithmetic, Q loads requested
    issue(kt_begin, 0);
    q_finish();
    ATT_STAMP(0)   // prologue: Q loads, first DMA issue
    // One key tile.  STG (the ring stage of tile kt = the parity of kt - kt_begin) and MORE (tile kt + 1 exists: its copies go out during
    // this tile) are compile-time: every LDS address of the tile is then the lane's precomputed offset + an IMMEDIATE (13 vector adds
    // per tile with a run-time stage base), and the eight `if (more)` around the copies are gone from the instruction stream.
    auto tile_body = [&](const int kt, auto stg_c, auto more_c) {
        constexpr int STG = decltype(stg_c)::value;
        constexpr bool MORE = decltype(more_c)::value;
        if (kt + 1 == last_tile) use_last_offsets();       // (uniform: the copies of tile kt + 1 go out during this iteration)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of tile kt has landed
        ATT_STAMP(1)
        __builtin_amdgcn_s_barrier();                      // ... and everybody else's; stage (kt+1)&1 is free again
        __builtin_amdgcn_sched_barrier(0);
        ATT_STAMP(2)
        // the DMA of tile kt+1 is issued one wave-instruction per QK^T step below, behind that step's MFMAs: a burst of
        // 8 right here stalls the wave ~850 cycles per tile in the CU's address path (tools/attn_trace.hip)
        const bool more = MORE && !(abl & 1);    // abl: timing ablations (diagnostics only; zero otherwise, see the top of the kernel)
        if (more && (abl & 16)) issue(kt + 1, 1 - STG);
        ATT_STAMP(3)
        if (wave_idle) {       // S = 1200: 2 of the 40 waves of a (sequence, head) - 5 % of the kernel's MFMA work
            if (more && !(abl & 16)) issue(kt + 1, 1 - STG);
            return;
        }
        const half_t* Kh = lds + ((abl & 1) ? 0 : STG) * ATT_STAGE;
        const half_t* Kl = Kh + ATT_KPLANE;
        const half_t* Vh = Kh + 2 * ATT_KPLANE;
        const half_t* Vl = Vh + ATT_VPLANE;

        f32x16 sm;
        if (SM) {
            sm = negm;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) sm[r] = 0.f;
        }
        constexpr int PFD = 3;        // PF: fragment reads PFD steps ahead (4: no better)
        // ATT_ISSUE_AT (tools/attn_issue_probe.hip only; 0 = what ships): where the copies of tile kt + 1 go out.  0: one per QK^T step, behind
        // that step's matrix instruction.  1: all of them between the last QK^T instruction and the softmax.  2: the same, after ALL eight V^T
        // fragments of this tile have been read into registers - the wave touches LDS again only after its wait at the top of the next tile
#ifdef ATT_ISSUE_AT
        constexpr int IA = (MX && PF && P1 && X2) ? ATT_ISSUE_AT : 0;
#else
        constexpr int IA = 0;
#endif
        f16x8 vpre[IA >= 2 ? 2 * NT : PFD];      // the first V^T fragments, requested before the softmax
        if (MX && PF) {
            // PF: fragment reads three steps ahead (one MFMA per step covers 32 cycles of an LDS round trip of > 64), the bf8 K
            // fragments requested inside the fp16 loop, the first V^T fragments before the softmax
            auto kread = [&](int ks) { return *reinterpret_cast<const f16x8*>(Kh + kbase + (((2 * ks + hi) ^ kx) << 3)); };
            const unsigned char* k8 = reinterpret_cast<const unsigned char*>(Kl);
            const int r8 = l31 * 128, sw = (l31 >> 1) & 7;
            auto k8read = [&](int img, int blk, int c) {
                return *reinterpret_cast<const i32x4*>(k8 + img * 4096 + r8 + (((blk * 4 + hi * 2 + c) ^ sw) << 4));
            };
            f16x8 kf[NKS];
            i32x4 k8f[2][2][2];      // [blk][image][chunk]
#pragma unroll
            for (int i = 0; i < PFD; ++i) kf[i] = kread(i);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks + PFD < NKS) kf[ks + PFD] = kread(ks + PFD);
                if (ks == 5 || ks == 7) {
                    const int blk = ks == 5 ? 0 : 1;
                    k8f[blk][0][0] = k8read(0, blk, 0); k8f[blk][0][1] = k8read(0, blk, 1);
                    k8f[blk][1][0] = k8read(1, blk, 0); k8f[blk][1][1] = k8read(1, blk, 1);
                }
                sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qh[ks], sm, 0, 0, 0);
                if (IA == 0 && more && !(abl & 16)) issue_one(kt + 1, ks, 1 - STG);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (P1 && IA < 2) {
#pragma unroll
                for (int i = 0; i < PFD; ++i) vpre[i] = *reinterpret_cast<const f16x8*>(Vh + (i & 3) * 1024 + vbase[i >> 2]);
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const i32x4 h0 = k8f[blk][0][0], h1 = k8f[blk][0][1], l0 = k8f[blk][1][0], l1 = k8f[blk][1][1];
                const i32x8 kh8 = i32x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                const i32x8 kl8 = i32x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                sm = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kh8, q8l[blk], sm, 1, 1, 0, 0, 0, 0);
                sm = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kl8, q8h[blk], sm, 1, 1, 0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (IA >= 2) {      // (behind the bf8 instructions: their K fragments are dead, 32 registers are free)
#pragma unroll
                for (int i = 0; i < 2 * NT; ++i) vpre[i] = *reinterpret_cast<const f16x8*>(Vh + (i & 3) * 1024 + vbase[i >> 2]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (IA >= 1 && more) {
                if (IA == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the fragments are in registers before the first copy goes out (3: no wait)
                issue(kt + 1, 1 - STG);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MX) {
            // fp16 part: K_hi . Q_hi, fragments read one step ahead; the next tile's DMA instructions go out one per step
            f16x8 kh_c = *reinterpret_cast<const f16x8*>(Kh + kbase + (((0 + hi) ^ kx) << 3));
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                f16x8 kh_n = kh_c;
                if (ks + 1 < NKS) kh_n = *reinterpret_cast<const f16x8*>(Kh + kbase + (((2 * (ks + 1) + hi) ^ kx) << 3));
                sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh_c, qh[ks], sm, 0, 0, 0);
                if (more && !(abl & 16)) issue_one(kt + 1, ks, 1 - STG);
                __builtin_amdgcn_sched_barrier(0);
                kh_c = kh_n;
            }
            // correction terms: bf8(K_hi) . bf8(Q_lo) + bf8(K_lo) . bf8(Q_hi), one instruction per 64-deep block and term
            const unsigned char* k8 = reinterpret_cast<const unsigned char*>(Kl);      // K_hi image, then (4 KB on) the K_lo image
            const int r8 = l31 * 128, sw = (l31 >> 1) & 7;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int c0 = blk * 4 + hi * 2;
                const i32x4 h0 = *reinterpret_cast<const i32x4*>(k8 + r8 + ((c0 ^ sw) << 4));
                const i32x4 h1 = *reinterpret_cast<const i32x4*>(k8 + r8 + (((c0 + 1) ^ sw) << 4));
                const i32x4 l0 = *reinterpret_cast<const i32x4*>(k8 + 4096 + r8 + ((c0 ^ sw) << 4));
                const i32x4 l1 = *reinterpret_cast<const i32x4*>(k8 + 4096 + r8 + (((c0 + 1) ^ sw) << 4));
                const i32x8 kh8 = i32x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                const i32x8 kl8 = i32x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                sm = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kh8, q8l[blk], sm, 1, 1, 0, 0, 0, 0);   // unscaled, both bf8
                sm = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kl8, q8h[blk], sm, 1, 1, 0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else if (!(abl & 8)) {
            // software-pipelined fragment reads: the K fragments of step ks+1 are requested before the MFMAs of step
            // ks issue, so the LDS latency hides under 96 cycles of MFMA instead of stalling the pipe every step
            f16x8 kh_c = *reinterpret_cast<const f16x8*>(Kh + kbase + (((0 + hi) ^ kx) << 3));
            f16x8 kl_c = *reinterpret_cast<const f16x8*>(Kl + kbase + (((0 + hi) ^ kx) << 3));
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                f16x8 kh_n = kh_c, kl_n = kl_c;
                if (ks + 1 < NKS) {
                    const int ok = kbase + (((2 * (ks + 1) + hi) ^ kx) << 3);
                    kh_n = *reinterpret_cast<const f16x8*>(Kh + ok);
                    kl_n = *reinterpret_cast<const f16x8*>(Kl + ok);
                }
                sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh_c, qh[ks], sm, 0, 0, 0);
                sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh_c, ql[MX ? 0 : ks], sm, 0, 0, 0);
                sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl_c, qh[ks], sm, 0, 0, 0);
                if (more && !(abl & 16)) issue_one(kt + 1, ks, 1 - STG);
                __builtin_amdgcn_sched_barrier(0);
                kh_c = kh_n;
                kl_c = kl_n;
            }
        }
        if (PF && P1 && !MX) {
#pragma unroll
            for (int i = 0; i < PFD; ++i) vpre[i] = *reinterpret_cast<const f16x8*>(Vh + (i & 3) * 1024 + vbase[i >> 2]);
        }
        ATT_STAMP(4)
        f16x8 ph[2], pl[2];
        if (SM) {
            if (kt == ntiles_all - 1) {              // only the last tile can hold keys past S
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * KT + frag_row(r, hi) >= S) sm[r] = -INFINITY;
            }
            // sm holds s - m_run.  P = 2^(sm - delta) as the fragments of P.V, psum = the tile's row sum over both lane halves
            float psum = 0.f;
            auto soft = [&](const float delta, auto shifted_c) {
                constexpr bool SHIFTED = decltype(shifted_c)::value;
                float p[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(SHIFTED ? sm[r] - delta : sm[r]);
                float ps = 0.f;
                if (P1) {       // one fp16 plane of P, rounded to nearest; the row sum of exactly those values
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) {
                        u32x4 hq;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            hq[i] = pk_f16_rne(p[8 * mf + 2 * i], p[8 * mf + 2 * i + 1]);
                            ps = dot2_ones(hq[i], ps);
                        }
                        ph[mf] = __builtin_bit_cast(f16x8, hq);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) ps += p[r];
                    split8(p, ph[0], pl[0]);
                    split8(p + 8, ph[1], pl[1]);
                }
                float x0, x1;
                half_swap(ps, x0, x1);
                psum = x0 + x1;
            };
            const bool first = kt == kt_begin;       // (uniform) a wave's first tile: m_run is still the placeholder 0
            if (!first) soft(0.f, std::false_type{});      // (a first tile goes straight to the slow path: with a key split a wave has ~6 tiles)
            if (first || __any(!(psum <= ATT_SM_THR))) {
                // slow path (a wave's first tile; afterwards only when a row's logits outgrow its reference maximum by 9 ... 14 in log2 units):
                // the reference maximum of the rows concerned moves to this tile's maximum
                float tmax = sm[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sm[r]);
                {
                    float x0, x1;
                    half_swap(tmax, x0, x1);
                    tmax = fmaxf(x0, x1);
                }
                const float delta = (first || !(psum <= ATT_SM_THR)) ? tmax : 0.f;       // m_new - m_run
                // (a wave's first tile: O and l are still zero and delta is measured from the placeholder 0 - it may be -1 000 with
                //  logits that large, 2^1000 = inf, and 0 x inf would poison the row: no rescale there - and 64 multiplications of zeros
                //  less in a tile that is a sixth of a wave's work when the key range is split)
                if (!first) {
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) ot[n][r] *= alpha;
                    l_run *= alpha;
                }
                m_run += delta;
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[r] = -m_run;
                soft(delta, std::true_type{});
            }
            l_run += psum;
        } else {
        if (!(abl & 2)) {
        if (kt == ntiles_all - 1) {              // only the last tile can hold keys past S
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kt * KT + frag_row(r, hi) >= S) sm[r] = -INFINITY;
        }
        float tmax = sm[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sm[r]);
        {
            float x0, x1;
            half_swap(tmax, x0, x1);
            tmax = fmaxf(x0, x1);
        }
        const float m_new = att_lazy_max(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        // the reference max of most rows stops moving after the first tiles: alpha == 1 exactly, skip the O rescale
        const bool rescale = !__all(m_new == m_run);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sm[r] = __builtin_amdgcn_exp2f(sm[r] - m_new);
            psum += sm[r];
        }
        {
            float x0, x1;
            half_swap(psum, x0, x1);
            psum = x0 + x1;
        }
        l_run = fmaf(l_run, alpha, psum);
        m_run = m_new;
        if (rescale) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[n][r] *= alpha;
        }
        }
        if (P1) {       // one fp16 plane of P, rounded to nearest
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                u32x4 hq;
#pragma unroll
                for (int i = 0; i < 4; ++i) hq[i] = pk_f16_rne(sm[8 * mf + 2 * i], sm[8 * mf + 2 * i + 1]);
                ph[mf] = __builtin_bit_cast(f16x8, hq);
            }
        } else {
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[r] = sm[r];
            split8(pv, ph[0], pl[0]);
            split8(pv + 8, ph[1], pl[1]);
        }
        }
        ATT_STAMP(5)
        if (PF && P1 && X2) {
            // step i = 4 mf + n: the 16-key half mf outermost, so that the two matrix instructions of an accumulator are four apart instead of
            // back to back (each accumulator still sees mf = 0, then mf = 1: the same bits)
            f16x8 vf[2 * NT];
#pragma unroll
            for (int i = 0; i < PFD; ++i) vf[i] = vpre[i];
#pragma unroll
            for (int i = 0; i < 2 * NT; ++i) {
                if (IA >= 2) vf[i] = vpre[i];
                else if (i + PFD < 2 * NT) vf[i + PFD] = *reinterpret_cast<const f16x8*>(Vh + ((i + PFD) & 3) * 1024 + vbase[(i + PFD) >> 2]);
                ot[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[i], ph[i >> 2], ot[i & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
        if (!(abl & 4)) {
            // same pipelining for the V^T fragments: step = (n, mf), 8 steps of three MFMAs
            auto vload = [&](int step, f16x8& vh, f16x8& vl) {
                const int n = step >> 1, mf = step & 1;
                vh = *reinterpret_cast<const f16x8*>(Vh + n * 1024 + vbase[mf]);
                if (!X2) vl = *reinterpret_cast<const f16x8*>(Vl + n * 1024 + vbase[mf]);
            };
            f16x8 vh_c, vl_c;
            vload(0, vh_c, vl_c);
#pragma unroll
            for (int step = 0; step < 2 * NT; ++step) {
                f16x8 vh_n = vh_c, vl_n = vl_c;
                if (step + 1 < 2 * NT) vload(step + 1, vh_n, vl_n);
                const int n = step >> 1, mf = step & 1;
                ot[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh_c, ph[mf], ot[n], 0, 0, 0);
                if (!P1) ot[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh_c, pl[mf], ot[n], 0, 0, 0);
                if (!X2) ot[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl_c, ph[mf], ot[n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                vh_c = vh_n;
                vl_c = vl_n;
            }
        }
        ATT_STAMP(6)
    };
    {
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        int kt = kt_begin;
        for (; kt + 2 < ntiles; kt += 2) {
            tile_body(kt, S0{}, std::true_type{});
            tile_body(kt + 1, S1{}, std::true_type{});
        }
        if (kt + 1 < ntiles) {
            tile_body(kt, S0{}, std::true_type{});
            tile_body(kt + 1, S1{}, std::false_type{});
        } else if (kt < ntiles) {      // (an empty key range - more splits than key tiles, a forced "attn_nsplit" - runs no tile)
            tile_body(kt, S0{}, std::false_type{});
        }
    }
    }
    if (TRACE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            unsigned long long* t = trace + ((size_t)blockIdx.x * 4 + wid) * 12;
            t[10] = rt0;
            t[11] = __builtin_amdgcn_s_memrealtime();
            for (int i = 0; i < 7; ++i) t[i] = tacc[i];
            t[9] = tacc[7];
            t[7] = tstart;
            t[8] = tprev;
        }
    }

    if (a.nsplit > 1) {
        if (q < S) {
            const size_t Mtot = (size_t)(a.nseq ? a.nseq : gridDim.x / (nqt * a.nhead * a.nsplit)) * S;   // nseq * S
            const size_t tok = tok0 + q;
            float* op = a.Opart + ((size_t)split * Mtot + tok) * d + h * HD;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int c0 = n * 32 + 8 * r4 + 4 * hi;
                    *reinterpret_cast<f32x4*>(op + c0) =
                        f32x4{ot[n][4 * r4 + 0], ot[n][4 * r4 + 1], ot[n][4 * r4 + 2], ot[n][4 * r4 + 3]};
                }
            if (hi == 0) {
                float* ml = a.MLpart + (((size_t)split * Mtot + tok) * a.nhead + h) * 2;
                ml[0] = m_run;
                ml[1] = l_run;
            }
        }
        return;
    }
    if (q < S) {
        const float inv = 1.0f / l_run;
        const int orow = (int)tok0 + q;
        bool overflow = false;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int c0 = n * 32 + 8 * r4 + 4 * hi;
                f16x4 vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = ot[n][4 * r4 + e] * inv;
                    half_t hh, ll;
                    split_f32(v, hh, ll);
                    overflow |= !(fabsf(v) <= kHalfMax);
                    vh[e] = hh;
                    vl[e] = ll;
                }
                const size_t ob = blk_index(orow, h * HD + c0, d);
                *reinterpret_cast<f16x4*>(a.Ohi + ob) = vh;
                if (!X2) *reinterpret_cast<f16x4*>(a.Olo + ob) = vl;   // F16X2: out_proj reads O_hi only
            }
        }
        if (overflow) atomicOr(a.range_flag, 1);
    }
}

// merge the nsplit partial results of the split-KV launch: O = sum_i 2^(m_i - M) O_i / sum_i 2^(m_i - M) l_i
static __global__ __launch_bounds__(256) void attn_combine_kernel(AttnHArgs a, size_t Mtot, int HD) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    args_now_each(a, Mtot, HD);
    const int d = a.d, d4 = d >> 2;
    const size_t total = Mtot * d4;
    // index arithmetic in 32 bits and shifts when d / 4 is a power of two (it is: 128 at d_model 512); the size_t divisions were
    // 250 instructions in front of the first load of a kernel whose body is 3 us
    const bool fast = (d4 & (d4 - 1)) == 0 && total < (1ull << 31);
    const int sh4 = __builtin_ctz((unsigned)d4);
    bool overflow = false;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t tok = fast ? (size_t)((unsigned)idx >> sh4) : idx / d4;
        const int c = (fast ? (int)((unsigned)idx & (unsigned)(d4 - 1)) : (int)(idx % d4)) * 4, h = HD == 128 ? c >> 7 : c / HD;
        float M = -INFINITY;
        float L = 0.f;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (a.nsplit <= 8) {
            // every partial requested before the first is used: one memory round trip instead of two (the same operations in the
            // same order as the loops below)
            f32x2_ ml[8];
            f32x4 p[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < a.nsplit) {
                    ml[i] = *reinterpret_cast<const f32x2_*>(a.MLpart + (((size_t)i * Mtot + tok) * a.nhead + h) * 2);
                    p[i] = *reinterpret_cast<const f32x4*>(a.Opart + ((size_t)i * Mtot + tok) * d + c);
                }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < a.nsplit) M = fmaxf(M, ml[i][0]);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < a.nsplit) {
                    const float w = __builtin_amdgcn_exp2f(ml[i][0] - M);
                    L = fmaf(w, ml[i][1], L);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaf(w, p[i][e], o[e]);
                }
        } else {
            for (int i = 0; i < a.nsplit; ++i) M = fmaxf(M, a.MLpart[(((size_t)i * Mtot + tok) * a.nhead + h) * 2]);
            for (int i = 0; i < a.nsplit; ++i) {
                const float* ml = a.MLpart + (((size_t)i * Mtot + tok) * a.nhead + h) * 2;
                const float w = __builtin_amdgcn_exp2f(ml[0] - M);
                L = fmaf(w, ml[1], L);
                const f32x4 p = *reinterpret_cast<const f32x4*>(a.Opart + ((size_t)i * Mtot + tok) * d + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf(w, p[e], o[e]);
            }
        }
        const float inv = 1.0f / L;
        f16x4 vh, vl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = o[e] * inv;
            half_t hh, ll;
            split_f32(v, hh, ll);
            overflow |= !(fabsf(v) <= kHalfMax);
            vh[e] = hh;
            vl[e] = ll;
        }
        const size_t ob = blk_index((int)tok, c, d);
        *reinterpret_cast<f16x4*>(a.Ohi + ob) = vh;
        if (!a.x2) *reinterpret_cast<f16x4*>(a.Olo + ob) = vl;   // F16X2: out_proj reads O_hi only
    }
    if (overflow) atomicOr(a.range_flag, 1);
}

// number of key-range splits for a launch with `base_blocks` workgroups.
//  * sequences of fewer than 128 key tiles (S < 4096), or at most one workgroup per CU: the largest split count that still
//    leaves ONE workgroup per CU (<= 256 workgroups) with >= 4 key tiles each - one scene (40 base workgroups) gets
//    6 splits = 240 workgroups: 12.11 ms per 50-step call against 12.12 (5), 12.25 (4), 12.84 (7), 12.90 (8); a 2-scene
//    launch (80) 3, a 4-scene launch (160) and anything larger none: with 38 tiles per workgroup the partial outputs and
//    the combine pass cost more than a better filled last round wins (8 episodes as 2 x 4: 28.3 ms unsplit, 30.0 with 3
//    splits; 16 as 2 x 8: 46.7 against 54.0; tools/single_scene_sweep.py attn_nsplit=... f16mx E);
//  * long sequences on more than 256 workgroups: the smallest split count (<= 8, >= 4 tiles each) that
//    fills the last round of the 512 resident workgroups to >= 90 % - one dense scene (N=25, K=64: 600 workgroups of 600
//    tiles = 1.17 rounds, 59 % efficient) becomes 2400 workgroups = 4.7 rounds (94 %).
inline int attn_pick_nsplit(int base_blocks, int S) {
    const int ntiles = (S + 31) / 32;
    if (base_blocks <= 256 || ntiles < 128) {
        int ns = 1;
        while (ns < 8 && base_blocks * (ns + 1) <= 256 && ntiles / (ns + 1) >= 4) ++ns;
        return ns;
    }
    auto eff = [&](int ns) {
        const long w = (long)base_blocks * ns;
        return (double)w / (double)(((w + 511) / 512) * 512);
    };
    int best = 1;
    for (int ns = 1; ns <= 8 && ntiles / ns >= 4; ++ns) {
        if (eff(ns) >= 0.9) return ns;
        if (eff(ns) > eff(best) + 1e-9) best = ns;
    }
    return eff(best) > eff(1) + 0.1 ? best : 1;
}


// one instantiation of the LDS-DMA kernel: its dynamic-LDS attribute once per device, then the launch
template <bool X2, bool MX, bool PIPE, bool P1, bool PF = false, int SM = 1>
inline void launch_attn_dma(const AttnHArgs& a, dim3 grid, int nqt, hipStream_t st) {
#ifdef JMID_DIAGNOSTICS
    if (SM == 1 && tune().attn_sm == 2) return launch_attn_dma<X2, MX, PIPE, P1, PF, 0>(a, grid, nqt, st);      // A/B: the round 2-5 softmax
#endif
    static DevSeen seen;
    const auto kern = &attn_f16x3_dma_kernel<false, X2, MX, PIPE, P1, PF, SM>;
    if (auto once_ = first_use_on_device(seen))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, grid, dim3(256), tune().attn_one_wg ? 160 * 1024 : ATT_DMA_LDS, st, a, nqt, PIPE ? 0 : attn_abl_bits(), (unsigned long long*)nullptr);
}

// (An 8-wave ping-pong form of this kernel - two wave groups per SIMD alternating between a matrix-instruction segment and the softmax across
//  s_barrier - was built three times: rounds 4-5 with the ~100-instruction softmax, round 6 with the SM = 1 softmax, whose ~38 vector
//  instructions fit inside the partner's matrix segment.  Bit-identical every time, and 7 % SLOWER than two free-running 4-wave workgroups
//  per CU even then: profiles/r06_attn_pp_check.log, docs/NOTEBOOK.md section 11.  The kernel is not in the tree.)
inline hipError_t launch_attn_f16x3(const AttnHArgs& a_in, int nseq, int head_dim, hipStream_t st) {
    AttnHArgs a = a_in;
    if (head_dim == 128 && tune().attn_h_variant != 1) {
        const int nqt = (a.S + 127) / 128;
        const dim3 grid1(nqt * a.nhead * nseq * a.nsplit);
        const unsigned long long x_max = std::max<unsigned long long>(grid1.x, (unsigned long long)a.nsplit * ((a.S + 31) / 32));
        a.mq = fast_div_magic(nqt, x_max);
        a.ms = fast_div_magic(a.nsplit, x_max);
        a.mh = fast_div_magic(a.nhead, x_max);
        a.nseq = nseq;
        a.prio = tune().attn_prio;
        // the mode is a template parameter (a run-time flag in the key-tile loop costs F16X3 ~4 %).  F16X2 / F16MX: one fp16 plane
        // of P unless "attn_mx" = 1; F16MX with bf8 K images: the logits' correction terms as bf8 MFMAs
        const bool p1 = tune().attn_mx != 1;
        if (a.x2 && a.K8h) {
            if (p1 && tune().attn_pf != 2) launch_attn_dma<true, true, false, true, true>(a, grid1, nqt, st);
            else if (p1) launch_attn_dma<true, true, false, true>(a, grid1, nqt, st);
            else launch_attn_dma<true, true, false, false>(a, grid1, nqt, st);
        } else if (a.x2) {
            if (p1 && tune().attn_pf != 2) launch_attn_dma<true, false, false, true, true>(a, grid1, nqt, st);
            else if (p1) launch_attn_dma<true, false, false, true>(a, grid1, nqt, st);
            else launch_attn_dma<true, false, false, false>(a, grid1, nqt, st);
        } else {
            launch_attn_dma<false, false, false, false>(a, grid1, nqt, st);
        }
        if (a.nsplit > 1 && !a.skip_combine) {
            const size_t Mtot = (size_t)nseq * a.S;
            const int blocks = (int)std::min<size_t>((Mtot * (a.d / 4) + 255) / 256, 2048);
            hipLaunchKernelGGL(attn_combine_kernel, dim3(blocks), dim3(256), bystander_lds(attn_combine_kernel), st, a, Mtot, 128);
        }
        return hipGetLastError();
    }
    dim3 grid((a.S + 127) / 128, a.nhead, nseq);
    switch (head_dim) {
        case 16: hipLaunchKernelGGL((attn_f16x3_kernel<16>), grid, dim3(256), 0, st, a); break;
        case 32: hipLaunchKernelGGL((attn_f16x3_kernel<32>), grid, dim3(256), 0, st, a); break;
        case 64: hipLaunchKernelGGL((attn_f16x3_kernel<64>), grid, dim3(256), 0, st, a); break;
        case 128: hipLaunchKernelGGL((attn_f16x3_kernel<128>), grid, dim3(256), 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace jmid
