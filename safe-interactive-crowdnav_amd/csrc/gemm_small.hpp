// Split-fp16 GEMMs for launches of AT MOST ONE workgroup per CU - the shape the MPC issues (one scene: M = 1200 tokens) and
// its neighbours (2 ... 4 scenes per chunk).  Included by gemm_f16x3.hpp (same operands, same epilogues, same bits).
//
// What bounds such a launch is not arithmetic (a 64 x 64 tile of a K = 512 product is 40 MFMAs per wave) but how the operands get
// to the CU (rocprofv3 PMC on the round-3 kernels, tools/small_pmc.sh: every launch fetches its weight 8 times, once per XCD,
// from the Infinity Cache - in_proj 20.3 MB for 2.4 MB of W - and a K loop of 16 k32 tiles with three 10 KB tiles in flight
// advances at one Infinity-Cache round trip per three tiles: 3.8 us per 512 of K, 0.24 us per tile):
//   * the ring holds k64 stages (half the barriers, wait / read / MFMA chains twice as long) and is as deep as the CU's LDS
//     allows with ONE workgroup per CU (up to 140 KB: six stages = 120 KB of a 160 KB operand set requested before the first
//     MFMA) - the K loop then runs at the LDS-DMA rate of the CU instead of at the latency of the Infinity Cache;
//   * the tile order is two-dimensional per XCD: the tile sequence is cut into `pn` column groups, each walked M-major, and XCD
//     x takes the x-th eighth of the sequence, so an XCD fetches |W| / pn + pn |A| / 8 instead of |W| + |A| / 8 - the host picks
//     the pn that minimises the bytes all eight XCDs pull (in_proj 20.1 -> 9.6 MB, linear1 13.8 -> 8.0, linear2 15.0 -> 11.2);
//   * per k64 block and accumulator the MFMA sequence is the one every other tile shape of the mode issues (F16MX: four fp16 steps
//     in k order, then the fp8 instruction; F16X2 / F16X3: per k16 step hi.hi, hi.lo, [lo.hi]) - results are bit-identical to the
//     large-tile kernels, so an episode's result still does not depend on the size of its batch.
// Tile 64 x 32 WC (WC = 2: 4 waves, N <= 512; WC = 4: 8 waves, in_proj / linear1), each wave one 32 x 32 accumulator.
#pragma once
#include "elementwise.hpp"
#include "gemm_ln2_mx.hpp"

namespace jmid {

enum SmallMode { SM_X3 = 0, SM_X2 = 1, SM_MX = 2 };

// tools/small_gemm_trace.hip only: wall-clock stamps (s_memrealtime, 100 MHz) of thread 0 of every workgroup at the phase boundaries
#ifdef JMID_SMALL_TRACE
__device__ unsigned long long* g_small_trace;
#define SM_STAMP(i)                                                                                      \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (threadIdx.x == 0) { sm_trace_p[i] = __builtin_amdgcn_s_memrealtime(); sm_trace_p[8 + (i)] = __builtin_amdgcn_s_memtime(); } \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    } while (0)
#ifndef JMID_SMALL_ABL
#define JMID_SMALL_ABL 0
#endif
#define SM_ABL(bit) ((JMID_SMALL_ABL & (bit)) != 0)      // compile-time ablations: 16 no copies, 32 no fp16 MFMAs, 64 the launch alone
#define SM_TRACE_ARG , sm_trace_p
#else
#define SM_STAMP(i)
#define SM_ABL(bit) false
#define SM_TRACE_ARG
#endif


template <int MODE, int WC, bool TWO = false>     // TWO: two workgroups per CU (launches of 257 ... 512 tiles): half the LDS budget each
struct SmCfg {
    static constexpr int NW = 2 * WC, NT = 64 * NW, BM = 64, BN = 32 * WC;
    static constexpr int A_SUB = BM * 64, W_SUB = BN * 64;                  // bytes of one k32 sub-tile of a plane
    static constexpr int A_PLANES = MODE == SM_X3 ? 2 : 1, W_PLANES = MODE == SM_MX ? 1 : 2;
    static constexpr int BLK = 2 * (A_PLANES * A_SUB + W_PLANES * W_SUB) + (MODE == SM_MX ? BN * 64 : 0);     // one k64 block, all planes
    // k64 blocks per ring stage: a stage costs a wave one wait + barrier + LDS round trip however deep it is (the wave is alone
    // on its SIMD: nothing hides them), so stages are k128 wherever TWO of them fit the LDS budget (three: one scene 11.92 vs
    // 11.71 ms in f16mx, 13.78 vs 13.66 in f16x3; -DJMID_SMALL_KB_STAGES=3 for the A/B)
    static constexpr int BUDGET = TWO ? 80 * 1024 : 144 * 1024;
#ifndef JMID_SMALL_KB_STAGES
#define JMID_SMALL_KB_STAGES 2
#endif
    static constexpr int KB = (TWO ? 2 : JMID_SMALL_KB_STAGES) * 2 * BLK <= BUDGET ? 2 : 1;
    static constexpr int A_BYTES = KB * 2 * A_SUB, W_BYTES = KB * 2 * W_SUB;          // one plane of a stage
    static constexpr int W8_BLOCK = MODE == SM_MX ? BN * 64 : 0;                    // bf8(W_lo) of one k64 block
    static constexpr int OFF_AH = 0, OFF_AL = A_BYTES, OFF_WH = A_PLANES * A_BYTES, OFF_WL = OFF_WH + W_BYTES;
    static constexpr int OFF_W8 = OFF_WH + W_PLANES * W_BYTES;
    static constexpr int STAGE = OFF_W8 + KB * W8_BLOCK;
    static constexpr int ROUND = NT * 16;                                   // bytes one DMA wave-instruction per wave moves
    static constexpr int RA = A_BYTES / ROUND, RW = W_BYTES / ROUND, R8B = W8_BLOCK / ROUND, R8 = KB * R8B;
    static constexpr int NR = A_PLANES * RA + W_PLANES * RW + R8;           // DMA wave-instructions per wave and stage
    static constexpr int NS_MAX = BUDGET / STAGE;
    static constexpr int NS = NS_MAX < 4 ? NS_MAX : 4;                      // ring slots; NS - 1 stages of look-ahead
    static constexpr int L = NS - 1;
    static constexpr size_t LDS_BYTES = size_t(NS) * STAGE;
    static_assert(A_BYTES % ROUND == 0 && W_BYTES % ROUND == 0 && W8_BLOCK % ROUND == 0, "plane / workgroup mismatch");
    static_assert(NS >= 2 && (L - 1) * NR <= 63, "ring depth against the vmcnt range");
};

__device__ __forceinline__ bool tune_small_qk(int flags) { return (flags & 16) == 0; }      // ("small_qk" = 2: the generic epilogue, A/B)
constexpr int SM_STG_LD = 68;                              // floats per row of a staged 64 x 64 tile (272 B: conflict-free b128)

// ---- OUT_LNX: out_proj / linear2 + residual + LayerNorm in ONE small launch WITHOUT concentrating the row-wise work: the eight
// workgroups of a 64-row tile (N = 512 = 8 column tiles of 64) each keep their own 64 x 64 block, exchange only the ROW STATISTICS -
// the block's partial sum P_c, then its partial sum of squared deviations - and normalise and store their own columns.  What goes
// between workgroups is 64 granules of 8 bytes per block and statistic: {fp32 partial, launch tag} written and polled as ONE relaxed
// agent-scope 64-bit atomic each (global_store / global_load_dwordx2 sc1: the value and its validity arrive together, no fence, no
// flag, no counter); a workgroup waits for the seven other blocks of its rows only, which were dispatched next to it.  The partials
// and the totals are formed in gemm_ln2_mx.hpp's canonical order (partial(c, h) by ONE thread over its 32 columns, P_c = partial(c, 0)
// + partial(c, 1), total = ((P0 + P1) + ...) + P7), so the rows are bit-identical to the GEMM + add_ln2_kernel pair and to the
// row-complete kernel of full launches.  The launch must fit the chip (the waiting workgroups need their partners resident, nothing
// else in flight on the handle): <= 256 tiles with one workgroup per CU, 257 ... 512 (33 ... 64 row tiles: two cfg2 scenes, the
// reference's shipped N = 3, K = 100, H = 8) with TWO per CU on half the LDS each (SmCfg<SM_MX, 2, true>: two ring slots of k128
// stages; the shipped point 0.728 -> 0.711 ms per call, 2 400 tokens x 50 steps 14.75 -> 14.24).  Either way every workgroup of
// the launch is resident at once (the blocks of a row tile also sit at the same position of their XCDs' dispatch ranges, so a tile in
// front of an incomplete one is complete and finishes).  The polls are BOUNDED all the same - a workgroup that does not see a
// partner within ~10^5 polls sets bit 1 of the range flag and leaves, and the host drops the fused path for the handle.
constexpr int SM_LNX_POLLS = 1 << 17;
constexpr int SM_LNX_MAX_TILES = 64;                            // row tiles of a launch: 32 with one workgroup per CU, 33 ... 64 with two
constexpr size_t SM_LNX_STATS = size_t(2) * SM_LNX_MAX_TILES * 8 * 64;        // per step workspace: two statistics x row tiles x 8 blocks x 64 rows ...
constexpr size_t SM_LNX_GRANULES = SM_LNX_STATS + SM_LNX_MAX_TILES * 8;       // ... + one flag per block of the A operand (lnx_combine)

__device__ __forceinline__ void lnx_publish(unsigned long long* slot, float v, unsigned tag) {
    __hip_atomic_store(slot, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the seven other blocks' partials of this workgroup's 64 rows -> red[row * 8 + block] (sums) and red[512 + row * 8 + block] (squared
// deviations), the granules of both statistics in one sweep; thread t polls blocks 2 (t & 3), + 1 of row t >> 2.  `polls`: the budget
// (0 = SM_LNX_POLLS); after every 256th unsuccessful poll a thread looks at the range flag and leaves when a workgroup of this call has
// already given up (bit 1): the first timeout of a call costs the whole budget, the launches behind it next to nothing.
__device__ __forceinline__ bool lnx_timed_out(const int* range_flag) {
    return (__hip_atomic_load(range_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) != 0;
}
__device__ __forceinline__ bool lnx_gather(const unsigned long long* slots_s, const unsigned long long* slots_q, float* red, int own,
                                           unsigned tag, int tid, int polls, const int* range_flag) {
    const int r = tid >> 2, k = tid & 3;
    const unsigned long long* p0 = slots_s + (2 * k) * 64 + r;
    const unsigned long long* p1 = p0 + 64;
    const unsigned long long* q0 = slots_q + (2 * k) * 64 + r;
    const unsigned long long* q1 = q0 + 64;
    float* d0 = red + r * 8 + 2 * k;
    bool n0 = 2 * k != own, n1 = 2 * k + 1 != own, m0 = n0, m1 = n1;
    int budget = polls > 0 ? polls : SM_LNX_POLLS, misses = 0;
    while ((n0 || n1 || m0 || m1) && budget > 0) {
        unsigned long long g0 = 0, g1 = 0, h0 = 0, h1 = 0;
        if (n0) g0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n1) g1 = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m0) h0 = __hip_atomic_load(q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m1) h1 = __hip_atomic_load(q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n0 && (unsigned)(g0 >> 32) == tag) { d0[0] = __uint_as_float((unsigned)g0); n0 = false; }
        if (n1 && (unsigned)(g1 >> 32) == tag) { d0[1] = __uint_as_float((unsigned)g1); n1 = false; }
        if (m0 && (unsigned)(h0 >> 32) == tag) { d0[512] = __uint_as_float((unsigned)h0); m0 = false; }
        if (m1 && (unsigned)(h1 >> 32) == tag) { d0[513] = __uint_as_float((unsigned)h1); m1 = false; }
        if (n0 || n1 || m0 || m1) {
            __builtin_amdgcn_s_sleep(1);
            if ((++misses & 255) == 0 && lnx_timed_out(range_flag)) break;      // (never on the first misses: the look costs a memory round trip)
        }
        --budget;
    }
    return !(n0 || n1 || m0 || m1);
}

// The split-KV merge in front of the out-projection's K loop (GemmHArgs::cmb_*): the workgroup of tile (tm, c) merges rows m0 .. m0 + 63,
// columns 64 c .. 64 c + 63 of the attention output - attn_combine_kernel's operations in its order: the same bits - and writes the fp16
// plane the K loops read (write-through), then tells the seven other workgroups of its row tile and waits for theirs: a flag granule
// {launch tag} per block behind the statistics granules.  The L2 of an XCD holds no line of the plane when the launch begins (a kernel
// boundary invalidates it) and nobody reads the plane before the flags are up, so the K loop's ordinary copies see the merged rows.
// head_dim 128, at most 8 partials, F16X2 / F16MX (one plane of A).
__device__ __forceinline__ void lnx_combine(const GemmHArgs& g, int tm, int c, int m0, int tid) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const int d = g.K, ns = g.cmb_ns;
    const size_t Mtot = g.cmb_Mtot;
    half_t* Ohi = const_cast<half_t*>(g.Ahi);
    f32x2_ ml[4][8];
    f32x4 p[4][8];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = it * 256 + tid, r = idx >> 4, col = c * 64 + (idx & 15) * 4, h = col >> 7;
        const int tok = m0 + r < g.M ? m0 + r : g.M - 1;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < ns) {
                ml[it][i] = *reinterpret_cast<const f32x2_*>(g.cmb_ML + (((size_t)i * Mtot + tok) * g.cmb_nhead + h) * 2);
                p[it][i] = *reinterpret_cast<const f32x4*>(g.cmb_O + ((size_t)i * Mtot + tok) * d + col);
            }
    }
    bool overflow = false;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = it * 256 + tid, r = idx >> 4, col = c * 64 + (idx & 15) * 4;
        float M = -INFINITY, L = 0.f;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < ns) M = fmaxf(M, ml[it][i][0]);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < ns) {
                const float w = __builtin_amdgcn_exp2f(ml[it][i][0] - M);
                L = fmaf(w, ml[it][i][1], L);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf(w, p[it][i][e], o[e]);
            }
        const float inv = 1.0f / L;
        f16x4 vh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = o[e] * inv;
            half_t hh, ll;
            split_f32(v, hh, ll);
            overflow |= m0 + r < g.M && !(fabsf(v) <= kHalfMax);
            vh[e] = hh;
        }
        if (m0 + r < g.M)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(Ohi + blk_index(m0 + r, col, d)), __builtin_bit_cast(unsigned long long, vh),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (overflow) atomicOr(g.range_flag, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's part of the block is out ...
    __syncthreads();                                       // ... and the workgroup's
    unsigned long long* flags = g.ln_xchg + SM_LNX_STATS + (size_t)tm * 8;
    if (tid == 0) __hip_atomic_store(flags + c, (unsigned long long)g.ln_epoch << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 8 && tid != c) {
        int budget = g.ln_polls > 0 ? g.ln_polls : SM_LNX_POLLS, misses = 0;
        bool need = true;
        while (need && budget > 0) {
            need = (unsigned)(__hip_atomic_load(flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) != g.ln_epoch;
            if (need) {
                __builtin_amdgcn_s_sleep(1);
                if ((++misses & 255) == 0 && lnx_timed_out(g.range_flag)) break;
            }
            --budget;
        }
        if (need) atomicOr(g.range_flag, 2);               // a partner never showed up: the call is repeated without this kernel
    }
    __syncthreads();
}

// what the tail wants from memory that does not depend on the product - the residual rows (planes the PREVIOUS launch wrote: an
// Infinity-Cache round trip) and gamma / beta of the thread's 32 columns - requested before the K loop by the waves that will use them
struct LnxPre {
    f16x8 xh[4];
    i32x2 xb[4];
    f32x4 gm[8], bt[8];
};
// (Inline assembly, so that they stay where they are and their registers are not touched before lnx_prefetch_landed().  The loads are
//  OLDER than every copy of the ring: the ring's counted waits cover them.)
__device__ __forceinline__ void lnx_prefetch(const GemmHArgs& g, int c, int m0, int tid, LnxPre& pre) {
    constexpr int d = GLN_BN;
    const int row = (tid >> 1) & 63, h = tid & 1;
    const int grow = m0 + row, rowc = grow < g.M ? grow : g.M - 1;
    if (tid < 128) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c0 = c * 64 + (u >> 1) * 32 + (u & 1) * 16 + h * 8;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pre.xh[u]) : "v"(g.ln_xh + blk_index(rowc, c0, d)) : "memory");
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pre.xb[u]) : "v"(g.ln_xl8 + blk8_index(rowc, c0, d)) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pre.gm[2 * u]) : "v"(g.ln_gamma + c0) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pre.gm[2 * u + 1]) : "v"(g.ln_gamma + c0 + 4) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pre.bt[2 * u]) : "v"(g.ln_beta + c0) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pre.bt[2 * u + 1]) : "v"(g.ln_beta + c0 + 4) : "memory");
        }
    }
}
__device__ __forceinline__ void lnx_prefetch_landed(LnxPre& pre) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(pre.xh[u]), "+v"(pre.xb[u]), "+v"(pre.gm[2 * u]), "+v"(pre.gm[2 * u + 1]), "+v"(pre.bt[2 * u]), "+v"(pre.bt[2 * u + 1]));
}

// stg: the workgroup's staged tile [64][SM_STG_LD] (accumulator + bias); red: [2][64][8] floats behind it
// ONE exchange: every block publishes its sum S_c AND the squared deviations Q_c from ITS OWN mean, and the row variance is merged from
// the eight (count, mean, M2) triples (Chan et al.) - gemm_ln2_mx.hpp's canonical order since round 6 (its header), which the batch
// kernels and add_ln2_kernel form the same way: the rows are bit-identical to theirs.  (Round 5 shipped TWO exchanges - sums, then squared
// deviations from the row mean, the canonical order of that round - and kept this form as a diagnostics knob: one memory round trip less
// per LayerNorm, one scene 10.47 against 10.67 ms per call.)
__device__ __forceinline__ void lnx_tail_mx(const GemmHArgs& g, const LnxPre& pre, const float* stg, float* red, int tm, int c, int m0, int tid) {
    constexpr int d = GLN_BN;
    const bool owner = tid < 128;                      // waves 0 and 1: thread (row, h) owns the 32 columns of s(c, h)
    const int row = (tid >> 1) & 63, h = tid & 1;
    const int grow = m0 + row;
    const int ntm = (g.M + 63) / 64;
    unsigned long long* slots_s = g.ln_xchg + ((size_t)tm * 8) * 64;                    // [block][row] of this row tile: sums ...
    unsigned long long* slots_q = g.ln_xchg + ((size_t)(ntm + tm) * 8) * 64;            // ... and squared deviations
    float v[32];
    float s = 0.f;
    if (owner) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cl = (u >> 1) * 32 + (u & 1) * 16 + h * 8;
            const f32x4 y0 = *reinterpret_cast<const f32x4*>(stg + row * SM_STG_LD + cl), y1 = *reinterpret_cast<const f32x4*>(stg + row * SM_STG_LD + cl + 4);
            float xl[8];
            f32_of_bf8x8(pre.xb[u], xl);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = (float)pre.xh[u][e] + xl[e];
                const float t = a + (e < 4 ? y0[e] : y1[e - 4]);
                v[u * 8 + e] = t;
                s += t;
            }
        }
        s += __shfl_xor(s, 1, 64);
        const float mc = s / 64.f;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const float t = v[e] - mc;
            q = fmaf(t, t, q);
        }
        q += __shfl_xor(q, 1, 64);
#ifdef JMID_DIAGNOSTICS
        const bool withheld = g.ln_withhold && tm == 0 && c == 7;      // (tests: a partner that never shows up)
#else
        constexpr bool withheld = false;
#endif
        if (h == 0) {
            if (!withheld) {
                lnx_publish(slots_s + c * 64 + row, s, g.ln_epoch);
                lnx_publish(slots_q + c * 64 + row, q, g.ln_epoch);
            }
            red[row * 8 + c] = s;
            red[512 + row * 8 + c] = q;
        }
    }
    const bool ok = lnx_gather(slots_s, slots_q, red, c, g.ln_epoch, tid, g.ln_polls, g.range_flag);
    __syncthreads();
    auto row_total = [&](const float* r8p) {
        float t = r8p[0] + r8p[1];
#pragma unroll
        for (int k = 2; k < 8; ++k) t += r8p[k];
        return t;
    };
    float mean = 0.f, rstd = 0.f;
    if (owner) {
        mean = row_total(red + row * 8) / (float)d;
        float dm = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float t = red[row * 8 + k] / 64.f - mean;
            dm += t * t;
        }
        rstd = rsqrtf((row_total(red + 512 + row * 8) + 64.f * dm) / (float)d + g.ln_eps);
    }
    if (!ok) atomicOr(g.range_flag, 2);                // a partner never showed up: the call is repeated without this kernel
    if (!owner) return;
    const float nmr = -mean * rstd;
    bool overflow = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c0 = c * 64 + (u >> 1) * 32 + (u & 1) * 16 + h * 8;
        const f32x4 g0 = pre.gm[2 * u], g1 = pre.gm[2 * u + 1], t0 = pre.bt[2 * u], t1 = pre.bt[2 * u + 1];
        f16x8 vh, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float o = fmaf(fmaf(v[u * 8 + e], rstd, nmr), e < 4 ? g0[e] : g1[e - 4], e < 4 ? t0[e] : t1[e - 4]);
            half_t hh, ll;
            split_f32(o, hh, ll);
            overflow |= !(fabsf(o) <= kHalfMax);
            vh[e] = hh;
            vl[e] = ll;
        }
        if (grow < g.M) {
            *reinterpret_cast<f16x8*>(g.ln_xh + blk_index(grow, c0, d)) = vh;
            if (!g.ln_no_lo) *reinterpret_cast<i32x2*>(g.ln_xl8 + blk8_index(grow, c0, d)) = bf8x8_of_f16(vl);
        }
    }
    if (overflow && grow < g.M) atomicOr(g.range_flag, 1);
}

// The K loop of one tile (tm, tn): ring primed, then per stage wait + barrier + the mode's canonical MFMA sequence per k64 block.
// SWAP: the transposed product (W fragment first).  Shared by gemm_small_kernel and gemm_small_out_kernel.
template <int MODE, int WC, bool TWO, bool SWAP>
__device__ __forceinline__ f32x16 small_kloop(const GemmHArgs& g, unsigned char* lds_raw, int tm, int tn, int tid
#ifdef JMID_SMALL_TRACE
                                              , unsigned long long* sm_trace_p
#endif
) {
    using C = SmCfg<MODE, WC, TWO>;
    constexpr bool X2 = MODE != SM_X3, MX = MODE == SM_MX;
    constexpr int BN = C::BN, NS = C::NS, L = C::L, KB = C::KB;
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wid / WC, wc = wid % WC;
    const int n0 = tn * BN;
    const int nk = g.K / 32, nst = g.K / (64 * KB);

    // DMA sources of stage 0, one per wave-instruction of a stage ("round"), in the fixed order A_hi [A_lo] W_hi [W_lo] [W8];
    // round q of a plane moves bytes [q ROUND, (q + 1) ROUND) of the plane's stage image = its 2 KB k32 sub-tiles back to back
    constexpr int IAL = C::RA, IWH = C::A_PLANES * C::RA, IWL = IWH + C::RW, I8 = IWH + C::W_PLANES * C::RW;
    // (wave-uniform: a wave's 1 KB slice of a round lies inside one k32 sub-tile; the lane's 16 bytes are at lane * 16 in every one of
    //  them, so a copy is scalar base + ONE 32-bit lane offset - a per-thread 64-bit pointer costs a 64-bit vector add per copy and a
    //  v_readfirstlane for its LDS destination, on a wave that is alone on its SIMD and issues ~one instruction per five cycles)
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);
    unsigned lane_off = (unsigned)lane * 16u;
    const bool a_merged = g.cmb_O != nullptr;      // (uniform: a kernel argument)
    const char* src[C::NR];
    auto plane_src = [&](const void* base, int tile, auto rows_c, int q) {
        constexpr int ROWS = decltype(rows_c)::value, SUB = ROWS * 64;
        static_assert(SUB % 1024 == 0, "a wave's slice of a round stays inside one sub-tile");
        const int o = q * C::ROUND + wid_s * 1024, sub = o / SUB, within = o - sub * SUB;
        const size_t panel = ROWS == 128 ? (size_t)tile * nk : (size_t)(tile >> 1) * nk;
        return reinterpret_cast<const char*>(base) + (panel + sub) * 8192 + (ROWS == 128 ? 0 : (tile & 1) * 4096) + within;
    };
#pragma unroll
    for (int q = 0; q < C::RA; ++q) {
        src[q] = plane_src(g.Ahi, tm, std::integral_constant<int, 64>{}, q);
        if (!X2) src[IAL + q] = plane_src(g.Alo, tm, std::integral_constant<int, 64>{}, q);
    }
#pragma unroll
    for (int q = 0; q < C::RW; ++q) {
        src[IWH + q] = plane_src(g.Whi, tn, std::integral_constant<int, BN>{}, q);
        if (!MX) src[IWL + q] = plane_src(g.Wlo, tn, std::integral_constant<int, BN>{}, q);
    }
    const size_t w8_kstride = (size_t)(g.N / 32) * 2048;
    if (MX) {
#pragma unroll
        for (int q = 0; q < C::R8; ++q)      // block q / R8B of the stage, round q % R8B of its image
            src[I8 + q] = reinterpret_cast<const char*>(g.W8) + (size_t)(q / C::R8B) * w8_kstride + (size_t)(n0 / 32) * 2048 +
                          (q % C::R8B) * C::ROUND + wid_s * 1024;
    }
    // destination of round r inside a stage (compile-time) + this wave's 1 KB slice of the round
    auto dst_of = [&](int r) {
        if (r < IAL) return C::OFF_AH + r * C::ROUND;
        if (r < IWH) return C::OFF_AL + (r - IAL) * C::ROUND;
        if (r < IWH + C::RW) return C::OFF_WH + (r - IWH) * C::ROUND;
        if (!MX) return C::OFF_WL + (r - IWL) * C::ROUND;
        return C::OFF_W8 + (r - I8) * C::ROUND;
    };
    auto issue = [&](int st_idx, int slot) {
        if (SM_ABL(16)) return;
        unsigned char* st = lds_raw + slot * C::STAGE + wid_s * 1024;
        asm volatile("" : "+v"(lane_off));      // (once per stage: hoisted out of the loop as a 64-bit pair it defeats the scalar-base form)
#pragma unroll
        for (int r = 0; r < C::NR; ++r) {
            const bool is8 = MX && r >= I8;
            const unsigned long long u = pin_uniform(reinterpret_cast<unsigned long long>(src[r] + (is8 ? (size_t)st_idx * KB * w8_kstride : (size_t)st_idx * KB * 16384)));
            // the A plane another workgroup of THIS launch has just merged (lnx_combine: sc1 write-through stores, then a flag): read it
            // with sc1 copies - the producer / consumer pair MI355X_MICROARCH.md lists as valid without an acquire fence (a fence is
            // ~1.7 us per launch) - instead of relying on no stale line of the plane having survived the kernel boundary
            if (r < IAL && a_merged)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) char*>(u) + lane_off,
                                                 (__attribute__((address_space(3))) void*)(st + dst_of(r)), 16, 0, 16);
            else
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) char*>(u) + lane_off,
                                                 (__attribute__((address_space(3))) void*)(st + dst_of(r)), 16, 0, 0);
        }
    };
    const int rowA = wr * 32 + l31, rowW = wc * 32 + l31;
    int offA[2], offW[2];               // bytes inside a k32 sub-tile, per 16-deep step
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        offA[ks] = (rowA * 32 + (((ks * 2 + hi) ^ ((rowA >> 2) & 3)) * 8)) * 2;
        offW[ks] = (rowW * 32 + (((ks * 2 + hi) ^ ((rowW >> 2) & 3)) * 8)) * 2;
    }

    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
    for (int t = 0; t < L; ++t)
        if (t < nst) issue(t, t);
    SM_STAMP(1);
    int slot = 0;
    for (int si = 0; si < nst; ++si) {
        if (si + L - 1 < nst) wait_vmcnt<(L - 1) * C::NR>();       // stage si has landed (this wave's share) ...
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                              // ... and everybody else's; the slot of stage si - 1 is free
        __builtin_amdgcn_sched_barrier(0);
#ifdef JMID_SMALL_TRACE
        if (si == 0) SM_STAMP(2);
#endif
        if (si + L < nst) issue(si + L, slot == 0 ? NS - 1 : slot - 1);
        const unsigned char* st = lds_raw + slot * C::STAGE;
#pragma unroll
        for (int j = 0; j < KB; ++j) {          // the k64 blocks of the stage; per block the canonical MFMA sequence of the mode
            f16x8 ah[4], al[4], wh[4], wl[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int sub = 2 * j + (ks >> 1), step = ks & 1;
                ah[ks] = *reinterpret_cast<const f16x8*>(st + C::OFF_AH + sub * C::A_SUB + offA[step]);
                wh[ks] = *reinterpret_cast<const f16x8*>(st + C::OFF_WH + sub * C::W_SUB + offW[step]);
                if (!X2) al[ks] = *reinterpret_cast<const f16x8*>(st + C::OFF_AL + sub * C::A_SUB + offA[step]);
                if (!MX) wl[ks] = *reinterpret_cast<const f16x8*>(st + C::OFF_WL + sub * C::W_SUB + offW[step]);
            }
            i32x8 w8, a8;
            if (MX) {
                const unsigned char* p = st + C::OFF_W8 + j * C::W8_BLOCK + wc * 2048 + lane * 16;
                const i32x4 lo = *reinterpret_cast<const i32x4*>(p);
                const i32x4 up = *reinterpret_cast<const i32x4*>(p + 1024);
                w8 = i32x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (!SM_ABL(32))
                c = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], ah[ks], c, 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], wh[ks], c, 0, 0, 0);
                if (MX) {
                    // the bf8 image of this step's A fragment, in the shadow of the MFMA just issued (pinned: left to itself the
                    // compiler sinks all eight packs in front of the fp8 instruction, onto the wave's critical path)
                    const i32x4 d = __builtin_bit_cast(i32x4, ah[ks]);
                    int p0 = bf8_of_f16x4(d[0], d[1]), p1 = bf8_of_f16x4(d[2], d[3]);
                    asm volatile("" : "+v"(p0), "+v"(p1));
                    a8[ks * 2 + 0] = p0;
                    a8[ks * 2 + 1] = p1;
                } else {
                    c = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks], ah[ks], c, 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], wl[ks], c, 0, 0, 0);
                    if (!X2)
                        c = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], al[ks], c, 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], wh[ks], c, 0, 0, 0);
                }
            }
            if (MX)      // both operands bf8, literal zero scales: the UNSCALED instruction (gemm_f16x3.hpp)
                c = SWAP ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8, a8, c, 1, 1, 0, 0, 0, 0)
                         : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, w8, c, 1, 1, 0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        slot = slot + 1 == NS ? 0 : slot + 1;
    }
    SM_STAMP(3);
    return c;
}

// Q / K tile (64 tokens x one head of 128 columns) of the in_proj GEMM of a small launch, out through LDS in full rows: the
// transposed product leaves a lane with four runs of 4 consecutive columns of ONE token, the workgroup's 8 waves fill one 64 x 128
// fp16 tile (qk_staged_store's swizzle), and the rows leave as 16 bytes per lane, 256 contiguous bytes per token and plane - the
// bf8 images (F16MX) made from the rows on their way out.  The generic epilogue wrote every element with a 2-byte store and every
// image byte with a 1-byte store (48 store instructions per lane of a K tile): the in_proj launch of one scene spent 1.8 us (its
// slowest workgroup 3.0) behind its K loop where linear1 spends 0.65 (tools/small_gemm_trace.hip).  Same values, same bits.
template <bool K8IMG>
__device__ __forceinline__ void small_qk_staged_store(const GemmHArgs& g, const f32x16& acc, int m0, int n0, half_t* tile, int tid) {
    const int lane = tid & 63, wid = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int wr = wid >> 2, wc = wid & 3;
    const int part = n0 / g.d, nn0 = n0 - part * g.d;
    half_t* dst_base[2] = {part == 0 ? g.Chi : g.Khi, part == 0 ? g.Clo : g.Klo};
    const float qs = part == 0 ? g.qscale : 1.0f;
    const bool k8 = K8IMG && part == 1 && g.K8h != nullptr;     // K tile in F16MX: plane 1 is the two bf8 images instead of fp16 K_lo
    const bool q8 = K8IMG && part == 0 && g.Q8l != nullptr;     // Q tile: the bf8 image of Q_lo instead of the fp16 plane
    const int row = wr * 32 + l31;
    auto lds_at = [&](int q) {
        const int c = (8 * wc + 2 * q + hi) ^ ((row & 15) << 1);       // 8-byte chunk of the row, swizzled in 16-byte units
        return reinterpret_cast<f16x4*>(tile + row * 128 + c * 4);
    };
    unsigned am = 0;
    i32x2_s lo_pk[4];
    {
        f32x4 bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = *reinterpret_cast<const f32x4*>(g.bias + n0 + wc * 32 + 8 * q + 4 * hi);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = fmaf(acc[4 * q + e], kWInv, bv[q][e]);
                if (part == 0) v[e] *= qs;
            }
            const Split4 sp = split_f32x4(v[0], v[1], v[2], v[3], am);
            *lds_at(q) = __builtin_bit_cast(f16x4, sp.hi);
            lo_pk[q] = sp.lo;
        }
    }
    if (m0 + row < g.M && split_range_exceeded(am)) atomicOr(g.range_flag, 1);       // (rows past M hold what the padding held)
#pragma unroll
    for (int plane = 0; plane < 2; ++plane) {
        if (plane == 1) {
            __syncthreads();                                   // the hi rows have left the tile
#pragma unroll
            for (int q = 0; q < 4; ++q) *lds_at(q) = __builtin_bit_cast(f16x4, lo_pk[q]);
        }
        __syncthreads();
        half_t* dst = dst_base[plane] + (size_t)m0 * g.d + nn0;
        const bool img = k8 || (q8 && plane == 1);
        unsigned char* dst8 = img ? (k8 ? (plane == 0 ? g.K8h : g.K8l) : g.Q8l) + (size_t)m0 * g.d + nn0 : nullptr;
#pragma unroll
        for (int t = 2 * wid; t < 2 * wid + 2; ++t) {          // 16 groups of 4 rows, two per wave
            const int r = t * 4 + (lane >> 4), u = lane & 15;
            const f16x8 v8 = *reinterpret_cast<const f16x8*>(tile + r * 128 + ((u ^ (r & 15)) << 3));
            if (m0 + r < g.M) {
                if (!((k8 || q8) && plane == 1)) *reinterpret_cast<f16x8*>(dst + (size_t)r * g.d + u * 8) = v8;
                if (img) {
                    const i32x4_e dw = __builtin_bit_cast(i32x4_e, v8);
                    i32x2_e b8;
                    b8[0] = bf8_of_f16x4_e(dw[0], dw[1]);
                    b8[1] = bf8_of_f16x4_e(dw[2], dw[3]);
                    *reinterpret_cast<i32x2_e*>(dst8 + (size_t)r * g.d + u * 8) = b8;
                }
            }
        }
    }
}

// The OUT_LNX launch of one tile (tm, c): [split-KV merge] -> K loop (transposed: a lane owns token row l31 of its wave's block and
// four runs of 4 consecutive columns) -> the fp32 rows (accumulator + bias: what the stand-alone GEMM hands add_ln2) staged in LDS, the
// operand ring's place, in the ownership of the row statistics -> the exchange and the workgroup's own 64 columns (lnx_tail_mx).
template <bool TWO = false>     // TWO: two workgroups per CU (257 ... 512 tiles: SmCfg<SM_MX, 2, true>, two ring slots of 40 KB)
__device__ __forceinline__ void lnx_body(const GemmHArgs& g, unsigned char* lds_raw, int tm, int tn, int tid
#ifdef JMID_SMALL_TRACE
                                         , unsigned long long* sm_trace_p
#endif
) {
    const int lane = tid & 63, wid = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int wr = wid / 2, wc = wid % 2;
    const int m0 = tm * 64, n0 = tn * 64;
    // (the tile indices reach the prefetch through an opaque vector copy: with a visible vector use in front of the K loop hipcc
    //  computes them in vector registers altogether, and the K loop's scalar-base copies fail to compile - "illegal VGPR to SGPR copy")
    int m0_v = m0, c_v = tn;
    asm volatile("" : "+v"(m0_v), "+v"(c_v));
    if (g.cmb_O) lnx_combine(g, tm, tn, m0, tid);
    LnxPre pre;
    lnx_prefetch(g, c_v, m0_v, tid, pre);
    const f32x16 acc = small_kloop<SM_MX, 2, TWO, true>(g, lds_raw, tm, tn, tid SM_TRACE_ARG);
    lnx_prefetch_landed(pre);
    __syncthreads();                                   // everybody is done with the operand ring: it becomes the staging tile
    float* stg = reinterpret_cast<float*>(lds_raw);
    {
        float* sr = stg + (wr * 32 + l31) * SM_STG_LD + wc * 32 + 4 * hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + n0 + wc * 32 + 8 * q + 4 * hi);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(acc[4 * q + e], kWInv, bv[e]);
            *reinterpret_cast<f32x4*>(sr + 8 * q) = o;
        }
    }
    __syncthreads();
    SM_STAMP(4);
    lnx_tail_mx(g, pre, stg, stg + 64 * SM_STG_LD, tm, tn, m0, tid);
    SM_STAMP(5);
}


template <int EPI, int OUT, int MODE, int WC, bool TWO = false>
__global__ __launch_bounds__(128 * WC, TWO ? WC : 1) void gemm_small_kernel(GemmHArgs g, int ntm, int ntn, int gw, int flags, unsigned mper, unsigned mgw) {
    using C = SmCfg<MODE, WC, TWO>;
    constexpr bool X2 = MODE != SM_X3, MX = MODE == SM_MX;
    constexpr int BM = C::BM, BN = C::BN, NS = C::NS, L = C::L, KB = C::KB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    args_now_each(g, ntm, ntn, gw, flags, mper, mgw);      // (four dependent scalar-cache misses in front of the first copy otherwise)
#ifdef JMID_SMALL_TRACE
    unsigned long long* sm_trace_p = g_small_trace + (size_t)blockIdx.x * 64;     // (loaded before the ring starts: vmcnt stays the ring's)
    sm_trace_p = reinterpret_cast<unsigned long long*>(pin_uniform_rfl(reinterpret_cast<unsigned long long>(sm_trace_p)));
    if (SM_ABL(64)) return;          // ablation: the launch alone
#endif
    SM_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wid / WC, wc = wid % WC;
    // XCD-contiguous ranges of the tile sequence (block b runs on XCD b % 8: for speed only); the sequence is cut into column
    // groups of gw N-tiles, each walked M-major with the group's N-tiles fastest
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int s = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const int per = ntm * gw, cg = fast_div(s, per, mper), rem = s - cg * per;      // (reciprocals from the host: common.hpp)
    const int tm = fast_div(rem, gw, mgw), tn = cg * gw + (rem - tm * gw);
    const int m0 = tm * BM, n0 = tn * BN;
    f32x16 acc[1][1];
    if constexpr (OUT == OUT_LNX) {
        // Transposed product: a lane owns token row l31 of its wave's block and four runs of 4 consecutive columns.  The fp32 rows
        // (accumulator + bias: what the stand-alone GEMM hands add_ln2) are staged in LDS - the operand ring's place - in the ownership
        // of the row statistics; then the exchange and the workgroup's own 64 columns (lnx_tail_mx)
        static_assert(WC == 2, "the LayerNorm tail is written for 256 threads");
        if constexpr (MODE == SM_MX) {
            lnx_body<TWO>(g, lds_raw, tm, tn, tid SM_TRACE_ARG);
            SM_STAMP(5);
        }
        return;
    }
    // the ConcatSquash GEMMs (and, on request, linear1) run transposed with the row-wise epilogue, as in the large-tile kernels
    if constexpr (csl_rowwise<EPI, OUT>() || (EPI == EPI_BIAS_RELU && OUT == OUT_SPLIT)) {
        if (csl_rowwise<EPI, OUT>() ? (!MX || (flags & 4)) : (flags & 8)) {
            acc[0][0] = small_kloop<MODE, WC, TWO, true>(g, lds_raw, tm, tn, tid SM_TRACE_ARG);
            csl_swapped_epilogue<1, 1, EPI, OUT, X2>(g, acc, m0 + wr * 32, n0 + wc * 32, l31, hi);
            SM_STAMP(4);
            return;
        }
    }
    // Q / K tiles of in_proj (a 64 x 128 tile is one head of one of Q, K, V): transposed product, rows out through LDS
    if constexpr (OUT == OUT_QKV && WC == 4) {
        if (n0 < 2 * g.d && g.d % BN == 0 && tune_small_qk(flags)) {
            acc[0][0] = small_kloop<MODE, WC, TWO, true>(g, lds_raw, tm, tn, tid SM_TRACE_ARG);
            __syncthreads();                                   // everybody is done with the operand ring: it becomes the staging tile
            small_qk_staged_store<MX>(g, acc[0][0], m0, n0, reinterpret_cast<half_t*>(lds_raw), tid);
            SM_STAMP(4);
            return;
        }
    }
    if constexpr (OUT != OUT_LNX) {
    acc[0][0] = small_kloop<MODE, WC, TWO, false>(g, lds_raw, tm, tn, tid SM_TRACE_ARG);
    gemm_h_epilogue<1, 1, EPI, OUT, X2, MX && OUT == OUT_QKV>(g, acc, m0, n0, wr, wc, l31, hi, BM, BN);
    }
    SM_STAMP(4);
#ifdef JMID_SMALL_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SM_STAMP(5);
#endif
}


// bytes all eight XCDs pull from the Infinity Cache with pn column groups: every XCD its column group's share of W and the A rows
// of its part of the M range
inline int small_pick_groups(const GemmHArgs& g, int ntn, double w_bytes_per_el, double a_bytes_per_el) {
    const double wb = (double)g.N * g.K * w_bytes_per_el, ab = (double)g.M * g.K * a_bytes_per_el;
    int best = 1;
    double best_bytes = 8.0 * wb + ab;
    for (int pn = 2; pn <= 8; pn *= 2) {
        if (ntn % pn != 0) break;
        const double bytes = 8.0 * wb / pn + pn * ab;
        if (bytes < best_bytes) {
            best = pn;
            best_bytes = bytes;
        }
    }
    return best;
}

template <int EPI, int OUT, int MODE, int WC, bool TWO = false>
inline hipError_t launch_gemm_small_cfg(const GemmHArgs& g, hipStream_t st) {
    using C = SmCfg<MODE, WC, TWO>;
    const int ntm = (g.M + C::BM - 1) / C::BM, ntn = g.N / C::BN;
    static DevSeen attr_seen;
    if (auto once_ = first_use_on_device(attr_seen))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_small_kernel<EPI, OUT, MODE, WC, TWO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
    const int pn = tune().small_pn > 0 ? (ntn % tune().small_pn == 0 ? tune().small_pn : 1)
                                       : small_pick_groups(g, ntn, MODE == SM_MX ? 3.0 : 4.0, MODE == SM_X3 ? 4.0 : 2.0);
    const int flags = (tune().csl_swap == 2 ? 0 : 4) | (tune().csl_swap == 3 ? 8 : 0) | (tune().small_qk == 2 ? 16 : 0);
    const int gw = ntn / pn;
    hipLaunchKernelGGL((gemm_small_kernel<EPI, OUT, MODE, WC, TWO>), dim3(ntm * ntn), dim3(C::NT), C::LDS_BYTES, st, g, ntm, ntn,
                       gw, flags, fast_div_magic(ntm * gw, (unsigned long long)ntm * ntn), fast_div_magic(gw, (unsigned long long)ntm * ntn));
    return hipGetLastError();
}

// Does this GEMM run on the small-launch kernel, and in which shape?  One workgroup per CU: at most 256 tiles.
//   -> 0 no, 2 / 4 = WC, 8 = WC 4 with two workgroups per CU.  "gemm_small" knob: 0 auto, 1 never.
inline int small_gemm_shape(const GemmHArgs& g) {
    if (tune().gemm_small == 1 || tune().gemm_h_variant != 0 || !tune().small_now) return 0;
    if (g.K % 128 != 0 || g.N % 128 != 0) return 0;       // (k128 ring stages)
    const long ntm = (g.M + 63) / 64;
    if (tune().small_now == 2) return g.x2 && ntm * (g.N / 128) <= 512 ? 8 : 0;      // (experiment "small_lanes" = 2)
    if (ntm * (g.N / 64) <= 256) return 2;
    if (ntm * (g.N / 128) <= 256) return 4;
    // 257 ... 512 tiles of 64 x 128 (two scenes; the reference's shipped K = 100): the same kernel, two workgroups per CU
    // (not F16X3: two k64 stages of its four operand planes do not fit half a CU's LDS)
    if (g.x2 && tune().gemm_small != 2 && ntm * (g.N / 128) <= 512) return 8;
    // (F16X3 at 257 ... 512 tiles in two rounds of one workgroup per CU measured slower than the round-3 kernels: 0.996 vs 0.977 ms
    //  per shipped-point call)
    return 0;
}

template <int EPI, int OUT, int MODE>
inline hipError_t launch_gemm_small_mode(const GemmHArgs& g, int wc, hipStream_t st) {
    if (wc == 2) return launch_gemm_small_cfg<EPI, OUT, MODE, 2>(g, st);
    if constexpr (OUT == OUT_LNX) {                                  // (N = 512: always the 64-column shape; 9 = two workgroups per CU)
        if constexpr (MODE == SM_MX)
            if (wc == 9) return launch_gemm_small_cfg<EPI, OUT, MODE, 2, true>(g, st);
        return hipErrorInvalidValue;
    }
    else {
        if constexpr (MODE != SM_X3)
            if (wc == 8) return launch_gemm_small_cfg<EPI, OUT, MODE, 4, true>(g, st);
        return launch_gemm_small_cfg<EPI, OUT, MODE, 4>(g, st);
    }
}

// does the out-projection's OUT_LNX launch also merge the partial outputs of a split-KV attention launch (lnx_combine)?  "small_cmb": 0 on, 2 off
inline bool small_cmb_fits(int nsplit, int head_dim, int x2) {
    return tune().small_cmb != 2 && tune().attn_h_variant != 1 && nsplit > 1 && nsplit <= 8 && head_dim == 128 && x2;      // (attn_h_variant 1: the register-staged kernel, which does not split)
}


// does out_proj / linear2 + residual + LayerNorm run as ONE small launch with the statistics exchange (OUT_LNX)?  F16MX at d_model 512,
// nothing else in flight on the handle, calls of ONE chunk (a call's bits must not depend on its chunk plan or its lanes), and EVERY
// workgroup of the launch resident at once - the waiting workgroups need their partners: 8 workgroups per 64-row tile against the
// device's compute units (Tuning::cus, from hipDeviceProp_t::multiProcessorCount at jmid_create - a partitioned or smaller device
// takes the unfused pair), one per CU, or two per CU on half the LDS each, and at most SM_LNX_MAX_TILES row tiles (the exchange
// buffer).  "small_lnx" = 2: off (GEMM + add_ln2, the same bits).
inline int small_lnx_fits(int M, int K) {       // 0 no; 2: one workgroup per CU; 9: two per CU ("small_lnx2" = 2 off)
    const long ntm = (M + 63) / 64;
    if (!(tune().gemm_small != 1 && tune().small_now == 1 && tune().one_chunk == 1 && tune().gemm_h_variant == 0 && tune().small_lnx != 2 && K % 128 == 0)) return 0;
    if (ntm * 8 <= tune().cus && ntm <= SM_LNX_MAX_TILES) return 2;
    return ntm * 8 <= 2L * tune().cus && ntm <= SM_LNX_MAX_TILES && tune().small_lnx2 != 2 ? 9 : 0;
}

template <int EPI, int OUT>
inline hipError_t launch_gemm_small(const GemmHArgs& g, int wc, hipStream_t st) {
    if (g.x2 && g.W8) return launch_gemm_small_mode<EPI, OUT, SM_MX>(g, wc, st);
    if (g.x2) return launch_gemm_small_mode<EPI, OUT, SM_X2>(g, wc, st);
    return launch_gemm_small_mode<EPI, OUT, SM_X3>(g, wc, st);
}

}  // namespace jmid
