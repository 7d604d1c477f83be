// JMID_PREC_F16MX, d_model 512: the row-complete GEMM + residual + LayerNorm of gemm_ln_f16x3.hpp, second generation.
//     X <- LayerNorm(X + A . W^T + b) * gamma + beta          (attention out-projection + norm1, linear2 + norm2;
//                                                              nn.TransformerEncoderLayer as built at MID/models/diffusion.py:161-166)
// What the first generation (gemm_ln_mx_kernel) lost: 55 us of a 101 / 145 us launch were its epilogue - every CU reading and
// writing its residual rows (4 bytes per element each way) in the same phase while HBM idled through the K loops, one
// workgroup per CU (the fp32 [64, 520] epilogue tile alone is 133 KB of LDS).  Here:
//   * the product is computed TRANSPOSED (W fragments as the first MFMA operand, their rows permuted so that a lane ends up with
//     runs of 8 consecutive columns of ONE token row): residual add, row statistics and the normalisation happen in the
//     accumulator registers - no epilogue tile; the statistics need one cross-lane and one cross-wave (1 KB of LDS) step;
//   * the residual stream's lo plane is a BYTE plane in this mode (bf8 of fp16(x - hi): hi + lo carries x to ~14 bits, and the
//     only readers of X_lo in F16X2 / F16MX are these LayerNorms - the GEMMs take X_hi; ADE against exact fp32 1.163e-5 vs
//     1.158e-5 m): 3 instead of 4 bytes per element each way, as 16-byte (hi) and 8-byte (lo) accesses per lane;
//   * bf8(W_lo) of a k64 block is wave-private: in the 128-row shape it comes through the LDS-DMA ring like the fp16 operands
//     (ordinary loads next to DMA copies make hipcc wait vmcnt(0) before the fp8 MFMAs: 121 -> 99 us per launch), in the 64-row
//     shape - whose 66 KB budget has no room for it - straight into registers.
// The row statistics are formed in ONE fixed order that both tile shapes, add_ln2_kernel (the unfused path for launches too small to
// fill the chip) and the one-launch GEMM + LayerNorm of small launches (gemm_small.hpp, OUT_LNX) reproduce, so all of them stay
// bit-identical and a chunk plan cannot change a result.  Since round 6 that order is BLOCK-WISE - a 64-column block forms its part of
// both statistics alone (mean AND squared deviations from its OWN mean, merged a la Chan et al.), so the batch kernels need one LDS
// exchange instead of two and the small-launch kernel one exchange between workgroups instead of two:
//     s(c, h) = sum over (j, p, e) in that order of v[64 c + 32 j + 16 p + 8 h + e]               c = 0..7, j, p, h = 0..1, e = 0..7
//     S_c = s(c, 0) + s(c, 1),   mu_c = S_c / 64
//     q(c, h) = sum in the same order of (v - mu_c)^2,   Q_c = q(c, 0) + q(c, 1)
//     mean = (((((((S_0 + S_1) + S_2) + S_3) + S_4) + S_5) + S_6) + S_7) / 512
//     dm   = sum over c = 0..7, in that order, of (mu_c - mean)^2
//     var  = ((((((((Q_0 + Q_1) + Q_2) + Q_3) + Q_4) + Q_5) + Q_6) + Q_7) + 64 dm) / 512,   rstd = rsqrt(var + eps)
//     (the squares accumulate as q = fma(t, t, q); the output is fma(fma(v, rstd, -mean rstd), gamma, beta): explicit fused operations,
//      the same in every kernel - the build has -ffp-contract=off)
// (Rounds 2-5 summed the squared deviations from the ROW mean: the same variance to fp32 accuracy, another summation order.)
#pragma once
#include "gemm_ln_f16x3.hpp"

namespace jmid {

// byte plane [rows, K] in 128-row x 32-column tiles like the fp16 panels (common.hpp::blk_index), rows of a tile 32 bytes apart
__host__ __device__ __forceinline__ size_t blk8_index(int row, int k, int K) {
    const int rb = row >> 7, r = row & 127, kb = k >> 5, kk = k & 31;
    return (((size_t)rb * (K >> 5) + kb) * 128 + r) * 32 + kk;
}
typedef i32x2_e i32x2;

// 8 bf8 bytes (two dwords) -> 8 floats: a bf8 value is the top byte of an fp16
__device__ __forceinline__ void f32_of_bf8x8(i32x2 b, float (&o)[8]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const unsigned d = (unsigned)b[u];
        const f16x2 p01 = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(0u, d, 0x010c000cu));
        const f16x2 p23 = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(0u, d, 0x030c020cu));
        o[4 * u + 0] = (float)p01[0];
        o[4 * u + 1] = (float)p01[1];
        o[4 * u + 2] = (float)p23[0];
        o[4 * u + 3] = (float)p23[1];
    }
}
// 8 fp16 lo values -> their bf8 images (two dwords)
__device__ __forceinline__ i32x2 bf8x8_of_f16(const f16x8& l) {
    const i32x4 d = __builtin_bit_cast(i32x4, l);
    i32x2 r;
    r[0] = bf8_of_f16x4(d[0], d[1]);
    r[1] = bf8_of_f16x4(d[2], d[3]);
    return r;
}

struct GemmLn2Args {
    const half_t* Ahi;            // [M, K] fp16 plane, blocked panel layout
    const half_t* W16hi;          // [K/16][512][16] hi plane of the 2^8-scaled weight
    const unsigned char* W8;      // bf8(W_lo) in MFMA-fragment order (gemm_f16x3.hpp::w8_image_kernel)
    const float *bias, *gamma, *beta;
    half_t* Xh;                   // residual stream [M, 512]: fp16 hi plane (blocked) ...
    unsigned char* Xl8;           // ... and the bf8 image of its lo plane (blk8_index): read, then overwritten with the result
    int M, K;
    float eps;
    int* range_flag;
    int no_lo_out;                // the last LayerNorm of the net: nobody reads its lo plane
};

// Tile shapes: wave tile (32 WM) rows x (32 WN) columns = 128 accumulators either way; 16 / WN waves side by side cover the 512 columns.
//   <4, 2>  128 rows, 8 waves, one workgroup per CU: W streams L2 -> LDS once per 128 rows (the production shape for full launches:
//           with 64-row tiles two workgroups per CU pull 64 B / clk / CU of W, which is all a CU gets from L2 - measured 10 % slower)
//   <2, 4>  64 rows, 4 waves, 66 KB of LDS: two workgroups per CU, for launches that fill the chip better with 64-row tiles
template <int WM, int WN>
struct Gl2Cfg {
    static constexpr int NWAVES = 16 / WN, NT = 64 * NWAVES, BM = 32 * WM;
    static constexpr int NSW = 3, NSA = 3;
    static constexpr int W_STAGE = GLN_BN * 16;                 // halfs: the hi plane of a k16 slice, 16 KB
    static constexpr int A_STAGE = BM * 32;                     // halfs: a k32 tile of BM rows
    static constexpr int A_OFF = NSW * W_STAGE;                 // halfs
    // bf8(W_lo) of a k64 block: through LDS (a wave-private 4 KB buffer, refilled by DMA right after the block's fp8 MFMAs took it
    // out) in the one-workgroup-per-CU shape; straight into registers in the other, whose LDS budget has no room for it - there
    // the compiler waits vmcnt(0) before the fp8 MFMAs (it treats ordinary loads next to LDS-DMA copies as out of order)
    static constexpr bool W8_LDS = WM == 4;
    static constexpr size_t W8_OFF = size_t(A_OFF + NSA * A_STAGE) * sizeof(half_t);           // bytes: 48 KB + 12 / 24 KB of rings
    static constexpr size_t RING_BYTES = W8_OFF + (W8_LDS ? size_t(GLN_BN) * 64 : 0);          // + 32 KB
    static constexpr size_t LDS_BYTES = RING_BYTES + 3 * GLN_BN * sizeof(float);               // + bias, gamma, beta
    static_assert(2 * 8 * BM * sizeof(float) <= size_t(A_OFF) * sizeof(half_t), "reduction scratch must fit the W ring");
};

// the weight row (column of the output) MFMA row `r` of a 32-row fragment carries: bits 2 and 3 of r swapped, so that the
// accumulator registers 8p .. 8p+7 of a lane are the 8 consecutive columns 16 p + 8 hi + 0..7 of its token row
__device__ __forceinline__ int gl2_frag_row(int r) { return (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1); }

template <int WM, int WN>
__global__ __launch_bounds__(64 * (16 / WN), 2) void gemm_ln2_mx_kernel(GemmLn2Args g) {
    using C = Gl2Cfg<WM, WN>;
    constexpr int d = GLN_BN, BM = C::BM, WCOLS = 32 * WN;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int tm = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const int m0 = tm * BM;
    const int nk = g.K / 32, nsteps = g.K / 16;
    // (De-phasing experiment of round 6: a launch is ~2 row tiles per CU, every CU alternating between an MFMA-bound K loop and an HBM-bound
    //  epilogue, all 256 in step.  Letting the first-round workgroups of every second CU start 4 ... 23 us late - so that half of the chip
    //  is in an epilogue while the other half is in a K loop - made the launch no faster at any delay, and slower from 8 us on:
    //  profiles/r06_gemm_ln2_stagger.log.  Two 64-row tiles in flight per CU - ln_rows = 64 - lose 8-12 % to the doubled W stream:
    //  profiles/r06_gemm_ln2_rows_check.log.)
    // bias, gamma, beta of all 512 columns into LDS (6 KB): the epilogue reads them with LDS latency.  These are ordinary VMEM
    // loads: the ds_writes retire them before the DMA ring starts, so that vmcnt counts only the ring (+ the bf8(W_lo) loads).
    float* par = reinterpret_cast<float*>(lds_raw + C::RING_BYTES);
    if (tid < 256) {
        const f32x4 pb = *reinterpret_cast<const f32x4*>((tid < 128 ? g.bias : g.gamma) + (tid & 127) * 4);
        const f32x4 pc = tid < 128 ? *reinterpret_cast<const f32x4*>(g.beta + tid * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(par + (tid < 128 ? 0 : GLN_BN) + (tid & 127) * 4) = pb;
        if (tid < 128) *reinterpret_cast<f32x4*>(par + 2 * GLN_BN + tid * 4) = pc;
    }

    // every copy: wave-uniform base (pinned in scalar registers, through an integer) + ONE 32-bit lane offset, lane * 16 bytes - per-thread
    // 64-bit pointers cost a 64-bit vector add per copy and a v_readfirstlane pair for its LDS destination (29 of the K loop's ~170 vector
    // instructions per k64 block).  The lane offset is re-pinned once per k64 block: zero-extended and hoisted out of the loop as a 64-bit
    // pair it defeats the scalar-base form.
    const int wcs = __builtin_amdgcn_readfirstlane(wc);
    unsigned lane_off = (unsigned)lane * 16u;
    auto dma16 = [&](const void* s, void* dd) {
        const unsigned long long u = pin_uniform(reinterpret_cast<unsigned long long>(s));
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) char*>(u) + lane_off,
                                         (__attribute__((address_space(3))) void*)dd, 16, 0, 0);
    };
    // A: this tile's rows of a k32 panel image (a whole 8 KB image for 128 rows, its 4 KB half for 64); one wave-instruction per
    // wave and tile.  Past the end the last tile / slice is copied again into its own stage (identical bytes: harmless), which
    // keeps the number of DMA instructions in flight at every wait a compile-time constant.
    const half_t* a_src = WM == 4 ? g.Ahi + (size_t)tm * nk * 4096 + wcs * 512
                                  : g.Ahi + (size_t)(tm >> 1) * nk * 4096 + (tm & 1) * 2048 + wcs * 512;
    auto issueA = [&](int ka) {
        const int kk = ka < nk ? ka : nk - 1;
        dma16(a_src + (size_t)kk * 4096, lds + C::A_OFF + (kk % C::NSA) * C::A_STAGE + wcs * 512);
    };
    // W_hi: the wave's own 32 WN rows of a k16 slice (WN wave-instructions of 1 KB): wave-private, no workgroup barrier
    const half_t* w_src = g.W16hi + (size_t)(wcs * WCOLS) * 16;
    auto issueW = [&](int s) {
        const int ss = s < nsteps ? s : nsteps - 1;
        half_t* st = lds + (ss % C::NSW) * C::W_STAGE + wcs * (WCOLS * 16);
#pragma unroll
        for (int q = 0; q < WN; ++q) dma16(w_src + ((size_t)ss * GLN_BN + q * 32) * 16, st + q * 512);
    };
    // bf8(W_lo) of a k64 block: this wave's WN 32-column blocks, two 16-byte pieces each, L2 -> registers (wave-private data)
    const int frow = gl2_frag_row(l31);
    const unsigned char* w8src = g.W8 + (size_t)(wc * WN) * 2048 + (size_t)(frow + 32 * hi) * 16;
    const size_t w8_kstride = (size_t)(GLN_BN / 32) * 2048;
    i32x8 w8[WN];
    auto loadW8 = [&](int kb) {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const unsigned char* p = w8src + (size_t)kb * w8_kstride + j * 2048;
            const i32x4 lo = *reinterpret_cast<const i32x4*>(p);
            const i32x4 up = *reinterpret_cast<const i32x4*>(p + 1024);
            w8[j] = i32x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
        }
    };
    // W8_LDS: the wave's WN blocks x 2 pieces of 1 KB in image order (lane-linear copies); fragment reads pick the permuted row
    unsigned char* w8buf = lds_raw + C::W8_OFF + wcs * (WN * 2048);
    const unsigned char* w8dma = g.W8 + (size_t)(wcs * WN) * 2048;
    const int nkb = g.K / 64;
    auto issueW8 = [&](int kb) {
        const int kk = kb < nkb ? kb : nkb - 1;
#pragma unroll
        for (int q = 0; q < 2 * WN; ++q) dma16(w8dma + (size_t)kk * w8_kstride + q * 1024, w8buf + q * 1024);
    };
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int offA[WM][2], offW[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int row = i * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offA[i][ks] = row * 32 + (((ks * 2 + hi) ^ ((row >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) offW[j] = (wc * WCOLS + j * 32 + frow) * 16 + hi * 8;
    i32x8 a8[WM];

    // VMEM issue order: A0 W0 A1 W1 [W8(0)] | step s: W(s+2) [A(s/2+2) on even s], and
    //   W8_LDS: the 2 WN copies of the next block's bf8(W_lo) at the END of step s % 4 == 3.  Younger than W(s) at the top of
    //           step s:  s%4 == 0: A, W, W8 = 1 + 3 WN;  1: W8, W, A = 1 + 3 WN;  2 and 3: A, W = 1 + WN;
    //   else:   the 2 WN register loads at s % 4 == 0.  s%4 == 0: A, W = 1 + WN;  1: W, A, the loads = 1 + 3 WN;  3: W, A = 1 + WN;
    //           2: A, W = 1 + WN plus the loads IF they went out after W(s) - the compiler may order them either way inside
    //           step s - 2, so that wait takes them along.
    issueA(0);
    issueW(0);
    issueA(1);
    issueW(1);
    if (C::W8_LDS) issueW8(0);
    auto step = [&](const int s, auto q_c) {
        constexpr int Q = decltype(q_c)::value, ks = Q & 1;
        if (C::W8_LDS ? Q <= 1 : Q == 1) wait_vmcnt<1 + 3 * WN>();
        else wait_vmcnt<1 + WN>();
        if (ks == 0) __builtin_amdgcn_s_barrier();      // A tile s/2 landed for everybody; A stage (s/2 - 1) % 3 is free again
        __builtin_amdgcn_sched_barrier(0);
        issueW(s + 2);
        if (ks == 0) issueA((s >> 1) + 2);
        if (!C::W8_LDS && Q == 0) loadW8(s >> 2);
        const half_t* stA = lds + C::A_OFF + ((s >> 1) % C::NSA) * C::A_STAGE;
        const half_t* stW = lds + (s % C::NSW) * C::W_STAGE;
        f16x8 ah[WM], wh[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) ah[i] = *reinterpret_cast<const f16x8*>(stA + offA[i][ks]);
#pragma unroll
        for (int j = 0; j < WN; ++j) wh[j] = *reinterpret_cast<const f16x8*>(stW + offW[j]);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const i32x4 dw = __builtin_bit_cast(i32x4, ah[i]);
            a8[i][Q * 2 + 0] = bf8_of_f16x4(dw[0], dw[1]);
            a8[i][Q * 2 + 1] = bf8_of_f16x4(dw[2], dw[3]);
        }
        if (Q == 3) {
            if (C::W8_LDS) {
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const unsigned char* p = w8buf + j * 2048 + (frow + 32 * hi) * 16;
                    const i32x4 lo = *reinterpret_cast<const i32x4*>(p);
                    const i32x4 up = *reinterpret_cast<const i32x4*>(p + 1024);
                    w8[j] = i32x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
                }
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)       // both operands bf8, literal zero scales: the UNSCALED instruction (gemm_f16x3.hpp)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8[j], a8[i], acc[i][j], 1, 1, 0, 0, 0, 0);
            if (C::W8_LDS) {
                __builtin_amdgcn_sched_barrier(0);
                issueW8((s >> 2) + 1);      // the buffer is out of LDS (its reads were waited for before the MFMAs)
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < nsteps; s += 4) {
        asm volatile("" : "+v"(lane_off));
        step(s, std::integral_constant<int, 0>{});
        step(s + 1, std::integral_constant<int, 1>{});
        step(s + 2, std::integral_constant<int, 2>{});
        step(s + 3, std::integral_constant<int, 3>{});
    }
    wait_vmcnt<0>();                     // the copies past the end
    __builtin_amdgcn_s_barrier();        // everybody is done with the rings: their first KBs become the reduction scratch
    float* red = reinterpret_cast<float*>(lds_raw);      // [2 passes][8 column blocks of 64][BM rows]

    // ---- epilogue, in the accumulators: register 8p + e of acc[i][j] is column 32 WN wc + 32 j + 16 p + 8 hi + e of row 32 i + l31.
    // Statistics in the canonical order of this file's header: a partial per 64-column block c (j = 2 c', 2 c' + 1) and lane half.
    constexpr int NC = WN / 2;            // 64-column blocks per wave
    float mean[WM], rstd[WM];      // (set in pass3)
    // one row block (a literal at every call site: every accumulator index is a compile-time constant).  The residual chunks of
    // a row (16 + 8 bytes each) are requested together: one memory round trip per row block.
    // The residual chunks of ALL row blocks are requested before the first is used: the K loop's operand registers are free here
    // (96 registers of residual next to the 128 accumulators), and the four dependent memory round trips of a row-block-at-a-time
    // epilogue (each waited to vmcnt(0) before the next block's loads went out) become one.
    f16x8 xh_all[WM][2 * WN];
    i32x2 xb_all[WM][2 * WN];
    // Addresses: the tile is ONE 128-row block of the blocked planes (BM == 128) or half of one, so chunk (i, u) of a lane sits at
    //   wave-uniform base + compile-time constant + a per-lane offset that does not depend on the row block
    // (blk_index / blk8_index spelled out: panel = 128 rows x 32 columns, 64 resp. 32 bytes per row) - three 32-bit VGPR offsets
    // instead of a 64-bit address per access (32 of them live across the statistics passes spilled).
    const int wcu = __builtin_amdgcn_readfirstlane(wc);
    const size_t pan0 = (size_t)(m0 >> 7) * (d >> 5) + (size_t)wcu * (WCOLS >> 5);      // first panel of this wave's columns
    typedef __attribute__((address_space(1))) char gchar;
    // a scalar-register base per 4 KB of constant offset (the rest fits the instruction's immediate): pinned, or hipcc folds the
    // constants into 64-bit VECTOR addresses again.  (Through an integer: a pointer that passes an asm operand comes back generic.)
    auto sbase = [](const void* p, size_t bytes) {
        return reinterpret_cast<gchar*>(pin_uniform(reinterpret_cast<unsigned long long>(p) + bytes));
    };
    const size_t tile0 = pan0 * 4096 + (size_t)(m0 & 127) * 32;      // elements
    gchar* xh_b[WN][WM / 2];      // [u >> 1][i >> 1]
    gchar* x8_b[WN];              // [u >> 1]
#pragma unroll
    for (int q = 0; q < WN; ++q) {
#pragma unroll
        for (int ih = 0; ih < WM / 2; ++ih) xh_b[q][ih] = sbase(g.Xh, (tile0 + q * 4096 + ih * 2048) * sizeof(half_t));
        x8_b[q] = sbase(g.Xl8, tile0 + q * 4096);
    }
    // the fp16 panels are XOR-swizzled in 16-byte chunks (common.hpp::blk_index: chunk c of row r at c ^ ((r >> 2) & 3), and bits 2-3 of
    // the tile row are those of l31): chunk 2 (u & 1) + hi of a lane sits at one of TWO lane offsets, by the parity of u
    const unsigned xoff8 = (unsigned)(l31 * 32 + hi * 8);
    const int xsw = (l31 >> 2) & 3;
    const unsigned xoffh[2] = {(unsigned)(l31 * 64 + ((hi ^ xsw) << 4)), (unsigned)(l31 * 64 + (((2 + hi) ^ xsw) << 4))};
    auto xh_at = [&](int i, int u) { return xh_b[u >> 1][i >> 1] + (i & 1) * 2048 + xoffh[u & 1]; };
    auto x8_at = [&](int i, int u) { return x8_b[u >> 1] + i * 1024 + (u & 1) * 16 + xoff8; };
    auto load1 = [&](auto i_c) {
        constexpr int i = decltype(i_c)::value;       // rows past M exist in the padded planes: loads need no guard
#pragma unroll
        for (int u = 0; u < 2 * WN; ++u) {
            xh_all[i][u] = *reinterpret_cast<const __attribute__((address_space(1))) f16x8*>(xh_at(i, u));
            xb_all[i][u] = *reinterpret_cast<const __attribute__((address_space(1))) i32x2*>(x8_at(i, u));
        }
    };
    auto pass1 = [&](auto i_c) {
        constexpr int i = decltype(i_c)::value;
        const f16x8 (&xh)[2 * WN] = xh_all[i];
        const i32x2 (&xb)[2 * WN] = xb_all[i];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float s = 0.f;
#pragma unroll
            for (int u = 4 * c; u < 4 * c + 4; ++u) {
                const int j = u >> 1, p = u & 1;
                const int c0 = wc * WCOLS + j * 32 + p * 16 + hi * 8;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(par + c0), b1 = *reinterpret_cast<const f32x4*>(par + c0 + 4);
                float xl[8];
                f32_of_bf8x8(xb[u], xl);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = fmaf(acc[i][j][8 * p + e], kWInv, e < 4 ? b0[e] : b1[e - 4]);
                    const float a = (float)xh[u][e] + xl[e];
                    const float v = a + y;
                    acc[i][j][8 * p + e] = v;
                    s += v;
                }
            }
            // the block's own mean, then its squared deviations from it: nothing of another wave is needed
            // (the cross-half exchange as v_permlane32_swap, not ds_bpermute: mc is needed right away, and an LDS round trip in this
            //  dependency chain - four row blocks in a row - made the block-wise order 6 % SLOWER than the two-pass form it replaced)
            float s0, s1;
            half_swap(s, s0, s1);
            const float S = s0 + s1;
            const float mc = S / 64.f;
            float q = 0.f;
#pragma unroll
            for (int j = 2 * c; j < 2 * c + 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float t = acc[i][j][e] - mc;
                    q = fmaf(t, t, q);
                }
            // both lane halves of a row write the same sums (a + b == b + a) to the same words: no divergent store
            red[(wc * NC + c) * BM + i * 32 + l31] = S;
            float q0, q1;
            half_swap(q, q0, q1);
            red[8 * BM + (wc * NC + c) * BM + i * 32 + l31] = q0 + q1;
        }
    };
    auto row_total = [&](const float* r8p, int r) {      // the 8 partials of row r, summed in column order
        float t = r8p[r] + r8p[BM + r];
#pragma unroll
        for (int c = 2; c < 8; ++c) t += r8p[c * BM + r];
        return t;
    };
    unsigned amax16 = 0;
    auto pass3 = [&](auto i_c) {
        constexpr int i = decltype(i_c)::value;
        const int r = i * 32 + l31, row = m0 + r;
        mean[i] = row_total(red, r) / (float)d;
        float dm = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float t = red[c * BM + r] / 64.f - mean[i];
            dm += t * t;
        }
        rstd[i] = rsqrtf((row_total(red + 8 * BM, r) + 64.f * dm) / (float)d + g.eps);
        const float nmr = -mean[i] * rstd[i];      // (x - mean) rstd gamma + beta as two fused multiply-adds per element: fma(fma(x, rstd, -mean rstd), gamma, beta)
        unsigned am = 0;
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int c0 = wc * WCOLS + j * 32 + p * 16 + hi * 8;
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(par + GLN_BN + c0), g1 = *reinterpret_cast<const f32x4*>(par + GLN_BN + c0 + 4);
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(par + 2 * GLN_BN + c0), t1 = *reinterpret_cast<const f32x4*>(par + 2 * GLN_BN + c0 + 4);
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    o[e] = fmaf(fmaf(acc[i][j][8 * p + e], rstd[i], nmr), e < 4 ? g0[e] : g1[e - 4], e < 4 ? t0[e] : t1[e - 4]);
                const Split4 s0 = split_f32x4(o[0], o[1], o[2], o[3], am), s1 = split_f32x4(o[4], o[5], o[6], o[7], am);
                if (row < g.M) {
                    *reinterpret_cast<__attribute__((address_space(1))) i32x4*>(xh_at(i, 2 * j + p)) = i32x4{s0.hi[0], s0.hi[1], s1.hi[0], s1.hi[1]};
                    if (!g.no_lo_out)
                        *reinterpret_cast<__attribute__((address_space(1))) i32x2*>(x8_at(i, 2 * j + p)) =
                            i32x2{bf8_of_f16x4(s0.lo[0], s0.lo[1]), bf8_of_f16x4(s1.lo[0], s1.lo[1])};
                }
            }
        if (row < g.M) {      // (rows past M hold whatever the padding held)
            const u16x2_s m = __builtin_elementwise_max(__builtin_bit_cast(u16x2_s, amax16), __builtin_bit_cast(u16x2_s, am));
            amax16 = __builtin_bit_cast(unsigned, m);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    // (all four at once is 5 registers more than the budget of two waves per SIMD holds: three, and the fourth as soon as the
    // first block's registers are free - its round trip passes under the statistics of blocks 1 and 2)
    load1(I0{}); load1(I1{});
    if constexpr (WM == 4) load1(I2{});
    __builtin_amdgcn_sched_barrier(0);
    pass1(I0{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (WM == 4) load1(I3{});
    __builtin_amdgcn_sched_barrier(0);
    pass1(I1{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (WM == 4) { pass1(I2{}); __builtin_amdgcn_sched_barrier(0); pass1(I3{}); }
    __syncthreads();
    pass3(I0{});
    __builtin_amdgcn_sched_barrier(0);
    pass3(I1{});
    if constexpr (WM == 4) {
        __builtin_amdgcn_sched_barrier(0);
        pass3(I2{});
        __builtin_amdgcn_sched_barrier(0);
        pass3(I3{});
    }
    if (split_range_exceeded(amax16)) atomicOr(g.range_flag, 1);
}

template <int WM, int WN>
inline void launch_gemm_ln2_cfg(const GemmLn2Args& g, hipStream_t st) {
    using C = Gl2Cfg<WM, WN>;
    static DevSeen seen;
    if (auto once_ = first_use_on_device(seen))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ln2_mx_kernel<WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)C::LDS_BYTES);
    hipLaunchKernelGGL((gemm_ln2_mx_kernel<WM, WN>), dim3((g.M + C::BM - 1) / C::BM), dim3(C::NT), C::LDS_BYTES, st, g);
}
inline hipError_t launch_gemm_ln2_mx(const GemmLn2Args& g, hipStream_t st) {
#ifdef JMID_DIAGNOSTICS
    // the 64-row shape (two workgroups per CU; measured 10 % slower on full launches, and hipcc spills 48 registers in it) exists
    // in the diagnostics flavour only, behind the "ln_rows" knob: the production library always runs the 128-row shape
    if (tune().ln_rows == 64) {
        launch_gemm_ln2_cfg<2, 4>(g, st);
        return hipGetLastError();
    }
#endif
    launch_gemm_ln2_cfg<4, 2>(g, st);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// The unfused partner (launches too small to fill the chip): X <- LN(X + Y) for the same planes, with the row statistics
// summed in the canonical order - lane (row, c, h) of a wave of 4 rows owns the 32 columns of partial(c, h).
static __global__ __launch_bounds__(256) void add_ln2_kernel(const float* Y, const float* gamma, const float* beta, int M, float eps,
                                                      half_t* Xh, unsigned char* Xl8, int no_lo_out, int* range_flag) {
    constexpr int d = GLN_BN;
    args_now_each(Y, gamma, beta, M, eps, Xh, Xl8, no_lo_out, range_flag);
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const int c = (lane >> 1) & 7, hi = lane & 1;
    const int rowc = row < M ? row : M - 1;
    float v[32];
    float s = 0.f;
    // gamma / beta requested with the row itself: behind the statistics they were four more dependent round trips (the compiler
    // cannot move them above the stores of the block before)
    f32x4 gm[8], bt[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c0 = c * 64 + (u >> 1) * 32 + (u & 1) * 16 + hi * 8;
        gm[2 * u] = *reinterpret_cast<const f32x4*>(gamma + c0), gm[2 * u + 1] = *reinterpret_cast<const f32x4*>(gamma + c0 + 4);
        bt[2 * u] = *reinterpret_cast<const f32x4*>(beta + c0), bt[2 * u + 1] = *reinterpret_cast<const f32x4*>(beta + c0 + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c0 = c * 64 + (u >> 1) * 32 + (u & 1) * 16 + hi * 8;
        const f16x8 xh = *reinterpret_cast<const f16x8*>(Xh + blk_index(rowc, c0, d));
        const i32x2 xb = *reinterpret_cast<const i32x2*>(Xl8 + blk8_index(rowc, c0, d));
        const f32x4 y0 = *reinterpret_cast<const f32x4*>(Y + (size_t)rowc * d + c0), y1 = *reinterpret_cast<const f32x4*>(Y + (size_t)rowc * d + c0 + 4);
        float xl[8];
        f32_of_bf8x8(xb, xl);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = (float)xh[e] + xl[e];
            const float t = a + (e < 4 ? y0[e] : y1[e - 4]);
            v[u * 8 + e] = t;
            s += t;
        }
    }
    const int base = lane & ~15;
    auto row_total = [&](float blk) {      // the 8 blocks' values (lanes base + 2 k of this row) in column order
        float t = __shfl(blk, base, 64) + __shfl(blk, base + 2, 64);
#pragma unroll
        for (int k = 2; k < 8; ++k) t += __shfl(blk, base + 2 * k, 64);
        return t;
    };
    // this file's canonical order: the block's sum S_c and its squared deviations from its OWN mean, then the merge over the 8 blocks
    const float S = s + __shfl_xor(s, 1, 64);
    const float mc = S / 64.f;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        const float t = v[e] - mc;
        q = fmaf(t, t, q);
    }
    const float Q = q + __shfl_xor(q, 1, 64);
    const float mean = row_total(S) / (float)d;
    float dm = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float t = __shfl(S, base + 2 * k, 64) / 64.f - mean;
        dm += t * t;
    }
    const float rstd = rsqrtf((row_total(Q) + 64.f * dm) / (float)d + eps);
    const float nmr = -mean * rstd;
    bool overflow = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c0 = c * 64 + (u >> 1) * 32 + (u & 1) * 16 + hi * 8;
        const f32x4 g0 = gm[2 * u], g1 = gm[2 * u + 1], t0 = bt[2 * u], t1 = bt[2 * u + 1];
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
        if (row < M) {
            *reinterpret_cast<f16x8*>(Xh + blk_index(row, c0, d)) = vh;
            if (!no_lo_out) *reinterpret_cast<i32x2*>(Xl8 + blk8_index(row, c0, d)) = bf8x8_of_f16(vl);
        }
    }
    if (overflow && row < M) atomicOr(range_flag, 1);
}

// fp32 row-major [rows, 512] <-> the residual-stream planes of this mode (fp16 hi, blocked; bf8 image of lo): diagnostics
static __global__ void split_planes_lo8_kernel(const float* in, half_t* hi, unsigned char* lo8, int rows) {
    const size_t n = (size_t)rows * GLN_BN;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / GLN_BN), k = (int)(i % GLN_BN);
        half_t h, l;
        split_f32(in[i], h, l);
        hi[blk_index(r, k, GLN_BN)] = h;
        lo8[blk8_index(r, k, GLN_BN)] = bf8_of_f16(l);
    }
}
static __global__ void merge_planes_lo8_kernel(const half_t* hi, const unsigned char* lo8, float* out, int rows) {
    const size_t n = (size_t)rows * GLN_BN;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / GLN_BN), k = (int)(i % GLN_BN);
        const unsigned short lb = (unsigned short)(lo8[blk8_index(r, k, GLN_BN)]) << 8;
        out[i] = (float)hi[blk_index(r, k, GLN_BN)] + (float)__builtin_bit_cast(half_t, lb);
    }
}

}  // namespace jmid
