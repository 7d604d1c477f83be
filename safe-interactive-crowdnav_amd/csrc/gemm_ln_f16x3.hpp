// Row-complete split-fp16 GEMM with the residual add and the post-LayerNorm of nn.TransformerEncoderLayer fused in:
//     X <- LayerNorm(X + A . W^T + b) * gamma + beta          (attention out-projection + norm1, linear2 + norm2)
// A workgroup owns 64 full rows (BN = N = 512), so the row statistics never leave the chip and the GEMM output never
// goes to HBM: per LayerNorm that is one fp32 [M, 512] write + read and one kernel launch less than GEMM + add_ln.
//
// 8 waves side by side along N (wave tile 64 x 64).  W is not shared between waves in this shape, A is shared by all:
//   A ring: 4 stages of one k32 tile (hi + lo = 8 KB, ONE DMA wave-instruction per wave) -> 3 tiles of look-ahead,
//           enough to cover an HBM miss (the A rows were just written by the previous kernel);
//   W ring: 3 stages of one k16 slice (2 planes x 16 KB) from a k16-panel copy of the weight ([K/16][512][16] halfs,
//           made at load time): each wave copies only its own 64 rows - four DMA instructions of 1 KB of contiguous
//           lines - so the W ring is wave-private and only the A tiles (every other step) need a workgroup barrier.
// All DMAs retire in order; the schedule below keeps exactly five wave-instructions in flight at every wait.
// Epilogue: accumulators + bias -> fp32 tile in LDS (row stride 520 floats: conflict-free), then one wave per row does
// the same arithmetic, in the same order, as add_ln_kernel<2, true> (elementwise.hpp) - the fused and the unfused
// paths give bit-identical rows.
#pragma once
#include "gemm_f16x3.hpp"

namespace jmid {

constexpr int GLN_BM = 64, GLN_BN = 512;
constexpr int GLN_W_STAGE = 2 * GLN_BN * 16;     // halfs per W stage (hi plane then lo plane)
constexpr int GLN_A_STAGE = 2 * GLN_BM * 32;     // halfs per A stage
constexpr int GLN_W_OFF = 0;
constexpr int GLN_A_OFF = 3 * GLN_W_STAGE;
constexpr size_t GLN_LDS_BYTES = size_t(GLN_A_OFF + 5 * GLN_A_STAGE) * sizeof(half_t);   // 136 KB (>= the epilogue tile)
constexpr int GLN_TILE_LD = GLN_BN + 8;          // floats per row of the epilogue tile
static_assert(size_t(GLN_BM) * GLN_TILE_LD * sizeof(float) <= GLN_LDS_BYTES, "epilogue tile must fit the ring");


struct GemmLnArgs {
    const half_t *Ahi, *Alo;      // [M, K] blocked panel layout (common.hpp::blk_index)
    const half_t *W16hi, *W16lo;  // [K/16][512][16]
    const float* bias;            // [512]
    const float *gamma, *beta;    // LayerNorm affine [512]
    half_t *Xh, *Xl;              // residual stream planes [M, 512] blocked: read, then overwritten with the result
    int M, K;
    float eps;
    int* range_flag;
    int x2;                       // JMID_PREC_F16X2: two-term product A_hi x (W_hi + W_lo)
    const unsigned char* W8;      // JMID_PREC_F16MX: bf8 image of W_lo (gemm_f16x3.hpp::w8_image_kernel), or null
    int no_lo_out;                // the lo plane of the result is not written: nobody reads it (the last LayerNorm of the net in
                                  // F16X2 / F16MX: the tail GEMM takes X_hi only, the next step starts from a fresh embedding)
};

// fp32 row-major [512, K] -> k16-panel hi/lo planes
static __global__ void split_planes_k16_kernel(const float* in, half_t* hi, half_t* lo, int rows, int K) {
    const size_t n = (size_t)rows * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / K), k = (int)(i % K);
        half_t h, l;
        split_f32(in[i] * kWScale, h, l);
        const size_t o = ((size_t)(k >> 4) * rows + r) * 16 + (k & 15);
        hi[o] = h;
        lo[o] = l;
    }
}

// epilogue of the 64-row kernels (same arithmetic, in the same order, as add_ln_kernel<2, true>)
__device__ __forceinline__ void gln64_epilogue(const GemmLnArgs& g, f32x16 (&accm)[2][2], unsigned char* lds_raw, int m0, int wid,
                                               int wc, int lane, int l31, int hi) {
    constexpr int WM = 2, WN = 2;
    // ---- epilogue.  The residual rows (8 per wave) are requested first: their HBM latency hides under the tile
    // write.  Rows past M exist in the padded panels, so the loads need no guard (the stores do).
    constexpr int d = GLN_BN;
    f16x4 rph[8][2], rpl[8][2];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const size_t ob = blk_index(m0 + wid * 8 + rr, (i * 64 + lane) * 4, d);
            rph[rr][i] = *reinterpret_cast<const f16x4*>(g.Xh + ob);
            rpl[rr][i] = *reinterpret_cast<const f16x4*>(g.Xl + ob);
        }
    f32x4 gm[2], bt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        gm[i] = *reinterpret_cast<const f32x4*>(g.gamma + (i * 64 + lane) * 4);
        bt[i] = *reinterpret_cast<const f32x4*>(g.beta + (i * 64 + lane) * 4);
    }
    // Y tile (fp32, + bias) into LDS
    __builtin_amdgcn_s_barrier();          // everybody is done with the rings (all DMAs have landed: vmcnt(0) above)
    float* tile = reinterpret_cast<float*>(lds_raw);
    {
        float bv[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) bv[j] = g.bias[wc * 64 + j * 32 + l31];
#pragma unroll
        for (int j = 0; j < WN; ++j) asm volatile("" : "+v"(bv[j]));
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tile[(i * 32 + frag_row(r, hi)) * GLN_TILE_LD + wc * 64 + j * 32 + l31] =
                        fmaf(accm[i][j][r], kWInv, bv[j]);
    }
    __syncthreads();
    // residual + LayerNorm, one wave per row, 8 rows per wave (the arithmetic of add_ln_kernel<2, true>)
    bool overflow = false;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int trow = wid * 8 + rr, row = m0 + trow;
        f32x4 v[2];
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = (i * 64 + lane) * 4;
            f32x4 a;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = (float)rph[rr][i][e] + (float)rpl[rr][i][e];
            const f32x4 y = *reinterpret_cast<const f32x4*>(tile + trow * GLN_TILE_LD + c);
            v[i] = a + y;
            sacc += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
        const float mean = wave_sum(sacc) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = v[i][e] - mean;
                q += t * t;
            }
        const float rstd = rsqrtf(wave_sum(q) / (float)d + g.eps);
        if (row < g.M) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = (i * 64 + lane) * 4;
                f16x4 vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o = (v[i][e] - mean) * rstd * gm[i][e] + bt[i][e];
                    half_t hh, ll;
                    split_f32(o, hh, ll);
                    overflow |= !(fabsf(o) <= kHalfMax);
                    vh[e] = hh;
                    vl[e] = ll;
                }
                const size_t ob = blk_index(row, c, d);
                *reinterpret_cast<f16x4*>(g.Xh + ob) = vh;
                if (!g.no_lo_out) *reinterpret_cast<f16x4*>(g.Xl + ob) = vl;
            }
        }
    }
    if (overflow) atomicOr(g.range_flag, 1);
}

template <bool X2>
__global__ __launch_bounds__(512, 2) void gemm_ln_f16x3_kernel(GemmLnArgs g, int ntm) {
    constexpr int WM = 2, WN = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wc = wid;
    // XCD-contiguous ranges of row tiles: the tiles an XCD works on concurrently stream the same W slices through its L2
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int tm = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const int m0 = tm * GLN_BM;
    const int nk = g.K / 32, nsteps = 2 * nk;

    const half_t* a_src = (tid < 256 ? g.Ahi : g.Alo) + (size_t)(tm >> 1) * nk * 4096 + (tm & 1) * 2048 + (tid & 255) * 8;
    const int a_dst = (tid < 256 ? 0 : 2048) + (wid & 3) * 512;
    auto issueA = [&](int ka) {     // one wave-instruction; past the end: the last tile again into its own stage
                                    // (identical bytes: harmless while that stage is being read) so that the DMA count stays fixed
        if (X2 && wid >= 4) return;   // F16X2: waves 4-7 would copy A_lo, which is not read (their vmcnt is 4 below)
        const int kk = ka < nk ? ka : nk - 1;
        half_t* dst = lds + GLN_A_OFF + (kk & 3) * GLN_A_STAGE + a_dst;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src + (size_t)kk * 4096),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    // W is private to a wave in this shape: every wave copies and reads only its own 64 rows of the slice (four
    // wave-instructions of 1 KB: hi rows 0-31, hi 32-63, lo 0-31, lo 32-63), so W needs no workgroup barrier - only A does
    auto issueW = [&](int s, int stage) {
        half_t* st = lds + GLN_W_OFF + stage * GLN_W_STAGE + wc * 64 * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const half_t* src = ((q >> 1) ? g.W16lo : g.W16hi) + ((size_t)s * GLN_BN + wc * 64 + (q & 1) * 32) * 16 + lane * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(st + (q >> 1) * GLN_BN * 16 + (q & 1) * 512),
                                             16, 0, 0);
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
    int offA[WM][2], offW[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int row = i * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offA[i][ks] = row * 32 + (((ks * 2 + hi) ^ ((row >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) offW[j] = (wc * 64 + j * 32 + l31) * 16 + hi * 8;

    // issue order A0 A1 W0 A2 W1, then per step W(s+2) [+ A(s/2+3) on even steps]: at the top of step s the
    // instructions younger than W(s) are always one A and one W slice = 5.
    issueA(0);
    issueA(1);
    issueW(0, 0);
    issueA(2);
    issueW(1, 1);
    int wst = 0;
    // one k16 step; ks (which half of the A tile) is a literal at both call sites so every LDS offset stays in a register
    auto step = [&](const int s, const int ks) {
        if (s + 1 >= nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (X2 && wid >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        if (ks == 0) __builtin_amdgcn_s_barrier();   // A tile s/2 landed for everybody; A stage (s/2 - 1) is free again
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < nsteps) issueW(s + 2, wst == 0 ? 2 : wst - 1);   // (wst + 2) % 3
        if (ks == 0) issueA((s >> 1) + 3);
        const half_t* stA = lds + GLN_A_OFF + ((s >> 1) & 3) * GLN_A_STAGE;
        const half_t* stW = lds + GLN_W_OFF + wst * GLN_W_STAGE;
        f16x8 ah[WM], al[WM], wh[WN], wl[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            ah[i] = *reinterpret_cast<const f16x8*>(stA + offA[i][ks]);
            if (!X2) al[i] = *reinterpret_cast<const f16x8*>(stA + 2048 + offA[i][ks]);
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            wh[j] = *reinterpret_cast<const f16x8*>(stW + offW[j]);
            wl[j] = *reinterpret_cast<const f16x8*>(stW + GLN_BN * 16 + offW[j]);
        }
        mfma3<WM, WN, X2>(ah, al, wh, wl, accm);
        wst = wst == 2 ? 0 : wst + 1;
    };
    for (int s = 0; s < nsteps; s += 2) {
        step(s, 0);
        step(s + 1, 1);
    }

    gln64_epilogue(g, accm, lds_raw, m0, wid, wc, lane, l31, hi);
}

// ---------------------------------------------------------------------------------------------------------------
// 128-row variant (chosen by grid fill, launch_gemm_ln): wave tile 128 x 64 (one accumulator set = 128 VGPRs), so W streams through
// L2 -> LDS once per 128 rows instead of once per 64.  A ring: 3 stages of a whole k32 panel image per plane (2 DMA
// instructions per wave); W ring as above.  Six DMA wave-instructions are younger than W(s) at every wait.  The fp32
// epilogue tile holds 64 rows, so the epilogue runs in two passes.
constexpr int GLN2_BM = 128;
constexpr int GLN2_A_STAGE = 2 * GLN2_BM * 32;   // halfs: hi image (4096) + lo image
constexpr int GLN2_A_OFF = 3 * GLN_W_STAGE;
constexpr size_t GLN2_LDS_BYTES = size_t(GLN2_A_OFF + 3 * GLN2_A_STAGE) * sizeof(half_t);   // 96 + 48 = 144 KB
static_assert(size_t(64) * GLN_TILE_LD * sizeof(float) <= GLN2_LDS_BYTES, "epilogue tile must fit the ring");

// epilogue of the 128-row kernels in two passes of 64 rows (same arithmetic as the 64-row kernel / add_ln_kernel<2, true>)
__device__ __forceinline__ void gln128_epilogue(const GemmLnArgs& g, f32x16 (&acc)[4][2], unsigned char* lds_raw, int m0, int wid,
                                                int wc, int lane, int l31, int hi) {
    constexpr int WN = 2;
    constexpr int d = GLN_BN;
    float* tile = reinterpret_cast<float*>(lds_raw);
    f32x4 gm[2], bt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        gm[i] = *reinterpret_cast<const f32x4*>(g.gamma + (i * 64 + lane) * 4);
        bt[i] = *reinterpret_cast<const f32x4*>(g.beta + (i * 64 + lane) * 4);
    }
    float bv[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) bv[j] = g.bias[wc * 64 + j * 32 + l31];
#pragma unroll
    for (int j = 0; j < WN; ++j) asm volatile("" : "+v"(bv[j]));
    bool overflow = false;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        f16x4 rph[8][2], rpl[8][2];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const size_t ob = blk_index(m0 + pass * 64 + wid * 8 + rr, (i * 64 + lane) * 4, d);
                rph[rr][i] = *reinterpret_cast<const f16x4*>(g.Xh + ob);
                rpl[rr][i] = *reinterpret_cast<const f16x4*>(g.Xl + ob);
            }
        __syncthreads();   // pass 0: everybody is done with the rings; pass 1: everybody has read the tile of pass 0
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tile[(i2 * 32 + frag_row(r, hi)) * GLN_TILE_LD + wc * 64 + j * 32 + l31] =
                        fmaf(acc[pass * 2 + i2][j][r], kWInv, bv[j]);
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int trow = wid * 8 + rr, row = m0 + pass * 64 + trow;
            f32x4 v[2];
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = (i * 64 + lane) * 4;
                f32x4 a;
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = (float)rph[rr][i][e] + (float)rpl[rr][i][e];
                const f32x4 y = *reinterpret_cast<const f32x4*>(tile + trow * GLN_TILE_LD + c);
                v[i] = a + y;
                sacc += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            }
            const float mean = wave_sum(sacc) / (float)d;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = v[i][e] - mean;
                    q += t * t;
                }
            const float rstd = rsqrtf(wave_sum(q) / (float)d + g.eps);
            if (row < g.M) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int c = (i * 64 + lane) * 4;
                    f16x4 vh, vl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float o = (v[i][e] - mean) * rstd * gm[i][e] + bt[i][e];
                        half_t hh, ll;
                        split_f32(o, hh, ll);
                        overflow |= !(fabsf(o) <= kHalfMax);
                        vh[e] = hh;
                        vl[e] = ll;
                    }
                    const size_t ob = blk_index(row, c, d);
                    *reinterpret_cast<f16x4*>(g.Xh + ob) = vh;
                    if (!g.no_lo_out) *reinterpret_cast<f16x4*>(g.Xl + ob) = vl;
                }
            }
        }
    }
    if (overflow) atomicOr(g.range_flag, 1);
}

template <bool X2>
__global__ __launch_bounds__(512, 2) void gemm_ln128_f16x3_kernel(GemmLnArgs g, int ntm) {
    constexpr int WM = 4, WN = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wc = wid;
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int tm = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const int m0 = tm * GLN2_BM;
    const int nk = g.K / 32, nsteps = 2 * nk;

    // copies: wave-uniform base (pinned in scalar registers, through an integer) + ONE 32-bit lane offset, lane * 16 bytes (per-thread
    // 64-bit pointers cost a 64-bit vector add per copy and a v_readfirstlane pair for its LDS destination)
    const int wcs = __builtin_amdgcn_readfirstlane(wid);
    unsigned lane_off = (unsigned)lane * 16u;      // (re-pinned per pair of steps below)
    auto dma16 = [&](const void* sp, void* dd) {
        const unsigned long long u = pin_uniform(reinterpret_cast<unsigned long long>(sp));
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) char*>(u) + lane_off,
                                         (__attribute__((address_space(3))) void*)dd, 16, 0, 0);
    };
    const half_t* a_hi = g.Ahi + (size_t)tm * nk * 4096 + wcs * 512;
    const half_t* a_lo = g.Alo + (size_t)tm * nk * 4096 + wcs * 512;
    auto issueA = [&](int ka) {     // two wave-instructions; past the end: the last tile again into its own stage
        const int kk = ka < nk ? ka : nk - 1;       // (identical bytes: harmless while that stage is being read)
        half_t* dst = lds + GLN2_A_OFF + (kk % 3) * GLN2_A_STAGE + wcs * 512;
        dma16(a_hi + (size_t)kk * 4096, dst);
        if (X2) return;             // F16X2: A_lo is neither copied nor read (one instruction per tile: vmcnt 5 below)
        dma16(a_lo + (size_t)kk * 4096, dst + 4096);
    };
    auto issueW = [&](int s, int stage) {
        half_t* st = lds + GLN_W_OFF + stage * GLN_W_STAGE + wcs * 64 * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            dma16(((q >> 1) ? g.W16lo : g.W16hi) + ((size_t)s * GLN_BN + wcs * 64 + (q & 1) * 32) * 16, st + (q >> 1) * GLN_BN * 16 + (q & 1) * 512);
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
    for (int j = 0; j < WN; ++j) offW[j] = (wc * 64 + j * 32 + l31) * 16 + hi * 8;

    // issue order A0 W0 A1 W1, then per step W(s+2) [+ A(s/2+2) on even steps]
    issueA(0);
    issueW(0, 0);
    issueA(1);
    issueW(1, 1);
    int wst = 0, ast = 0;
    auto step = [&](const int s, const int ks) {
        if (s + 1 >= nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (X2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        if (ks == 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < nsteps) issueW(s + 2, wst == 0 ? 2 : wst - 1);
        if (ks == 0) issueA((s >> 1) + 2);
        const half_t* stA = lds + GLN2_A_OFF + ast * GLN2_A_STAGE;
        const half_t* stW = lds + GLN_W_OFF + wst * GLN_W_STAGE;
        f16x8 ah[WM], al[WM], wh[WN], wl[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            ah[i] = *reinterpret_cast<const f16x8*>(stA + offA[i][ks]);
            if (!X2) al[i] = *reinterpret_cast<const f16x8*>(stA + 4096 + offA[i][ks]);
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            wh[j] = *reinterpret_cast<const f16x8*>(stW + offW[j]);
            wl[j] = *reinterpret_cast<const f16x8*>(stW + GLN_BN * 16 + offW[j]);
        }
        mfma3<WM, WN, X2>(ah, al, wh, wl, acc);
        wst = wst == 2 ? 0 : wst + 1;
        if (ks == 1) ast = ast == 2 ? 0 : ast + 1;
    };
    for (int s = 0; s < nsteps; s += 2) {
        asm volatile("" : "+v"(lane_off));
        step(s, 0);
        step(s + 1, 1);
    }

    gln128_epilogue(g, acc, lds_raw, m0, wid, wc, lane, l31, hi);
}


// ---------------------------------------------------------------------------------------------------------------
// JMID_PREC_F16MX variant of the 128-row kernel (gemm_f16x3.hpp, gemm_mx_dma256x256_kernel): A_hi . W_hi in fp16 as above and
// the correction term bf8(A_hi) . fp8(W_lo) as ONE block-scaled fp8 MFMA per k64 and output tile.  The W ring carries only the
// hi slices (3 x 16 KB); the fp8 image of a k64 block (a wave's own 64 columns: 4 KB, four DMA instructions) has ONE buffer,
// refilled right after the block's fp8 MFMAs have taken it out of LDS - four k16 steps ahead of its next use.  DMA
// instructions younger than W(s) at the top of step s: W(s + 1) and one A tile = 3, plus the four of the fp8 image when it was
// issued one or two steps ago (s % 4 <= 1).
constexpr int GLNX_W_STAGE = GLN_BN * 16;                 // halfs per W stage: the hi plane of a k16 slice
constexpr int GLNX_W8_OFF = 3 * GLNX_W_STAGE * 2;         // bytes
constexpr int GLNX_A_OFF = (GLNX_W8_OFF + GLN_BN * 64) / 2;   // halfs
constexpr int GLNX_A_STAGE = GLN2_BM * 32;                // halfs: the hi image of a k32 tile
constexpr size_t GLNX_RING_BYTES = size_t(GLNX_A_OFF + 3 * GLNX_A_STAGE) * sizeof(half_t);        // 48 + 32 + 24 = 104 KB
constexpr size_t GLNX_LDS_BYTES = GLNX_RING_BYTES > size_t(64) * GLN_TILE_LD * sizeof(float) ? GLNX_RING_BYTES
                                                                                              : size_t(64) * GLN_TILE_LD * sizeof(float);

template <int WM>      // 4: 128-row tiles, 2: 64-row tiles (chosen by grid fill like the F16X2 kernels)
__global__ __launch_bounds__(512, 2) void gemm_ln_mx_kernel(GemmLnArgs g, int ntm) {
    constexpr int WN = 2, BM = 32 * WM;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wc = wid;
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int tm = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const int m0 = tm * BM;
    const int nk = g.K / 32, nsteps = 2 * nk, nkb = g.K / 64;

    auto dma16 = [](const void* s, void* d) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s, (__attribute__((address_space(3))) void*)d, 16, 0, 0);
    };
    // a k32 tile of A_hi: a whole 8 KB panel image (128 rows, one instruction per wave) or a 4 KB half (64 rows: waves 0-3 only)
    const bool a_wave = WM == 4 || wid < 4;
    const half_t* a_hi = WM == 4 ? g.Ahi + (size_t)tm * nk * 4096 + tid * 8
                                 : g.Ahi + (size_t)(tm >> 1) * nk * 4096 + (tm & 1) * 2048 + (tid & 255) * 8;
    auto issueA = [&](int ka) {     // one wave-instruction; past the end: the last tile again into its own stage
        if (!a_wave) return;
        const int kk = ka < nk ? ka : nk - 1;
        dma16(a_hi + (size_t)kk * 4096, lds + GLNX_A_OFF + (kk % 3) * GLNX_A_STAGE + (WM == 4 ? wid : wid & 3) * 512);
    };
    auto issueW = [&](int s, int stage) {
        half_t* st = lds + stage * GLNX_W_STAGE + wc * 64 * 16;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            dma16(g.W16hi + ((size_t)s * GLN_BN + wc * 64 + q * 32) * 16 + lane * 8, st + q * 512);
    };
    unsigned char* w8buf = lds_raw + GLNX_W8_OFF + wc * 4096;      // [2 column blocks][2 pieces][64 lanes][16 B]
    const unsigned char* w8src = g.W8 + (size_t)wc * 2 * 2048 + lane * 16;
    auto issueW8 = [&](int kb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16(w8src + (size_t)kb * (GLN_BN / 32) * 2048 + q * 1024, w8buf + q * 1024);
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
    for (int j = 0; j < WN; ++j) offW[j] = (wc * 64 + j * 32 + l31) * 16 + hi * 8;
    i32x8 a8[WM];

    issueW8(0);
    issueA(0);
    issueW(0, 0);
    issueA(1);
    issueW(1, 1);
    int wst = 0, ast = 0;
    auto step = [&](const int s, auto q_c) {
        constexpr int Q = decltype(q_c)::value, ks = Q & 1;
        if (s + 1 >= nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (a_wave) {
            if (Q <= 1 && s >= 4) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {                    // 64-row tiles, waves 4-7: no A instruction of their own in flight
            if (Q <= 1 && s >= 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        if (ks == 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < nsteps) issueW(s + 2, wst == 0 ? 2 : wst - 1);
        if (ks == 0) issueA((s >> 1) + 2);
        const half_t* stA = lds + GLNX_A_OFF + ast * GLNX_A_STAGE;
        const half_t* stW = lds + wst * GLNX_W_STAGE;
        f16x8 ah[WM], wh[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) ah[i] = *reinterpret_cast<const f16x8*>(stA + offA[i][ks]);
#pragma unroll
        for (int j = 0; j < WN; ++j) wh[j] = *reinterpret_cast<const f16x8*>(stW + offW[j]);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const i32x4 dw = __builtin_bit_cast(i32x4, ah[i]);
            a8[i][Q * 2 + 0] = bf8_of_f16x4(dw[0], dw[1]);
            a8[i][Q * 2 + 1] = bf8_of_f16x4(dw[2], dw[3]);
        }
        if (Q == 3) {
            i32x8 w8[WN];
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const unsigned char* p = w8buf + j * 2048 + lane * 16;
                const i32x4 lo = *reinterpret_cast<const i32x4*>(p);
                const i32x4 up = *reinterpret_cast<const i32x4*>(p + 1024);
                w8[j] = i32x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
#ifdef JMID_PROBE_SCALED_MFMA   // tools/concurrency_probe9.hip only: the same product through the scaled instruction PAIR, both scales 2^0
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], w8[j], acc[i][j], 1, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#else
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], w8[j], acc[i][j], 1, 1, 0, 0, 0, 0);   // unscaled, both bf8
#endif
            __builtin_amdgcn_sched_barrier(0);
            if ((s >> 2) + 1 < nkb) issueW8((s >> 2) + 1);
        }
        wst = wst == 2 ? 0 : wst + 1;
        if (ks == 1) ast = ast == 2 ? 0 : ast + 1;
    };
    for (int s = 0; s < nsteps; s += 4) {
        step(s, std::integral_constant<int, 0>{});
        step(s + 1, std::integral_constant<int, 1>{});
        step(s + 2, std::integral_constant<int, 2>{});
        step(s + 3, std::integral_constant<int, 3>{});
    }
    if constexpr (WM == 4) gln128_epilogue(g, acc, lds_raw, m0, wid, wc, lane, l31, hi);
    else gln64_epilogue(g, acc, lds_raw, m0, wid, wc, lane, l31, hi);
}

template <bool X2>
inline hipError_t launch_gemm_ln_mode(const GemmLnArgs& g, hipStream_t st) {
    static DevSeen attr_seen;
    if (auto once_ = first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ln_f16x3_kernel<X2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)GLN_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ln128_f16x3_kernel<X2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)GLN2_LDS_BYTES);
    }
    // row tile by how well the grid fills whole rounds of the 256 CUs (one workgroup per CU); at equal fill the
    // 128-row kernel is ~4 % faster (W streams through L2 -> LDS half as often)
    auto fill = [](long n) { return (double)n / (double)(((n + 255) / 256) * 256); };
    const long n128 = (g.M + GLN2_BM - 1) / GLN2_BM, n64 = (g.M + GLN_BM - 1) / GLN_BM;
    const bool rows128 = tune().ln_rows == 128 || (tune().ln_rows == 0 && 1.04 * fill(n128) >= fill(n64));
    if (rows128) {
        const int ntm = (g.M + GLN2_BM - 1) / GLN2_BM;
        hipLaunchKernelGGL(gemm_ln128_f16x3_kernel<X2>, dim3(ntm), dim3(512), GLN2_LDS_BYTES, st, g, ntm);
        return hipGetLastError();
    }
    const int ntm = (g.M + GLN_BM - 1) / GLN_BM;
    hipLaunchKernelGGL(gemm_ln_f16x3_kernel<X2>, dim3(ntm), dim3(512), GLN_LDS_BYTES, st, g, ntm);
    return hipGetLastError();
}

inline hipError_t launch_gemm_ln_mx(const GemmLnArgs& g, hipStream_t st) {
    static DevSeen attr_seen;
    if (auto once_ = first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ln_mx_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)GLNX_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ln_mx_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)GLNX_LDS_BYTES);
    }
    // the row tile by grid fill, as launch_gemm_ln_mode
    auto fill = [](long n) { return (double)n / (double)(((n + 255) / 256) * 256); };
    const long n128 = (g.M + GLN2_BM - 1) / GLN2_BM, n64 = (g.M + GLN_BM - 1) / GLN_BM;
    if (tune().ln_rows == 128 || (tune().ln_rows == 0 && 1.04 * fill(n128) >= fill(n64))) {
        hipLaunchKernelGGL(gemm_ln_mx_kernel<4>, dim3((int)n128), dim3(512), GLNX_LDS_BYTES, st, g, (int)n128);
    } else {
        hipLaunchKernelGGL(gemm_ln_mx_kernel<2>, dim3((int)n64), dim3(512), GLNX_LDS_BYTES, st, g, (int)n64);
    }
    return hipGetLastError();
}

inline hipError_t launch_gemm_ln(const GemmLnArgs& g, hipStream_t st) {
    if (g.x2 && g.W8 && g.K % 64 == 0) return launch_gemm_ln_mx(g, st);
    return g.x2 ? launch_gemm_ln_mode<true>(g, st) : launch_gemm_ln_mode<false>(g, st);
}

}  // namespace jmid
