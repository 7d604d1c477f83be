// The tail of one net evaluation in ONE kernel (d_model 512 only):
//     concat3 (ConcatSquash 512 -> 256)  ->  concat4 (256 -> 128)  ->  linear (128 -> 2)  ->  DDIM / DDPM update
//     ->  embedding of the updated x for the next step                 diffusion.py:207-209, 524-528, 183-185
// None of the three layers has an activation between them and every one is row-local, so a workgroup that owns BM
// complete rows can run them back to back with the 256-wide and the 128-wide intermediates in LDS: three launches
// (two split-fp16 GEMMs with ConcatSquash epilogues + out_ddim_kernel) become one, and the Y3 hi/lo planes and the
// fp32 Y4 rows (61 200 x (256 x 4 + 128 x 4 x 2) B = 125 MB per chunk and step) never go to HBM.
//
// Same arithmetic, in the same order, as the unfused path (gemm_h_epilogue_impl<EPI_CSL> for the two ConcatSquash
// epilogues, mfma3 per k16 step, out_ddim_row for the rest): results are bit-identical to it, whatever BM.
//
// 4 waves.  Phase A (K = 512): wave w owns columns 64w..64w+63 of Y3 for all BM rows; X k32-tiles and W3 k32-tiles
// (256 rows: two 8 KB panel images per plane) come in through a 2-stage LDS-DMA ring, one barrier per tile.  Its
// epilogue writes the gated Y3 tile as hi/lo planes in the blocked k32 layout into LDS - the A operand of phase B
// (K = 256, wave w owns columns 32w..32w+31 of Y4, W4 tiles through the recycled ring).  Phase B's epilogue leaves
// the gated fp32 Y4 rows in LDS and phase C is out_ddim_row with one wave per row.
#pragma once
#include "elementwise.hpp"
#include "gemm_f16x3.hpp"

namespace jmid {

struct TailArgs {
    const half_t *Xh, *Xl;       // [M, 512] blocked planes: input of concat3 (the next-step embedding overwrites them)
    const half_t *W3h, *W3l;     // [256, 512] blocked, pre-scaled by kWScale
    const half_t *W4h, *W4l;     // [128, 256] blocked
    const float *b3, *b4;
    const float* hyp;            // [EA, hyp_ld] ctx part of the hyper nets
    const float* thyp;           // [hyp_ld] time part of THIS step
    int hyp_ld, g3, bb3, g4, bb4;
    RowMap rmap;
    int M;
    int* range_flag;
};

constexpr int TAIL_D = 512, TAIL_DM = 256, TAIL_DL = 128;
constexpr int TAIL_Y4_LD = TAIL_DL + 4;          // floats per row of the Y4 tile
template <int WM>
constexpr size_t tail_lds_bytes() {
    // ring of phase A (2 stages of [X tile hi/lo | W3 tile hi/lo]) + Y3 tile hi/lo
    return (size_t(2) * (2 * WM * 32 * 32 + 2 * TAIL_DM * 32) + size_t(2) * WM * 32 * TAIL_DM) * sizeof(half_t);
}

template <int WM, bool X2, bool EMBED_NEXT>
__global__ __launch_bounds__(256, 1) void tail_f16x3_kernel(TailArgs g, OutArgs oa, EmbedArgs nxt) {
    constexpr int BM = 32 * WM;
    constexpr int A_ST = 2 * BM * 32;                 // halfs: X k32 tile, hi then lo
    constexpr int W_ST = 2 * TAIL_DM * 32;            // halfs: W3 k32 tile, hi (2 panels) then lo
    constexpr int STAGE = A_ST + W_ST;
    constexpr int Y3_OFF = 2 * STAGE;                 // halfs
    constexpr int Y3_PLANE = BM * TAIL_DM;
    constexpr int W4_STAGE = 2 * TAIL_DL * 32;        // halfs: hi plane (4096) then lo
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.x * BM;
    static_assert(Y3_OFF * sizeof(half_t) >= 2 * W4_STAGE * sizeof(half_t) + size_t(BM) * TAIL_Y4_LD * sizeof(float),
                  "phase B ring + Y4 tile must fit the phase A ring");

    // ------------------------------------------------------------------ phase A: Y3 = CSL3(X . W3^T)
    constexpr int nkA = TAIL_D / 32;
    const half_t* xa = g.Xh + (size_t)(m0 >> 7) * nkA * 4096 + (m0 & 127) * 32 + tid * 8;
    const half_t* xl = g.Xl + (size_t)(m0 >> 7) * nkA * 4096 + (m0 & 127) * 32 + tid * 8;
    auto issueA = [&](int kt) {
        half_t* st = lds + (kt & 1) * STAGE;
        if (wid < BM / 16) {      // BM * 32 halfs per plane = BM / 16 wave-instructions of 1 KB
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xa + (size_t)kt * 4096),
                                             (__attribute__((address_space(3))) void*)(st + wid * 512), 16, 0, 0);
            if (!X2)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xl + (size_t)kt * 4096),
                                                 (__attribute__((address_space(3))) void*)(st + BM * 32 + wid * 512), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {          // W3 tile: 16 chunks of 512 halfs per plane, chunk c = panel (c >> 3), part (c & 7)
            const int c = q * 4 + wid;
            const size_t so = ((size_t)(c >> 3) * nkA + kt) * 4096 + (c & 7) * 512 + lane * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.W3h + so),
                                             (__attribute__((address_space(3))) void*)(st + A_ST + c * 512), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.W3l + so),
                                             (__attribute__((address_space(3))) void*)(st + A_ST + TAIL_DM * 32 + c * 512), 16, 0, 0);
        }
    };
    int offA[WM][2], offW3[2][2];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int row = i * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offA[i][ks] = row * 32 + (((ks * 2 + hi) ^ ((row >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = wid * 64 + j * 32 + l31, r = n & 127;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offW3[j][ks] = (n >> 7) * 4096 + r * 32 + (((ks * 2 + hi) ^ ((r >> 2) & 3)) * 8);
    }
    {
        f32x16 acc[WM][2];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        issueA(0);
        for (int kt = 0; kt < nkA; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // tile kt landed for everybody; the other stage is free again
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nkA) issueA(kt + 1);
            const half_t* st = lds + (kt & 1) * STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 ah[WM], al[WM], wh[2], wl[2];
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8*>(st + offA[i][ks]);
                    if (!X2) al[i] = *reinterpret_cast<const f16x8*>(st + BM * 32 + offA[i][ks]);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    wh[j] = *reinterpret_cast<const f16x8*>(st + A_ST + offW3[j][ks]);
                    wl[j] = *reinterpret_cast<const f16x8*>(st + A_ST + TAIL_DM * 32 + offW3[j][ks]);
                }
                mfma3<WM, 2, X2>(ah, al, wh, wl, acc);
            }
        }
        // epilogue A: bias + ConcatSquash gate / bias (the arithmetic of gemm_h_epilogue_impl<EPI_CSL, OUT_SPLIT>),
        // hi/lo planes of the tile into LDS in the blocked k32 layout
        bool overflow = false;
        half_t* y3h = lds + Y3_OFF;
        half_t* y3l = y3h + Y3_PLANE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = wid * 64 + j * 32 + l31;
            float bv = g.b3[n], tg = g.thyp[g.g3 + n], tb = g.thyp[g.bb3 + n];
            asm volatile("" : "+v"(bv), "+v"(tg), "+v"(tb));
            const int kb = n >> 5;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = i * 32 + frag_row(r, hi), m = m0 + ml;
                    const int mc = m < g.M ? m : g.M - 1;
                    float v = fmaf(acc[i][j][r], kWInv, bv);
                    const float* hrow = g.hyp + (size_t)g.rmap.ea(mc) * g.hyp_ld;
                    v = fmaf(v, sigmoidf_(hrow[g.g3 + n] + tg), hrow[g.bb3 + n] + tb);
                    half_t h, l;
                    split_f32(v, h, l);
                    overflow |= (m < g.M) && !(fabsf(v) <= kHalfMax);
                    const int o = kb * (BM * 32) + ml * 32 + ((((l31 >> 3) ^ ((ml >> 2) & 3)) << 3) | (l31 & 7));
                    y3h[o] = h;
                    if (!X2) y3l[o] = l;
                }
        }
        if (overflow) atomicOr(g.range_flag, 1);
    }
    __syncthreads();                               // Y3 tile complete, phase A ring free

    // ------------------------------------------------------------------ phase B: Y4 = CSL4(Y3 . W4^T)
    constexpr int nkB = TAIL_DM / 32;
    float* y4 = reinterpret_cast<float*>(lds_raw + 2 * W4_STAGE * sizeof(half_t));
    {
        auto issueB = [&](int kt) {
            half_t* st = lds + (kt & 1) * W4_STAGE;
#pragma unroll
            for (int q = 0; q < 2; ++q) {          // 8 chunks of 512 halfs per plane
                const int c = q * 4 + wid;
                const size_t so = (size_t)kt * 4096 + c * 512 + lane * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.W4h + so),
                                                 (__attribute__((address_space(3))) void*)(st + c * 512), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.W4l + so),
                                                 (__attribute__((address_space(3))) void*)(st + TAIL_DL * 32 + c * 512), 16, 0, 0);
            }
        };
        int offW4[2];
        {
            const int n = wid * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) offW4[ks] = n * 32 + (((ks * 2 + hi) ^ ((n >> 2) & 3)) * 8);
        }
        f32x16 acc[WM][1];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
        const half_t* y3h = lds + Y3_OFF;
        const half_t* y3l = y3h + Y3_PLANE;
        issueB(0);
        for (int kt = 0; kt < nkB; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nkB) issueB(kt + 1);
            const half_t* st = lds + (kt & 1) * W4_STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 ah[WM], al[WM], wh[1], wl[1];
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8*>(y3h + kt * (BM * 32) + offA[i][ks]);
                    if (!X2) al[i] = *reinterpret_cast<const f16x8*>(y3l + kt * (BM * 32) + offA[i][ks]);
                }
                wh[0] = *reinterpret_cast<const f16x8*>(st + offW4[ks]);
                wl[0] = *reinterpret_cast<const f16x8*>(st + TAIL_DL * 32 + offW4[ks]);
                mfma3<WM, 1, X2>(ah, al, wh, wl, acc);
            }
        }
        // epilogue B (gemm_h_epilogue_impl<EPI_CSL, OUT_F32>): gated fp32 rows into the Y4 tile.  The tile lies behind
        // the W4 ring, whose last stage other waves may still be reading: it does not overlap it
        const int n = wid * 32 + l31;
        float bv = g.b4[n], tg = g.thyp[g.g4 + n], tb = g.thyp[g.bb4 + n];
        asm volatile("" : "+v"(bv), "+v"(tg), "+v"(tb));
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = i * 32 + frag_row(r, hi), m = m0 + ml;
                const int mc = m < g.M ? m : g.M - 1;
                float v = fmaf(acc[i][0][r], kWInv, bv);
                const float* hrow = g.hyp + (size_t)g.rmap.ea(mc) * g.hyp_ld;
                v = fmaf(v, sigmoidf_(hrow[g.g4 + n] + tg), hrow[g.bb4 + n] + tb);
                y4[ml * TAIL_Y4_LD + n] = v;
            }
    }
    __syncthreads();

    // ------------------------------------------------------------------ phase C: output layer + sampler update + next embedding
    for (int rr = wid; rr < BM; rr += 4) {
        const int m = m0 + rr;
        if (m < g.M) out_ddim_row<EMBED_NEXT>(oa, nxt, m, lane, y4 + rr * TAIL_Y4_LD);
    }
}

template <bool X2>
inline hipError_t launch_tail_mode(const TailArgs& g, const OutArgs& oa, const EmbedArgs& nxt, bool embed_next, int rows,
                                   hipStream_t st) {
    static bool attr_seen[64] = {};
    if (first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_f16x3_kernel<1, X2, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)tail_lds_bytes<1>());
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_f16x3_kernel<1, X2, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)tail_lds_bytes<1>());
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_f16x3_kernel<2, X2, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)tail_lds_bytes<2>());
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_f16x3_kernel<2, X2, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)tail_lds_bytes<2>());
    }
    if (rows == 32) {
        const dim3 grid((g.M + 31) / 32);
        if (embed_next) hipLaunchKernelGGL((tail_f16x3_kernel<1, X2, true>), grid, dim3(256), tail_lds_bytes<1>(), st, g, oa, nxt);
        else hipLaunchKernelGGL((tail_f16x3_kernel<1, X2, false>), grid, dim3(256), tail_lds_bytes<1>(), st, g, oa, nxt);
    } else {
        const dim3 grid((g.M + 63) / 64);
        if (embed_next) hipLaunchKernelGGL((tail_f16x3_kernel<2, X2, true>), grid, dim3(256), tail_lds_bytes<2>(), st, g, oa, nxt);
        else hipLaunchKernelGGL((tail_f16x3_kernel<2, X2, false>), grid, dim3(256), tail_lds_bytes<2>(), st, g, oa, nxt);
    }
    return hipGetLastError();
}

inline hipError_t launch_tail(const TailArgs& g, const OutArgs& oa, const EmbedArgs& nxt, bool embed_next, int rows, bool x2,
                              hipStream_t st) {
    return x2 ? launch_tail_mode<true>(g, oa, nxt, embed_next, rows, st) : launch_tail_mode<false>(g, oa, nxt, embed_next, rows, st);
}

}  // namespace jmid
