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
// 4 waves.  Phase A (K = 512): wave w owns columns 64w..64w+63 of Y3 for all BM rows (pipeline of gemm_ln_f16x3_kernel:
// wave-private W3 k16 slices, shared X k32 tiles).  Its epilogue writes the gated Y3 tile as hi/lo planes in the blocked
// k32 layout into LDS - the A operand of phase B (K = 256, wave w owns columns 32w..32w+31 of Y4, wave-private W4
// slices, no barrier).  Phase B's epilogue leaves the gated fp32 Y4 rows in LDS and phase C is out_ddim_row, one wave
// per row.
#pragma once
#include "elementwise.hpp"
#include "gemm_f16x3.hpp"

namespace jmid {

struct TailArgs {
    const half_t *Xh, *Xl;       // [M, 512] blocked planes: input of concat3 (the next-step embedding overwrites them)
    const half_t *W3h, *W3l;     // [512/16][256][16] k16-panel copy of concat3's weight, pre-scaled by kWScale
    const half_t *W4h, *W4l;     // [256/16][128][16] k16-panel copy of concat4's weight
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
// LDS (halfs): [W3 ring: 3 stages x (hi 256 x 16 | lo 256 x 16)] [X ring: 4 stages x (hi BM x 32 | lo BM x 32)] [Y3 hi | Y3 lo]
constexpr int TAIL_W3_STAGE = 2 * TAIL_DM * 16;
template <int WM>
constexpr size_t tail_lds_bytes() {
    return (size_t(3) * TAIL_W3_STAGE + size_t(4) * 2 * WM * 32 * 32 + size_t(2) * WM * 32 * TAIL_DM) * sizeof(half_t);
}

// Phase A follows gemm_ln_f16x3_kernel's pipeline with 4 waves: every wave owns 64 columns of Y3, copies only its own
// 64 rows of a W3 k16 slice (four DMA instructions of 1 KB, 3-stage ring, no workgroup barrier) and shares the X k32
// tiles (4-stage ring, one barrier per tile).  Six (F16X2: five) DMA wave-instructions are younger than W(s) at every wait.
template <int WM, bool X2, bool EMBED_NEXT>
__global__ __launch_bounds__(256, 1) void tail_f16x3_kernel(TailArgs g, OutArgs oa, EmbedArgs nxt) {
    constexpr int BM = 32 * WM;
    constexpr int A_STAGE = 2 * BM * 32;              // halfs: X k32 tile, hi then lo
    constexpr int A_OFF = 3 * TAIL_W3_STAGE;
    constexpr int Y3_OFF = A_OFF + 4 * A_STAGE;
    constexpr int Y3_PLANE = BM * TAIL_DM;
    constexpr int W4_STAGE = 2 * TAIL_DL * 16;        // halfs: k16 slice of W4, hi (128 x 16) then lo
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.x * BM;
    static_assert(size_t(Y3_OFF) * sizeof(half_t) >= size_t(3) * W4_STAGE * sizeof(half_t) + size_t(BM) * TAIL_Y4_LD * sizeof(float),
                  "phase B ring + Y4 tile must fit in front of the Y3 tile");

    // ------------------------------------------------------------------ phase A: Y3 = CSL3(X . W3^T)
    constexpr int nkA = TAIL_D / 32, nstepsA = 2 * nkA;
    const half_t* xa = g.Xh + (size_t)(m0 >> 7) * nkA * 4096 + (m0 & 127) * 32 + tid * 8;
    const half_t* xl = g.Xl + (size_t)(m0 >> 7) * nkA * 4096 + (m0 & 127) * 32 + tid * 8;
    // X tile: BM * 32 halfs per plane = BM / 16 wave-instructions; waves beyond that copy the last one again (identical
    // bytes into the same place) so that every wave has the same number of DMA instructions in flight
    const int a_w = wid < BM / 16 ? wid : BM / 16 - 1;
    const int a_fix = (a_w - wid) * 512;              // source correction (halfs) of a duplicating wave
    auto issueA = [&](int ka) {
        const int kk = ka < nkA ? ka : nkA - 1;       // past the end: the last tile again into its own stage (harmless)
        half_t* dst = lds + A_OFF + (kk & 3) * A_STAGE + a_w * 512;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xa + a_fix + (size_t)kk * 4096),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        if (X2) return;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xl + a_fix + (size_t)kk * 4096),
                                         (__attribute__((address_space(3))) void*)(dst + BM * 32), 16, 0, 0);
    };
    auto issueW3 = [&](int s, int stage) {
        half_t* st = lds + stage * TAIL_W3_STAGE + wid * 64 * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const half_t* src = ((q >> 1) ? g.W3l : g.W3h) + ((size_t)s * TAIL_DM + wid * 64 + (q & 1) * 32) * 16 + lane * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(st + (q >> 1) * TAIL_DM * 16 + (q & 1) * 512),
                                             16, 0, 0);
        }
    };
    int offA[WM][2], offW3[2];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int row = i * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offA[i][ks] = row * 32 + (((ks * 2 + hi) ^ ((row >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) offW3[j] = (wid * 64 + j * 32 + l31) * 16 + hi * 8;
    {
        f32x16 acc[WM][2];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // issue order A0 A1 W0 A2 W1, then per step W(s+2) [+ A(s/2+3) on even steps]
        issueA(0);
        issueA(1);
        issueW3(0, 0);
        issueA(2);
        issueW3(1, 1);
        int wst = 0;
        auto step = [&](const int s, const int ks) {
            if (s + 1 >= nstepsA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (X2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            if (ks == 0) __builtin_amdgcn_s_barrier();   // X tile s/2 landed for everybody; X stage (s/2 - 1) is free again
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < nstepsA) issueW3(s + 2, wst == 0 ? 2 : wst - 1);
            if (ks == 0) issueA((s >> 1) + 3);
            const half_t* stA = lds + A_OFF + ((s >> 1) & 3) * A_STAGE;
            const half_t* stW = lds + wst * TAIL_W3_STAGE;
            f16x8 ah[WM], al[WM], wh[2], wl[2];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(stA + offA[i][ks]);
                if (!X2) al[i] = *reinterpret_cast<const f16x8*>(stA + BM * 32 + offA[i][ks]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                wh[j] = *reinterpret_cast<const f16x8*>(stW + offW3[j]);
                wl[j] = *reinterpret_cast<const f16x8*>(stW + TAIL_DM * 16 + offW3[j]);
            }
            mfma3<WM, 2, X2>(ah, al, wh, wl, acc);
            wst = wst == 2 ? 0 : wst + 1;
        };
        for (int s = 0; s < nstepsA; s += 2) {
            step(s, 0);
            step(s + 1, 1);
        }
        // epilogue A: bias + ConcatSquash gate / bias (the arithmetic of gemm_h_epilogue_impl<EPI_CSL, OUT_SPLIT>),
        // hi/lo planes of the tile into LDS in the blocked k32 layout (a region of its own: no barrier needed first)
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
    __syncthreads();                               // Y3 tile complete, phase A rings free (all DMAs landed: vmcnt(0) above)

    // ------------------------------------------------------------------ phase B: Y4 = CSL4(Y3 . W4^T)
    // wave w owns columns 32w..32w+31: its W4 k16 slices (32 rows: two DMA instructions of 1 KB) go through a wave-private
    // 3-stage ring, the A operand is the Y3 tile in LDS - no workgroup barrier inside the loop
    constexpr int nstepsB = TAIL_DM / 16;
    float* y4 = reinterpret_cast<float*>(lds_raw + size_t(3) * W4_STAGE * sizeof(half_t));
    {
        auto issueW4 = [&](int s, int stage) {
            half_t* st = lds + stage * W4_STAGE + wid * 32 * 16;
            const size_t so = ((size_t)s * TAIL_DL + wid * 32) * 16 + lane * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.W4h + so),
                                             (__attribute__((address_space(3))) void*)st, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.W4l + so),
                                             (__attribute__((address_space(3))) void*)(st + TAIL_DL * 16), 16, 0, 0);
        };
        const int offW4 = (wid * 32 + l31) * 16 + hi * 8;
        f32x16 acc[WM][1];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
        const half_t* y3h = lds + Y3_OFF;
        const half_t* y3l = y3h + Y3_PLANE;
        issueW4(0, 0);
        issueW4(1, 1);
        int wst = 0;
#pragma unroll
        for (int s = 0; s < nstepsB; ++s) {
            if (s + 1 >= nstepsB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // W4(s) landed; W4(s + 1) may still be in flight
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < nstepsB) issueW4(s + 2, wst == 0 ? 2 : wst - 1);
            const half_t* stW = lds + wst * W4_STAGE;
            const int kt = s >> 1, ks = s & 1;
            f16x8 ah[WM], al[WM], wh[1], wl[1];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(y3h + kt * (BM * 32) + offA[i][ks]);
                if (!X2) al[i] = *reinterpret_cast<const f16x8*>(y3l + kt * (BM * 32) + offA[i][ks]);
            }
            wh[0] = *reinterpret_cast<const f16x8*>(stW + offW4);
            wl[0] = *reinterpret_cast<const f16x8*>(stW + TAIL_DL * 16 + offW4);
            mfma3<WM, 1, X2>(ah, al, wh, wl, acc);
            wst = wst == 2 ? 0 : wst + 1;
        }
        // epilogue B (gemm_h_epilogue_impl<EPI_CSL, OUT_F32>): gated fp32 rows into the Y4 tile, which lies behind the
        // wave-private W4 rings and in front of the Y3 tile: nobody else's reads are disturbed
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
    // a wave owns rows wid, wid + 4, ...: all their dot products first (independent chains), then lane q does the
    // scalar update of row q (what lane 0 does in out_ddim_kernel - the same expressions), then the embeddings
    constexpr int RPW = BM / 4;
    float s0[RPW], s1[RPW];
#pragma unroll
    for (int q = 0; q < RPW; ++q) out_dot(oa, y4 + (wid + 4 * q) * TAIL_Y4_LD, lane, s0[q], s1[q]);
    float my0 = 0.f, my1 = 0.f;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        my0 = lane == q ? s0[q] : my0;
        my1 = lane == q ? s1[q] : my1;
    }
    float xn0 = 0.f, xn1 = 0.f;
    {
        const int m = m0 + wid + 4 * lane;
        if (lane < RPW && m < g.M) out_update(oa, m, my0, my1, xn0, xn1);
    }
    if (EMBED_NEXT) {
#pragma unroll 4
        for (int q = 0; q < RPW; ++q) {
            const int m = m0 + wid + 4 * q;
            const float x0 = __shfl(xn0, q, 64), x1 = __shfl(xn1, q, 64);
            if (m < g.M) embed_row(nxt, m, lane, x0, x1);
        }
    }
}

template <bool X2>
inline hipError_t launch_tail_mode(const TailArgs& g, const OutArgs& oa, const EmbedArgs& nxt, bool embed_next, int rows,
                                   hipStream_t st) {
    static DevSeen attr_seen;
    if (auto once_ = first_use_on_device(attr_seen)) {
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
