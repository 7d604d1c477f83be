// Exact-fp32 flash attention for the JMID / iMID encoder layers (nn.MultiheadAttention inside
// nn.TransformerEncoderLayer; MID/models/diffusion.py:120-125,161-166; no mask, dropout off).
//
//   JMID: one sequence per episode, S = T*K*A tokens (diffusion.py:196-204)  -> 1200 at N=5,K=20,H=12
//   iMID: one sequence per row,     S = T                                    (diffusion.py:144-147)
//
// Tokens of a sequence are contiguous rows of the packed QKV buffer [M, 3*d] (softmax attention is
// permutation-equivariant, so the (t, row) order of the reference need not be reproduced).
//
// Everything is computed TRANSPOSED so that a lane owns one query row and the softmax state is lane-local:
//   S^T = K . Q^T     A = K tile (LDS, ds_read_b128 of 4 consecutive d), B = Q (registers)
//   O^T = V^T . P^T   A = V tile (LDS, ds_read_b32, lanes along d),      B = P (the S^T accumulators themselves)
// The S^T accumulator register j of lane (q, hi) holds key (j&3)+8*(j>>2)+4*hi, so MFMA j of the PV step
// takes keys {k0, k0+4} straight from the registers - no cross-lane shuffle of P at all.
#pragma once
#include "common.hpp"
#include "gemm_f16x3.hpp"

namespace jmid {

struct AttnArgs {
    const float* QKV;  // [nseq*S, 3*d]
    float* OUT;        // [nseq*S, d]
    int S, d, nhead;
    float scale;       // 1/sqrt(head_dim)
    half_t* Ohi;       // optional: write hi/lo planes [nseq*S, d] (blocked panel layout) instead of OUT
    half_t* Olo;
};

template <int HD, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void attn_f32_kernel(AttnArgs a) {
    constexpr int KT = 32;                 // keys per tile
    constexpr int KLD = HD + 4;            // padded K row (floats)
    constexpr int NT = (HD + 31) / 32;     // 32-wide tiles of the head dim in O^T
    constexpr int NG = HD / 8;             // groups of 8 d per S^T step
    __shared__ __attribute__((aligned(16))) float Ks[KT * KLD];
    __shared__ __attribute__((aligned(16))) float Vs[KT * HD];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y, seq = blockIdx.z;
    const int S = a.S, d = a.d;
    const size_t ld = (size_t)3 * d;
    const float* base = a.QKV + (size_t)seq * S * ld;
    const int q = (blockIdx.x * NWAVES + wid) * 32 + l31;
    const int qc = q < S ? q : S - 1;

    // Q fragment: qreg[g][e] = scale * Q[q][8g + 4hi + e]
    f32x4 qreg[NG];
    {
        const float* qp = base + (size_t)qc * ld + h * HD + 4 * hi;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            f32x4 v = *reinterpret_cast<const f32x4*>(qp + 8 * g);
            qreg[g] = v * a.scale;
        }
    }

    f32x16 ot[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[n][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (S + KT - 1) / KT;
    const float* kbase = base + d + h * HD;
    const float* vbase = base + 2 * d + h * HD;
    constexpr int F4_PER_ROW = HD / 4;
    constexpr int NF4 = KT * F4_PER_ROW;

    for (int kt = 0; kt < ntiles; ++kt) {
        // ---- cooperative K/V tile load (rows past S are zero-filled)
        for (int idx = tid; idx < NF4; idx += NWAVES * 64) {
            const int row = idx / F4_PER_ROW, c4 = idx % F4_PER_ROW;
            const int key = kt * KT + row;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (key < S) {
                kv = *reinterpret_cast<const f32x4*>(kbase + (size_t)key * ld + c4 * 4);
                vv = *reinterpret_cast<const f32x4*>(vbase + (size_t)key * ld + c4 * 4);
            }
            *reinterpret_cast<f32x4*>(&Ks[row * KLD + c4 * 4]) = kv;
            *reinterpret_cast<f32x4*>(&Vs[row * HD + c4 * 4]) = vv;
        }
        __syncthreads();

        // ---- S^T = K . Q^T   (rows = keys, cols = queries)
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
        {
            const float* kp = &Ks[l31 * KLD + 4 * hi];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                f32x4 kv = *reinterpret_cast<const f32x4*>(kp + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[e], qreg[g][e], st, 0, 0, 0);
            }
        }
        // ---- online softmax over the 32 keys of this tile (16 in this lane, 16 in lane^32)
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * KT + frag_row(r, hi);
            if (key >= S) st[r] = -INFINITY;
            tmax = fmaxf(tmax, st[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = expf(st[r] - m_new);
            psum += st[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = fmaf(l_run, alpha, psum);
        m_run = m_new;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[n][r] *= alpha;

        // ---- O^T += V^T . P^T   (rows = head-dim, cols = queries)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int krow = (j & 3) + 8 * (j >> 2) + 4 * hi;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int col = n * 32 + l31;
                const float v = (HD % 32 == 0 || col < HD) ? Vs[krow * HD + (col < HD ? col : 0)] : 0.f;
                ot[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, st[j], ot[n], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- normalise and store: lane owns query q, register r of tile n is head-dim n*32 + frag_row(r, hi)
    if (q < S) {
        const float inv = 1.0f / l_run;
        float* op = a.OUT + ((size_t)seq * S + q) * d + h * HD;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int c0 = n * 32 + 8 * r4 + 4 * hi;  // 4 consecutive head-dim entries
                if (HD % 32 == 0 || c0 < HD) {
                    f32x4 v = {ot[n][4 * r4 + 0] * inv, ot[n][4 * r4 + 1] * inv, ot[n][4 * r4 + 2] * inv,
                               ot[n][4 * r4 + 3] * inv};
                    if (a.Ohi) {
                        f16x4 vh, vl;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            half_t hh, ll;
                            split_f32(v[e], hh, ll);
                            vh[e] = hh;
                            vl[e] = ll;
                        }
                        const size_t oo = blk_index(seq * S + q, h * HD + c0, d);
                        *reinterpret_cast<f16x4*>(a.Ohi + oo) = vh;
                        *reinterpret_cast<f16x4*>(a.Olo + oo) = vl;
                    } else {
                        *reinterpret_cast<f32x4*>(op + c0) = v;
                    }
                }
            }
        }
    }
}

template <int HD>
inline hipError_t launch_attn_f32_hd(const AttnArgs& a, int nseq, hipStream_t st) {
    if (a.S > 32) {
        dim3 grid((a.S + 127) / 128, a.nhead, nseq);
        hipLaunchKernelGGL((attn_f32_kernel<HD, 4>), grid, dim3(256), 0, st, a);
    } else {
        dim3 grid(1, a.nhead, nseq);
        hipLaunchKernelGGL((attn_f32_kernel<HD, 1>), grid, dim3(64), 0, st, a);
    }
    return hipGetLastError();
}

inline hipError_t launch_attn_f32(const AttnArgs& a, int nseq, int head_dim, hipStream_t st) {
    switch (head_dim) {
        case 16: return launch_attn_f32_hd<16>(a, nseq, st);
        case 32: return launch_attn_f32_hd<32>(a, nseq, st);
        case 64: return launch_attn_f32_hd<64>(a, nseq, st);
        case 128: return launch_attn_f32_hd<128>(a, nseq, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace jmid
