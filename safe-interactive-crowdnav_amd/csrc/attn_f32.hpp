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
#include <algorithm>
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


// ---------------------------------------------------------------------------------------------------------------
// Short sequences (iMID: S = T <= 16): G = several sequences share one wave's 32 x 32 score tile, block-diagonally
// masked, and K / V tiles are private to the wave (no workgroup barrier; 4 waves of a workgroup = 4 heads).
// The result of a sequence must not depend on which slot of the tile it lands in (chunking-invariance is tested
// bit for bit), so slot s puts its local key k on accumulator register j = s*J + k/2 of lane-half hi = k%2
// (J = ceil(S/2)): every slot then adds its keys in the same order, and the masked positions contribute exact zeros.
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_f32_packed_kernel(AttnArgs a, int nseq, int G, int J) {
    constexpr int KLD = HD + 4;            // padded K / V row (floats)
    constexpr int NT = (HD + 31) / 32;
    constexpr int NG = HD / 8;
    constexpr int F4 = HD / 4;             // float4 per row
    constexpr int NLD = (32 * F4 + 63) / 64;   // float4 loads per lane and tile
    extern __shared__ __attribute__((aligned(16))) float att_pk_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y * 4 + wid;
    if (h >= a.nhead) return;              // no workgroup barrier below: a whole wave may leave
    float* Ks = att_pk_lds + wid * 32 * KLD;
    const int S = a.S, d = a.d;
    const size_t ld = (size_t)3 * d;
    const int seq0 = blockIdx.x * G;
    const int nvs = (nseq - seq0) < G ? (nseq - seq0) : G;      // sequences in this tile

    // this lane's query: column l31 = slot sq, local token kq
    const int sq = l31 / S, kq = l31 - sq * S;
    const bool qvalid = sq < nvs;
    const size_t qtok = (size_t)(seq0 + (qvalid ? sq : 0)) * S + (qvalid ? kq : 0);
    f32x4 qreg[NG];
    {
        const float* qp = a.QKV + qtok * ld + h * HD + 4 * hi;
#pragma unroll
        for (int g = 0; g < NG; ++g) qreg[g] = *reinterpret_cast<const f32x4*>(qp + 8 * g) * a.scale;
    }
    // tile row rho holds key (slot, k): rho = frag_row(j, hh) with j = slot*J + k/2, hh = k%2
    auto load_tile = [&](int col0, f32x4 (&regs)[NLD]) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = i * 64 + lane;
            const int rho = idx / F4, c4 = idx - rho * F4;
            const int j = ((rho >> 3) << 2) | (rho & 3), hh = (rho >> 2) & 1;
            const int sl = j / J, k = (j - sl * J) * 2 + hh;
            regs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < 32 * F4 && sl < nvs && k < S)
                regs[i] = *reinterpret_cast<const f32x4*>(a.QKV + ((size_t)(seq0 + sl) * S + k) * ld + col0 + h * HD + c4 * 4);
        }
    };
    auto store_tile = [&](const f32x4 (&regs)[NLD]) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = i * 64 + lane;
            const int rho = idx / F4, c4 = idx - rho * F4;
            if (idx < 32 * F4) *reinterpret_cast<f32x4*>(&Ks[rho * KLD + c4 * 4]) = regs[i];
        }
    };
    f32x4 kreg[NLD], vreg[NLD];
    load_tile(d, kreg);
    store_tile(kreg);
    load_tile(2 * d, vreg);               // in flight during the S^T MFMAs

    // ---- S^T = K . Q^T
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    {
        const float* kp = &Ks[l31 * KLD + 4 * hi];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            f32x4 kv = *reinterpret_cast<const f32x4*>(kp + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[e], qreg[g][e], st, 0, 0, 0);
        }
    }
    // ---- block-diagonal mask + softmax (single tile: no running state)
    float tmax = -INFINITY;
    {
        int sl = 0, jj = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = qvalid && sl == sq && (jj * 2 + hi) < S;
            if (!ok) st[r] = -INFINITY;
            tmax = fmaxf(tmax, st[r]);
            if (++jj == J) {
                jj = 0;
                ++sl;
            }
        }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    if (tmax == -INFINITY) tmax = 0.f;    // column without a query: P = 0, nothing is stored
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        st[r] = expf(st[r] - tmax);
        psum += st[r];
    }
    psum += __shfl_xor(psum, 32, 64);

    // ---- O^T = V^T . P^T : V goes into the same LDS tile (the K reads above are complete: their values were used)
    store_tile(vreg);
    f32x16 ot[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[n][r] = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int krow = (j & 3) + 8 * (j >> 2) + 4 * hi;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int col = n * 32 + l31;
            const float v = (HD % 32 == 0 || col < HD) ? Ks[krow * KLD + (col < HD ? col : 0)] : 0.f;
            ot[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, st[j], ot[n], 0, 0, 0);
        }
    }
    if (qvalid) {
        const float inv = 1.0f / psum;
        float* op = a.OUT ? a.OUT + qtok * d + h * HD : nullptr;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int c0 = n * 32 + 8 * r4 + 4 * hi;
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
                        const size_t oo = blk_index((int)qtok, h * HD + c0, d);
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
    if (a.S <= 16 && tune().attn_pack) {
        const int J = (a.S + 1) / 2;
        const int G = std::min(32 / a.S, 16 / J);
        const size_t lds = size_t(4) * 32 * (HD + 4) * sizeof(float);
        static DevSeen attr_seen;
        if (auto once_ = first_use_on_device(attr_seen)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32_packed_kernel<HD>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        dim3 grid((nseq + G - 1) / G, (a.nhead + 3) / 4);
        hipLaunchKernelGGL((attn_f32_packed_kernel<HD>), grid, dim3(256), lds, st, a, nseq, G, J);
        return hipGetLastError();
    }
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
