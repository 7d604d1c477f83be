// EXPERIMENT (diagnostics flavour, knob "attn_q64" = 1; measured slower - see the end of this comment).
// Flash attention for full F16MX launches (head_dim 128, one fp16 plane of P) with ONE wave per SIMD that owns the whole 512-entry
// register file and TWO 32-query blocks (A, B): O (128 registers) and the Q operands (128) live in the accumulation half, the scores, P
// and the K / V^T fragments in the vector half.  attn_f16x3_dma_kernel's two waves per SIMD (two workgroups per CU, 32 queries per wave)
// run their QK^T / softmax / PV phases back to back and overlap only by chance - its matrix pipes are busy 39 % of the time; here the
// softmax of tile t issues in the gaps of the matrix instructions of tile t + 1, cut into 58 "atoms" of one to four instructions.
//
// Same arithmetic, operation by operation, as attn_f16x3_dma_kernel<.., X2, MX, .., P1 = true, PF = true>: S^T = K . Q^T (K_hi . Q_hi in
// eight fp16 MFMAs, then per 64-deep block bf8(K_hi) . bf8(Q_lo) + bf8(K_lo) . bf8(Q_hi)), the online softmax in log2 units with the lazy
// reference maximum and the sums taken in register order, P rounded to nearest into one fp16 plane, O^T += V^T . P^T: the results are
// bit-identical to that kernel's (tools/attn_q64_check.hip; tests/test_gpu_parity.py::test_attention_q64_equals_the_two_wave_kernel).
//
// Per key tile t (32 keys):
//     phase I   S(t + 1) = QK^T of both blocks, the two accumulator chains interleaved  | softmax(t) of both blocks -> P; the copies of
//                                                                                        | K(t + 2) and V^T(t + 1) go out
//     phase II  O += V^T(t) . P of both blocks, eight accumulators interleaved           | the scores of tile t + 1 move to the vector half
//     s_waitcnt vmcnt(0) + s_barrier
// One barrier per tile; K tiles (fp16 K_hi + the two bf8 images, 16 KB) and V^T tiles (8 KB) in 2-stage rings: at the barrier every
// wave is past K(t + 1) and V^T(t), so the copies of K(t + 2) / V^T(t + 1), issued at the top of the next tile, have a whole tile to land.
//
// Measured (tools/attn_q64_check.hip, 51 sequences of 1200 tokens, random planes, alternating launches): 0.305 ms per launch against
// 0.285 ms of the two-wave kernel.  In-kernel cycle stamps: phase I 2 000 cycles for 1 024 cycles of matrix instructions, phase II 650
// for 512, barrier 330 - a lone in-order wave issues its ~290 non-matrix instructions of phase I at ~7 cycles each, and hipcc's
// placement cannot be pinned finer than the source order between sched_barriers.  What the work left behind in the shipped kernel:
// the lazy reference maximum (with O in accumulation registers a rescale costs 192 instructions per block, which is what made its
// frequency visible), the one-instruction cross-half exchange (half_swap), the mf-outermost order of the PV instructions.
#pragma once
#include "attn_f16x3.hpp"

namespace jmid {

constexpr int AQ_KSTAGE = 8192;                       // halfs per K stage: K_hi plane 8 KB, bf8(K_hi) 4 KB, bf8(K_lo) 4 KB
constexpr int AQ_VSTAGE = 4096;                       // halfs per V^T stage: the hi plane, 128 rows x 32 keys
constexpr size_t AQ_LDS = size_t(2 * AQ_KSTAGE + 2 * AQ_VSTAGE) * sizeof(half_t);      // 48 KB

// compile-time loops: every index below (register of the scores, atom of the softmax, gap of the matrix-instruction stream) is a
// constant expression - a run-time index into the score registers sends them to scratch memory
template <int... I, typename F>
__device__ __forceinline__ void aq_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void aq_for(F&& f) { aq_for_impl(std::make_integer_sequence<int, N>{}, f); }
// the 58 softmax atoms of a key tile (29 per block, interleaved A0 B0 A1 B1 ...) over the 24 gaps of the score phase: two per fp16 matrix
// instruction (32 cycles), 4 3 3 3 4 3 3 3 over the bf8 ones (64 cycles)
constexpr int aq_gap_first(int g) {
    constexpr int wide[9] = {0, 4, 7, 10, 13, 17, 20, 23, 26};
    return g < 16 ? 2 * g : 32 + wide[g - 16];
}

static __global__ __launch_bounds__(256, 1) void attn_q64_kernel(AttnHArgs a, int nqt) {
    constexpr int HD = 128, KT = 32, NT = 4, NKS = 8;
    args_now_each(a, nqt);
    extern __shared__ __attribute__((aligned(16))) unsigned char att_lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(att_lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // XCD-aware order: the q-tiles of one (sequence, head) share K / V^T, keep them on one XCD's L2
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const int sh = fast_div(swz, nqt, a.mq), qt = swz - sh * nqt;
    const int seq = fast_div(sh, a.nhead, a.mh), h = sh - seq * a.nhead;
    const int S = a.S, d = a.d;
    const size_t tok0 = (size_t)seq * S;
    const int qbase = (qt * 4 + wid) * 64;
    // a wave whose 64 queries all lie past the end of the sequence keeps copying its share of every tile and meets the barriers
    const bool wave_idle = qbase >= S;
    int q[2], qc[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        q[blk] = qbase + 32 * blk + l31;
        qc[blk] = q[blk] < S ? q[blk] : S - 1;
    }

    // ---- Q operands of both blocks: fp16 Q_hi fragments, bf8 images of Q_hi (made here) and of Q_lo (written by the in_proj GEMM)
    f16x8 qh[2][NKS];
    i32x8 q8h[2][2], q8l[2][2];
    i32x4 q8raw[2][2][4], q8lraw[2][2][2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const size_t o = (tok0 + qc[blk]) * d + h * HD + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qh[blk][ks] = *reinterpret_cast<const f16x8*>(a.Qhi + o + 16 * ks);
        const size_t o8 = (tok0 + qc[blk]) * d + h * HD + 32 * hi;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                q8raw[blk][kb][c] = __builtin_bit_cast(i32x4, *reinterpret_cast<const f16x8*>(a.Qhi + o8 + 64 * kb + 8 * c));
            q8lraw[blk][kb][0] = *reinterpret_cast<const i32x4*>(a.Q8l + o8 + 64 * kb);
            q8lraw[blk][kb][1] = *reinterpret_cast<const i32x4*>(a.Q8l + o8 + 64 * kb + 16);
        }
    }

    // ---- DMA sources: wave-uniform base + 32-bit lane offset, as in attn_f16x3_dma_kernel (K rounds: fp16 plane rows 0-15 / 16-31,
    // then the two whole bf8 images; V^T rounds: rows 0-63 / 64-127 of the hi plane); the last tile's rows past S / chunks past Spad
    // are clamped to valid memory (those keys are masked)
    const size_t vt0 = ((size_t)seq * a.nhead + h) * HD * a.Spad;
    const int k_row = tid >> 4, k_c = (tid & 15) ^ (k_row & 15);
    const int v_row = tid >> 2, v_c = (tid & 3) ^ ((v_row >> 2) & 3);
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);
    const char* const kh_b = reinterpret_cast<const char*>(a.Khi + tok0 * d + h * HD);
    const char* const k8h_b = reinterpret_cast<const char*>(a.K8h) + (tok0 * d + h * HD);
    const char* const k8l_b = reinterpret_cast<const char*>(a.K8l) + (tok0 * d + h * HD);
    const char* const vth_b = reinterpret_cast<const char*>(a.Vthi + vt0);
    const int ntiles = (S + KT - 1) / KT, last_tile = ntiles - 1;
    const int rows_last = S - last_tile * KT - 1;
    const int chunks_last = a.Spad / 8 - 1 - last_tile * 4;
    auto rowc = [&](int r) { return r < rows_last ? r : rows_last; };
    const int swz8 = ((tid & 7) ^ (((tid >> 3) >> 1) & 7)) << 4;
    unsigned offK16[2] = {(unsigned)(k_row * d + k_c * 8) * 2u, (unsigned)((16 + k_row) * d + k_c * 8) * 2u};
    unsigned offK8 = (unsigned)((tid >> 3) * d + swz8);
    unsigned offV = (unsigned)(v_row * a.Spad + v_c * 8) * 2u;
    const unsigned offK16_last[2] = {(unsigned)(rowc(k_row) * d + k_c * 8) * 2u, (unsigned)(rowc(16 + k_row) * d + k_c * 8) * 2u};
    const unsigned offK8_last = (unsigned)(rowc(tid >> 3) * d + swz8);
    const unsigned offV_last = (unsigned)(v_row * a.Spad + (v_c < chunks_last ? v_c : chunks_last) * 8) * 2u;
    auto dma16 = [](const char* s, half_t* dd) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s, (__attribute__((address_space(3))) void*)dd, 16, 0, 0);
    };
    // copy i (0 ... 5) of key tile kt: this wave's 1 KB of K_hi rows 0-15, rows 16-31, bf8(K_hi), bf8(K_lo) into K stage kt % 2, of
    // V^T rows 0-63, 64-127 into V stage kt % 2
    auto issue_one = [&](int kt, int i) {
        half_t* const ks = lds + (kt & 1) * AQ_KSTAGE + wid_s * 512;
        if (i < 2) dma16(kh_b + (size_t)kt * (KT * d) * 2 + offK16[i], ks + i * 2048);
        else if (i < 4) dma16((i == 2 ? k8h_b : k8l_b) + (size_t)kt * (KT * d) + offK8, ks + i * 2048);
        else dma16(vth_b + (size_t)(64 * (i - 4)) * a.Spad * 2 + (size_t)kt * 64 + offV, lds + 2 * AQ_KSTAGE + (kt & 1) * AQ_VSTAGE + (i - 4) * 2048 + wid_s * 512);
    };
    // the last tile's copies use the clamped offsets.  K runs one tile ahead of V^T: the K offsets switch with K(last), the V^T offset
    // with V^T(last)
    auto use_last_k = [&]() {
        offK16[0] = offK16_last[0];
        offK16[1] = offK16_last[1];
        offK8 = offK8_last;
    };
    auto use_last_v = [&]() { offV = offV_last; };
    auto issue_k = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_one(kt, i);
    };
    auto issue_v = [&](int kt) {
#pragma unroll
        for (int i = 4; i < 6; ++i) issue_one(kt, i);
    };
    if (last_tile == 0) { use_last_k(); use_last_v(); }
    issue_k(0);
    issue_v(0);
    if (ntiles > 1) {
        if (last_tile == 1) use_last_k();
        issue_k(1);
    }

    // ---- Q conversions under the flight of the first copies
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                q8h[blk][kb][2 * c] = bf8_of_f16x4(q8raw[blk][kb][c][0], q8raw[blk][kb][c][1]);
                q8h[blk][kb][2 * c + 1] = bf8_of_f16x4(q8raw[blk][kb][c][2], q8raw[blk][kb][c][3]);
            }
            const i32x4 l0 = q8lraw[blk][kb][0], l1 = q8lraw[blk][kb][1];
            q8l[blk][kb] = i32x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        }
    // The Q operands (128 registers) go to the accumulation half of the register file next to O (128): matrix instructions read
    // their B operand from there directly, and the vector half keeps the scores, P and the K / V^T fragments.  (Pinned: left alone,
    // hipcc parks them there anyway and copies four registers back in front of every matrix instruction.)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
#ifndef AQ_NOPIN
        for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+a"(qh[blk][ks]));
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) asm volatile("" : "+a"(q8h[blk][kb]), "+a"(q8l[blk][kb]));
#endif
    }

    f32x16 ot[2][NT];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[blk][n][r] = 0.f;
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};     // running max in log2 units (Q is pre-scaled)

    // fragment read offsets (halfs / bytes), as in attn_f16x3_dma_kernel
    const int kbase = l31 * 128, kx = l31 & 15;
    const int r8b = l31 * 128, sw8 = (l31 >> 1) & 7;
    int vbase[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) vbase[mf] = l31 * 32 + (((2 * mf + hi) ^ ((l31 >> 2) & 3)) * 8);

    f32x16 sc[2];            // the scores of tile t, being turned into P
    u32x4 phq[2][2];         // P of tile t as packed fp16 pairs: [block][16-key half]
    float alpha[2] = {1.f, 1.f};
    bool rescale[2] = {false, false};
    float tmax_[2], psum_[2], mnew_[2], xa_[2], xb_[2];

    // ---- the online softmax of block `blk` over sc[blk], cut into 29 atoms of one to four instructions for the gaps between the
    // matrix instructions of the other tile (the arithmetic of attn_f16x3_dma_kernel, operation by operation):
    //   0-3    (last tile: keys past S masked, in atom 0) the maximum of the lane's 16 scores, four at a time
    //   4      the cross-half exchange of that maximum
    //   5      the row's tile maximum; the reference maximum moves only when it is exceeded by more than ATT_LAZY_TAU
    //   6      alpha, the does-anybody-rescale vote
    //   7-22   one exponential each, summed in register order
    //   23     the cross-half exchange of the sum
    //   24     running sum and maximum
    //   25-28  P rounded to nearest into fp16 pairs, two dwords each
    constexpr int NATOM = 29;
    auto atom = [&](auto blk_c, auto idx_c, auto last_c, int kt) {
        constexpr int blk = decltype(blk_c)::value, idx = decltype(idx_c)::value;
        constexpr bool last = decltype(last_c)::value;
        f32x16& s = sc[blk];
        if constexpr (idx < 4) {
            if constexpr (idx == 0 && last) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * KT + frag_row(r, hi) >= S) s[r] = -INFINITY;
            }
            float t = idx == 0 ? s[0] : tmax_[blk];
#pragma unroll
            for (int r = (idx == 0 ? 1 : 4 * idx); r < 4 * idx + 4; ++r) t = fmaxf(t, s[r]);
            tmax_[blk] = t;
        } else if constexpr (idx == 4) {
            half_swap(tmax_[blk], xa_[blk], xb_[blk]);
        } else if constexpr (idx == 5) {
            const float tm = fmaxf(xa_[blk], xb_[blk]);
            mnew_[blk] = att_lazy_max(m_run[blk], tm);
        } else if constexpr (idx == 6) {
            alpha[blk] = __builtin_amdgcn_exp2f(m_run[blk] - mnew_[blk]);
            rescale[blk] = !__all(mnew_[blk] == m_run[blk]);
            psum_[blk] = 0.f;
        } else if constexpr (idx < 23) {
            // element r: its exponential; the subtraction of element r + 2 and the addition of element r - 2 ride along, so that no
            // instruction waits for the one before it (a lone wave has nobody to hide a dependent pair behind).  Same operations, and
            // the sum still runs in register order.
            constexpr int r = idx - 7;
            if constexpr (r == 0) {
                s[0] -= mnew_[blk];
                s[1] -= mnew_[blk];
            }
            s[r] = __builtin_amdgcn_exp2f(s[r]);
            if constexpr (r + 2 < 16) s[r + 2] -= mnew_[blk];
            if constexpr (r >= 2) psum_[blk] += s[r - 2];
        } else if constexpr (idx == 23) {
            psum_[blk] += s[14];
            psum_[blk] += s[15];
            half_swap(psum_[blk], xa_[blk], xb_[blk]);
        } else if constexpr (idx == 24) {
            l_run[blk] = fmaf(l_run[blk], alpha[blk], xa_[blk] + xb_[blk]);
            m_run[blk] = mnew_[blk];
        } else {
            constexpr int j = idx - 25, mf = j >> 1, i0 = 2 * (j & 1);
#pragma unroll
            for (int i = i0; i < i0 + 2; ++i) phq[blk][mf][i] = pk_f16_rne(s[8 * mf + 2 * i], s[8 * mf + 2 * i + 1]);
        }
    };
    // atoms first ... first + count - 1 of the two blocks' interleaved list (neighbours are independent)
    auto atoms = [&](auto first_c, auto count_c, auto last_c, int kt) {
        constexpr int first = decltype(first_c)::value, count = decltype(count_c)::value;
        aq_for<count>([&](auto j_c) {
            constexpr int k = first + decltype(j_c)::value;
            if constexpr (k < 2 * NATOM) atom(std::integral_constant<int, (k & 1)>{}, std::integral_constant<int, (k >> 1)>{}, last_c, kt);
        });
    };
    // The O accumulators and the Q operands live in the accumulation half of the register file; the pins keep the rare rescale from
    // dragging all 128 accumulators through vector registers on EVERY key tile (hipcc merges the two paths with copies at the top of
    // the loop, and hoists plain reads above the branch)
    auto pin_o = [&](int blk) {
#pragma unroll
        for (int n = 0; n < NT; ++n) asm volatile("" : "+a"(ot[blk][n]));
    };
    auto rescale_o = [&](int blk) {
        if (rescale[blk]) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                asm volatile("" : "+a"(ot[blk][n]));
                f32x16 t = ot[blk][n];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] *= alpha[blk];
                ot[blk][n] = t;
                asm volatile("" : "+a"(ot[blk][n]));
            }
        }
    };

    // S^T of BOTH blocks against the K tile in stage `Ks`, the two accumulator chains interleaved (a lone wave cannot hide the latency of
    // a chain of dependent matrix instructions behind anything else) and sharing every K fragment: per accumulator eight fp16 steps, then the
    // two bf8 terms of each 64-deep block.  gap(g), g = 0 ... 23, is the vector work issued after matrix instruction g.
    auto qk2 = [&](const half_t* Ks, f32x16 (&sn)[2], auto&& gap) {
        constexpr int PFD = 3;
        const unsigned char* k8 = reinterpret_cast<const unsigned char*>(Ks + 4096);
        auto kread = [&](int ks) { return *reinterpret_cast<const f16x8*>(Ks + kbase + (((2 * ks + hi) ^ kx) << 3)); };
        auto k8read = [&](int img, int kb, int c) {
            return *reinterpret_cast<const i32x4*>(k8 + img * 4096 + r8b + (((kb * 4 + hi * 2 + c) ^ sw8) << 4));
        };
        f32x16 sa, sb;
#pragma unroll
        for (int r = 0; r < 16; ++r) sa[r] = 0.f, sb[r] = 0.f;
        f16x8 kf[NKS];
        i32x4 k8f[2][2][2];      // [kb][image][chunk]
#pragma unroll
        for (int i = 0; i < PFD; ++i) kf[i] = kread(i);
        aq_for<NKS>([&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value;
            if constexpr (ks + PFD < NKS) kf[ks + PFD] = kread(ks + PFD);
            if constexpr (ks == 5 || ks == 7) {
                constexpr int kb = ks == 5 ? 0 : 1;
                k8f[kb][0][0] = k8read(0, kb, 0); k8f[kb][0][1] = k8read(0, kb, 1);
                k8f[kb][1][0] = k8read(1, kb, 0); k8f[kb][1][1] = k8read(1, kb, 1);
            }
            sa = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qh[0][ks], sa, 0, 0, 0);
            gap(std::integral_constant<int, 2 * ks>{});
            __builtin_amdgcn_sched_barrier(0);
            sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qh[1][ks], sb, 0, 0, 0);
            gap(std::integral_constant<int, 2 * ks + 1>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        aq_for<2>([&](auto kb_c) {
            constexpr int kb = decltype(kb_c)::value;
            const i32x4 h0 = k8f[kb][0][0], h1 = k8f[kb][0][1], l0 = k8f[kb][1][0], l1 = k8f[kb][1][1];
            const i32x8 kh8 = i32x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            const i32x8 kl8 = i32x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            sa = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kh8, q8l[0][kb], sa, 1, 1, 0, 0, 0, 0);
            gap(std::integral_constant<int, 16 + 4 * kb>{});
            __builtin_amdgcn_sched_barrier(0);
            sb = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kh8, q8l[1][kb], sb, 1, 1, 0, 0, 0, 0);
            gap(std::integral_constant<int, 17 + 4 * kb>{});
            __builtin_amdgcn_sched_barrier(0);
            sa = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kl8, q8h[0][kb], sa, 1, 1, 0, 0, 0, 0);
            gap(std::integral_constant<int, 18 + 4 * kb>{});
            __builtin_amdgcn_sched_barrier(0);
            sb = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kl8, q8h[1][kb], sb, 1, 1, 0, 0, 0, 0);
            gap(std::integral_constant<int, 19 + 4 * kb>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        sn[0] = sa;
        sn[1] = sb;
    };
    // O^T of both blocks += V^T tile in stage `Vs` . P^T: the eight accumulators interleaved (16-key half mf outermost, so that the two
    // matrix instructions of an accumulator are eight apart) and sharing every V^T fragment; gap(g), g = 0 ... 15
    auto pv2 = [&](const half_t* Vs, auto&& gap) {
        constexpr int PFD = 3;
        auto vread = [&](int i) { return *reinterpret_cast<const f16x8*>(Vs + (i & 3) * 1024 + vbase[i >> 2]); };      // i = 4 mf + n
        f16x8 vf[2 * NT];
#pragma unroll
        for (int i = 0; i < PFD; ++i) vf[i] = vread(i);
        aq_for<2 * NT>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value, n = i & 3, mf = i >> 2;
            if constexpr (i + PFD < 2 * NT) vf[i + PFD] = vread(i + PFD);
            ot[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[i], __builtin_bit_cast(f16x8, phq[0][mf]), ot[0][n], 0, 0, 0);
            gap(std::integral_constant<int, 2 * i>{});
            __builtin_amdgcn_sched_barrier(0);
            ot[1][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[i], __builtin_bit_cast(f16x8, phq[1][mf]), ot[1][n], 0, 0, 0);
            gap(std::integral_constant<int, 2 * i + 1>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto nogap = [](auto) {};

#ifdef AQ_TRACE      // (tools/attn_q64_check.hip: cycles per phase of the key-tile loop, accumulated per wave into a.Opart)
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define AQ_STAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tprev; tprev = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define AQ_STAMP(i)
#endif

    // ---- prologue: tiles 0 (K, V^T) and 1 (K) have landed for everybody; the scores of tile 0; a second barrier frees K stage 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (!wave_idle) qk2(lds, sc, nogap);
    asm volatile("" : "+v"(sc[0]), "+v"(sc[1]));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // body of key tile kt: phase I = the scores of tile kt + 1 | the softmax of tile kt; phase II = O += V^T(kt) . P | the copies of
    // K(kt + 2) and V^T(kt + 1), whose stages every wave left before the last barrier
    auto body = [&](int kt, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        const bool more_k = kt + 2 < ntiles, more_v = kt + 1 < ntiles;
        if (more_k && kt + 2 == last_tile) use_last_k();      // (uniform)
        if (more_v && kt + 1 == last_tile) use_last_v();
        if (wave_idle) {
            if (more_k) issue_k(kt + 2);
            if (more_v) issue_v(kt + 1);
        } else {
            pin_o(0);
            pin_o(1);
            f32x16 sn[2];
            if (!LAST) {
                qk2(lds + ((kt + 1) & 1) * AQ_KSTAGE, sn, [&](auto g_c) {
                    constexpr int g = decltype(g_c)::value;
                    atoms(std::integral_constant<int, aq_gap_first(g)>{}, std::integral_constant<int, aq_gap_first(g + 1) - aq_gap_first(g)>{},
                          std::false_type{}, kt);
                    // the copies of K(kt + 2) (four) and V^T(kt + 1) (two): their stages were left by everybody before the last barrier, and
                    // issued here they have the whole tile to land
                    if constexpr ((g & 1) && g < 12) {
                        constexpr int i = g >> 1;
                        if constexpr (i < 4) { if (more_k) issue_one(kt + 2, i); }
                        else if (more_v) issue_one(kt + 1, i);
                    }
                });
            } else {
                atoms(std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * NATOM>{}, std::true_type{}, kt);
            }
            AQ_STAMP(0)
            rescale_o(0);
            rescale_o(1);
            AQ_STAMP(1)
            // the scores of tile kt + 1 move from the accumulation half (where hipcc lets the matrix instructions build them) to the
            // vector half two registers per gap
            pv2(lds + 2 * AQ_KSTAGE + (kt & 1) * AQ_VSTAGE, [&](auto g_c) {
                constexpr int g = decltype(g_c)::value;
                if constexpr (!LAST) {
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk) {
                        float t = sn[blk][g];
                        asm volatile("" : "+v"(t));
                        sc[blk][g] = t;
                    }
                }
            });
            AQ_STAMP(2)
        }
        if (!LAST) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's copies of K(kt + 2) and V^T(kt + 1) have landed
            AQ_STAMP(3)
            __builtin_amdgcn_s_barrier();                          // ... and everybody else's; everybody is past K(kt + 1) and V^T(kt)
            __builtin_amdgcn_sched_barrier(0);
            AQ_STAMP(4)
        }
    };
    for (int kt = 0; kt < last_tile; ++kt) body(kt, std::false_type{});
    body(last_tile, std::true_type{});

#ifdef AQ_TRACE
    if (lane == 0 && a.Opart) {
        unsigned long long* t = reinterpret_cast<unsigned long long*>(a.Opart) + ((size_t)blockIdx.x * 4 + wid) * 8;
        for (int i = 0; i < 8; ++i) t[i] = tacc[i];
    }
#endif
    // ---- normalise, round to the fp16 plane and store: register r of tile n is head dim n * 32 + frag_row(r, hi)  (blocked panel layout)
    bool overflow = false;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        if (q[blk] < S) {
            const float inv = 1.0f / l_run[blk];
            const int orow = (int)tok0 + q[blk];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int c0 = n * 32 + 8 * r4 + 4 * hi;
                    f16x4 vh;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = ot[blk][n][4 * r4 + e] * inv;
                        half_t hh, ll;
                        split_f32(v, hh, ll);
                        overflow |= !(fabsf(v) <= kHalfMax);
                        vh[e] = hh;
                    }
                    *reinterpret_cast<f16x4*>(a.Ohi + blk_index(orow, h * HD + c0, d)) = vh;
                }
        }
    }
    if (overflow) atomicOr(a.range_flag, 1);
}

// does this launch take the one-wave-per-SIMD kernel?  Only on request ("attn_q64" = 1): F16MX with the bf8 images of K and Q_lo (the
// mode's default operands) and no key split
inline bool attn_q64_applies(const AttnHArgs& a, int nseq) {
    (void)nseq;
    return tune().attn_q64 == 1 && a.x2 && a.K8h && a.K8l && a.Q8l && a.nsplit == 1 && tune().attn_mx == 0;
}
inline hipError_t launch_attn_q64(const AttnHArgs& a_in, int nseq, hipStream_t st) {
    AttnHArgs a = a_in;
    const int nqt = (a.S + 255) / 256;
    const dim3 grid(nqt * a.nhead * nseq);
    a.mq = fast_div_magic(nqt, grid.x);
    a.mh = fast_div_magic(a.nhead, grid.x);
    a.nseq = nseq;
    static DevSeen seen;
    if (auto once_ = first_use_on_device(seen))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_q64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AQ_LDS);
    hipLaunchKernelGGL(attn_q64_kernel, grid, dim3(256), AQ_LDS, st, a, nqt);
    return hipGetLastError();
}

}  // namespace jmid
