// EXPERIMENT (-DJMID_EXPERIMENTS builds, knob "attn_pp" = 1; bit-identical to the shipped kernel, measured NO FASTER - see the end).
//
// Head-dim-128 flash attention as an 8-wave PING-PONG: the same arithmetic as attn_f16x3_dma_kernel (attn_f16x3.hpp: transposed
// formulation, split-fp16 operands, LDS-DMA K / V^T tiles, lazy reference maximum), re-timed the way the CDNA4 guide describes its
// tuned 8-wave attention.  The shipped kernel runs two independent 4-wave workgroups per CU; a wave walks QK^T -> softmax -> P.V per
// key tile in order, and whether its matrix instructions meet the partner wave's softmax or the partner's matrix instructions on
// their shared SIMD is left to chance (matrix pipes 46 % busy).  Here a workgroup is 8 waves = 256 queries, wave w and wave w + 4
// share a SIMD, and the key-tile loop is cut into two kinds of segments separated by s_barrier:
//
//     compute segment   P.V of tile t - 1 and QK^T of tile t: matrix instructions and LDS fragment reads only; the score chain
//                       (one accumulator, dependent) alternates with the four independent O accumulators
//     vector  segment   the LDS-DMA copies of later tiles FIRST, softmax of tile t (fp32, lane-local), P packed to fp16, the first
//                       fragment reads of the coming compute segment, wait for the copies
//
//   segment      2t                                          2t + 1
//   group 0      C: P.V(t-1), QK^T(t)                        V: copies (rest of K(t+2)); softmax(t); pre-read V(t), K(t+1)
//   group 1      V: copies (V(t+1), half of K_hi(t+2));      C: P.V(t-1), QK^T(t)
//                   softmax(t-1); pre-read V(t-1), K(t)
//
// Ring: three stages of K and V^T (stage = tile % 3), so that everything a vector segment copies is first read two segments after
// the barrier that publishes it.  Per accumulator the matrix instructions are the ones of attn_f16x3_dma_kernel in the same order:
// results are bit-identical in F16MX, F16X2 and F16X3, with and without a key split (tools/attn_pp_check.hip,
// tests/test_gpu_parity.py::test_attention_pingpong_equals_the_two_wave_kernel).
//
// MEASURED (51 sequences of 1 200 tokens, random planes, alternating launches, warm clocks; tools/attn_pp_check.hip):
//     F16MX 0.272 ms against 0.250 (shipped kernel)    F16X2 0.289 against 0.300    F16X3 0.449 against 0.426
// and with cycle stamps per kind of segment (-DATT_PP_TRACE; F16MX, cycles per wave and key tile):
//     compute segment 905-919 (768 of them matrix instructions)   vector segment 960-1 080   wait for the copies 480-500
//     barrier 510-740   total 3 000-3 160   (shipped kernel, stamped the same way: 2 660)
// What the stamps say: (1) the segments do what they were built for - a compute segment keeps its SIMD's matrix pipe 84 % busy;
// (2) the vector segment is as long as the compute segment although it holds ~100 vector instructions: beside a partner that issues
// matrix instructions back to back a wave gets ~5 of 8 issue slots per 32 cycles (17 v_exp_f32 at four slots + 83 others = 151
// slots = 966 cycles: the measured number) - the softmax is bound by ISSUE SLOTS, and no placement of it changes their count;
// (3) an LDS-DMA copy needs ~1 500 cycles from issue to landed under this load (most K / V^T lines come from the Infinity Cache), half
// a segment more than the vector segment lasts; (4) deferring the wait to the end of the wave's NEXT compute segment - what the third
// ring stage was built for (-DATT_PP_LATE_WAIT) - makes that compute segment 1 667 cycles long: the wait for the copies' remaining
// flight sits inside it then (not stalled fragment reads: tools/attn_issue_probe.hip on the two-wave kernel).  With no copies at all
// in the loop (-DATT_PP_NO_DMA, wrong results) a tile still takes 2 470-2 600 cycles: two serialised segments of ~960 plus ~250
// per barrier.  The shipped kernel's 2 660 cycles per tile is within 8 % of what this structure can reach; what would move both is
// fewer vector instructions per key (docs/NOTEBOOK.md section 10).
#pragma once

namespace jmid {

constexpr size_t ATT_PP_LDS = size_t(3) * ATT_STAGE * sizeof(half_t);      // 96 KB: three stages of K and V^T

template <bool X2, bool MX>
__global__ __launch_bounds__(512, 1) void attn_pp_kernel(AttnHArgs a, int nqt) {
    constexpr int HD = 128, KT = 32, NT = 4, NKS = 8;
    constexpr bool P1 = X2;          // F16X2 / F16MX: one fp16 plane of P (the production setting of the two-wave kernel)
    static_assert(!MX || X2, "F16MX implies the F16X2 operand set");
    args_now_each(a, nqt);
    extern __shared__ __attribute__((aligned(16))) unsigned char att_lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(att_lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, t256 = tid & 255;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // (said to the compiler: everything derived from it is wave-uniform)
    const int l31 = lane & 31, hi = lane >> 5;
    const int grp = wid >> 2, w4 = wid & 3;
    // XCD-aware order: the q-tiles of one (sequence, head) share K/V, keep them on one XCD's L2
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const int sh0 = fast_div(swz, nqt, a.mq), qt = swz - sh0 * nqt;
    const int sh = fast_div(sh0, a.nsplit, a.ms), split = sh0 - sh * a.nsplit;
    const int seq = fast_div(sh, a.nhead, a.mh), h = sh - seq * a.nhead;
    const int S = a.S, d = a.d;
    const size_t tok0 = (size_t)seq * S;
    const int q = (qt * 8 + wid) * 32 + l31;
    const int qc = q < S ? q : S - 1;
    const bool wave_idle = (qt * 8 + wid) * 32 >= S;      // copies and barriers only

    // ---- Q operands (B operand of S^T = K . Q^T): raw loads now, conversions after the first copies are out
    f16x8 qh[NKS], ql[MX ? 1 : NKS];
    i32x8 q8h[2], q8l[2];
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
            if (a.Q8l) {
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
    float m_run = -INFINITY, l_run = 0.f;   // running reference maximum / row sum, log2 units (Q is pre-scaled)

    // ---- DMA sources: the layout of attn_f16x3_dma_kernel with a 256-thread half of the workgroup in the place of its workgroup
    //      piece i of a tile: 0, 1 = K_hi rows 0-15 / 16-31; 2, 3 = K_lo (MX: the bf8 images of K_hi, K_lo); 4, 5 = V^T_hi rows 0-63 / 64-127;
    //      6, 7 = V^T_lo (F16X3 only); each piece is 4 KB = one wave-instruction of each of the four waves of a group
    const half_t* kh_g = a.Khi + tok0 * d + h * HD;
    const half_t* kl_g = a.Klo + tok0 * d + h * HD;
    const size_t vt0 = ((size_t)seq * a.nhead + h) * HD * a.Spad;
    const int k_row = t256 >> 4, k_c = (t256 & 15) ^ (k_row & 15);
    const int v_row = t256 >> 2, v_c = (t256 & 3) ^ ((v_row >> 2) & 3);
    const int last_vchunk = a.Spad / 8 - 1;
    const char* const kh_b = reinterpret_cast<const char*>(kh_g);
    const char* const kl_b = reinterpret_cast<const char*>(kl_g);
    const char* const k8h_b = reinterpret_cast<const char*>(a.K8h) + (tok0 * d + h * HD);
    const char* const k8l_b = reinterpret_cast<const char*>(a.K8l) + (tok0 * d + h * HD);
    const char* const vth_b = reinterpret_cast<const char*>(a.Vthi + vt0);
    const char* const vtl_b = reinterpret_cast<const char*>(a.Vtlo + vt0);
    const int last_tile = (S + KT - 1) / KT - 1;
    const int rows_last = S - last_tile * KT - 1;
    const int chunks_last = last_vchunk - last_tile * 4;
    auto rowc = [&](int r) { return r < rows_last ? r : rows_last; };
    unsigned offK16[2] = {(unsigned)(k_row * d + k_c * 8) * 2u, (unsigned)((16 + k_row) * d + k_c * 8) * 2u};
    unsigned offK8 = (unsigned)((t256 >> 3) * d) + (unsigned)(((t256 & 7) ^ (((t256 >> 3) >> 1) & 7)) << 4);
    unsigned offV = (unsigned)(v_row * a.Spad + v_c * 8) * 2u;
    // the sequence's LAST tile: rows past S / chunks past Spad are clamped to valid memory (its keys are masked afterwards).  A wave
    // issues tiles in increasing order, so the clamped offsets stay in place once selected.
    auto issue_one = [&](int kt, int i, int stage) {
        if (i >= 6 && X2) return;
#ifdef ATT_PP_NO_DMA      // (timing experiment: WRONG results)
        if (kt > 1) return;
#endif
        if (kt == last_tile) {       // (recomputed where it is needed - once or twice per wave - instead of held in four registers)
            const int kr = t256 >> 4, kcc = (t256 & 15) ^ (kr & 15);      // = k_row, k_c
            offK16[0] = (unsigned)(rowc(kr) * d + kcc * 8) * 2u;
            offK16[1] = (unsigned)(rowc(16 + kr) * d + kcc * 8) * 2u;
            offK8 = (unsigned)(rowc(t256 >> 3) * d) + (unsigned)(((t256 & 7) ^ (((t256 >> 3) >> 1) & 7)) << 4);
            const int vc = (t256 & 3) ^ (((t256 >> 2) >> 2) & 3);
            offV = (unsigned)((t256 >> 2) * a.Spad + (vc < chunks_last ? vc : chunks_last) * 8) * 2u;
        }
        half_t* st = lds + stage * ATT_STAGE + w4 * 512;
        const char* src;
        half_t* dst;
        if (MX && (i == 2 || i == 3)) {
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
    // fragment read offsets (halfs): K row l31, chunk (2ks+hi) ^ (l31&15); V row n*32+l31, chunk (2mf+hi) ^ ((row>>2)&3)
    const int kbase = l31 * 128, kx = l31 & 15;
    int vbase[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) vbase[mf] = l31 * 32 + (((2 * mf + hi) ^ ((l31 >> 2) & 3)) * 8);
    const int r8k = l31 * 128, sw8 = (l31 >> 1) & 7;

    const int ntiles_all = (S + KT - 1) / KT;
    const bool fd = a.ms != 0 || a.nsplit == 1;
    const int kt_begin = fd ? fast_div(split * ntiles_all, a.nsplit, a.ms) : (int)((long)split * ntiles_all / a.nsplit);
    const int kt_end = fd ? fast_div((split + 1) * ntiles_all, a.nsplit, a.ms) : (int)((long)(split + 1) * ntiles_all / a.nsplit);
    const int T = kt_end - kt_begin;       // key tiles of this split; tile index t = 0 .. T - 1 below is kt_begin + t, stage = t & 1

    // ---- state that crosses segment boundaries
    f32x16 sm;                   // scores of the tile whose softmax comes next
    f16x8 ph[2], pl[P1 ? 1 : 2];  // P of the tile whose P.V comes next
    f16x8 pre[2];                // fragments read ahead of the barrier for the coming compute segment: [0] K, [1] V (V_lo in F16X3: after the barrier)

    // fragment readers (stage is a compile-time constant: every address is a lane offset + immediate)
    auto rdK = [&](int stg, int ks) { return *reinterpret_cast<const f16x8*>(lds + stg * ATT_STAGE + kbase + (((2 * ks + hi) ^ kx) << 3)); };
    auto rdKl = [&](int stg, int ks) {
        return *reinterpret_cast<const f16x8*>(lds + stg * ATT_STAGE + ATT_KPLANE + kbase + (((2 * ks + hi) ^ kx) << 3));
    };
    auto rdK8 = [&](int stg, int img, int blk, int c) {
        const unsigned char* k8 = reinterpret_cast<const unsigned char*>(lds + stg * ATT_STAGE + ATT_KPLANE);
        return *reinterpret_cast<const i32x4*>(k8 + img * 4096 + r8k + (((blk * 4 + hi * 2 + c) ^ sw8) << 4));
    };
    auto rdV = [&](int stg, int n, int mf) { return *reinterpret_cast<const f16x8*>(lds + stg * ATT_STAGE + 2 * ATT_KPLANE + n * 1024 + vbase[mf]); };
    auto rdVl = [&](int stg, int n, int mf) {
        return *reinterpret_cast<const f16x8*>(lds + stg * ATT_STAGE + 2 * ATT_KPLANE + ATT_VPLANE + n * 1024 + vbase[mf]);
    };

    // ---- compute segment: P.V of the tile in stage SV (if PV) and QK^T of the tile in stage SK (if QK), interleaved
    //      F16MX / F16X2 with one plane of P: P.V step i = 4 mf + n (as in the two-wave kernel's PF path: per accumulator mf = 0, then 1)
    //      PRE: pre[] holds the first fragments (read ahead of the barrier): K0, V0 (those that exist)
    auto compute = [&](auto sv_c, auto sk_c, auto pv_c, auto qk_c, auto pre_c) {
        constexpr int SV = decltype(sv_c)::value, SK = decltype(sk_c)::value;
        constexpr bool PV = decltype(pv_c)::value, QK = decltype(qk_c)::value, PRE = decltype(pre_c)::value;
        if (QK) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sm[r] = 0.f;
        }
        if constexpr (P1) {
            // stream of 8 steps: [QK ks = i] [PV i]; fragments two steps (four reads) ahead
            f16x8 kf[NKS], vf[2 * NT];
            i32x4 k8f[2][2][2];
            if (QK) { kf[0] = PRE ? pre[0] : rdK(SK, 0); kf[1] = rdK(SK, 1); }
            if (PV) { vf[0] = PRE ? pre[1] : rdV(SV, 0, 0); vf[1] = rdV(SV, 1, 0); }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i + 2 < 8) {
                    if (QK) kf[i + 2] = rdK(SK, i + 2);
                    if (PV) vf[i + 2] = rdV(SV, (i + 2) & 3, (i + 2) >> 2);
                }
                if (QK && MX && i == 6) {
                    k8f[0][0][0] = rdK8(SK, 0, 0, 0); k8f[0][0][1] = rdK8(SK, 0, 0, 1);
                    k8f[0][1][0] = rdK8(SK, 1, 0, 0); k8f[0][1][1] = rdK8(SK, 1, 0, 1);
                }
                if (QK) {
                    sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[i], qh[i], sm, 0, 0, 0);
                    if (!MX) {
                        const f16x8 kl = rdKl(SK, i);
                        sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[i], ql[MX ? 0 : i], sm, 0, 0, 0);
                        sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[i], sm, 0, 0, 0);
                    }
                }
                if (PV) ot[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[i], ph[i >> 2], ot[i & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (QK && MX) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    if (blk == 0) {      // block 1's fragments arrive under block 0's two 64-cycle instructions
                        k8f[1][0][0] = rdK8(SK, 0, 1, 0); k8f[1][0][1] = rdK8(SK, 0, 1, 1);
                        k8f[1][1][0] = rdK8(SK, 1, 1, 0); k8f[1][1][1] = rdK8(SK, 1, 1, 1);
                    }
                    const i32x4 h0 = k8f[blk][0][0], h1 = k8f[blk][0][1], l0 = k8f[blk][1][0], l1 = k8f[blk][1][1];
                    const i32x8 kh8 = i32x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    const i32x8 kl8 = i32x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                    sm = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kh8, q8l[blk], sm, 1, 1, 0, 0, 0, 0);
                    sm = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kl8, q8h[blk], sm, 1, 1, 0, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            // F16X3: step s = ks for QK^T (kh.qh, kh.ql, kl.qh) and s = 2n + mf for P.V (vh.ph, vh.pl, vl.ph): six matrix instructions
            // per step, alternating between the score chain and an O accumulator; fragments one step ahead
            f16x8 kh_c, kl_c, vh_c, vl_c;
            if (QK) { kh_c = PRE ? pre[0] : rdK(SK, 0); kl_c = rdKl(SK, 0); }      // (K_lo may have arrived with this very barrier)
            if (PV) { vh_c = PRE ? pre[1] : rdV(SV, 0, 0); vl_c = rdVl(SV, 0, 0); }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                f16x8 kh_n = kh_c, kl_n = kl_c, vh_n = vh_c, vl_n = vl_c;
                if (s + 1 < 8) {
                    if (QK) { kh_n = rdK(SK, s + 1); kl_n = rdKl(SK, s + 1); }
                    if (PV) { vh_n = rdV(SV, (s + 1) >> 1, (s + 1) & 1); vl_n = rdVl(SV, (s + 1) >> 1, (s + 1) & 1); }
                }
                const int n = s >> 1, mf = s & 1;
                if (QK) sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh_c, qh[s], sm, 0, 0, 0);
                if (PV) ot[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh_c, ph[mf], ot[n], 0, 0, 0);
                if (QK) sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh_c, ql[MX ? 0 : s], sm, 0, 0, 0);
                if (PV) ot[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh_c, pl[P1 ? 0 : mf], ot[n], 0, 0, 0);
                if (QK) sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl_c, qh[s], sm, 0, 0, 0);
                if (PV) ot[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl_c, ph[mf], ot[n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                kh_c = kh_n; kl_c = kl_n; vh_c = vh_n; vl_c = vl_n;
            }
        }
    };
    // ---- the fragments a compute segment starts with, read before the barrier that opens it (their tiles are already visible)
    auto preread = [&](auto sv_c, auto sk_c, bool pv, bool qk) {
        constexpr int SV = decltype(sv_c)::value, SK = decltype(sk_c)::value;
        if (qk) pre[0] = rdK(SK, 0);       // (not K_lo: group 0 copies it in the segment that ends with this barrier)
        if (pv) pre[1] = rdV(SV, 0, 0);
    };
    // ---- vector segment, arithmetic part: online softmax of the scores in sm (tile kt), O rescale when the reference maximum moved,
    //      P packed for the coming P.V
    auto softmax = [&](int kt) {
        if (kt == ntiles_all - 1) {
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
        if constexpr (P1) {
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
            split8(pv + 8, ph[1], pl[P1 ? 0 : 1]);
        }
    };
#ifdef ATT_PP_TRACE      // (tools/attn_pp_check.hip -DATT_PP_TRACE: cycles per kind of segment, per wave, into a.Opart)
    unsigned long long tacc[4] = {0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define PP_STAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tacc[i] += n_ - tprev; tprev = n_; __builtin_amdgcn_sched_barrier(0); }
#else
#define PP_STAMP(i)
#endif
    // closes a segment.  `after_compute`: the barrier behind a compute segment (or the prologue).  A wave's copies are waited for at
    // the end of the VECTOR segment that issued them: deferring the wait to the end of the wave's next compute segment
    // (-DATT_PP_LATE_WAIT; the ring has the third stage for it) lengthens that compute segment from 905 to 1 667 cycles - the rest
    // of the copies' ~1 500-cycle flight is then paid there (tools/attn_pp_check.hip -DATT_PP_TRACE, docs/NOTEBOOK.md section 10)
    auto bar = [&](bool after_compute) {
#ifdef ATT_PP_LATE_WAIT
        const bool landed = after_compute;
#else
        const bool landed = !after_compute;
#endif
        if (landed) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's copies have landed ...
        PP_STAMP(2)
        __builtin_amdgcn_s_barrier();                          // ... and everybody else's; the previous segment's readers are done
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(3)
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using Yes = std::true_type;
    using No = std::false_type;

    // ---- prologue: K(0), K(1) and V(0), what the steady state would have copied in segments -4 .. -1
    auto copies_g1 = [&](int tv, int tk, int sv, int sk) {      // group 1: V^T(tv) and the first half of K_hi(tk)
        if (tv < T) {
            issue_one(kt_begin + tv, 4, sv);
            issue_one(kt_begin + tv, 5, sv);
            if (!X2) { issue_one(kt_begin + tv, 6, sv); issue_one(kt_begin + tv, 7, sv); }
        }
        if (X2 && tk < T) issue_one(kt_begin + tk, 0, sk);
    };
    auto copies_g0 = [&](int tk, int sk) {                      // group 0: the rest of K(tk)
        if (tk < T) {
            if (!X2) issue_one(kt_begin + tk, 0, sk);
            issue_one(kt_begin + tk, 1, sk);
            issue_one(kt_begin + tk, 2, sk);
            issue_one(kt_begin + tk, 3, sk);
        }
    };
    if (T > 0) {
        if (grp) { copies_g1(T, 0, 0, 0); copies_g1(0, 1, 0, 1); }
        else { copies_g0(0, 0); copies_g0(1, 1); }
    }
    q_finish();
    // The segment schedule of one wave.  ACT = false: a wave whose 32 queries all lie past the end of the sequence (S = 1200: 2.5 of
    // the 40 waves of a (sequence, head)) copies its share of every tile and meets every barrier, nothing else.  The first tile
    // (no P.V yet) is peeled off so that the loop body holds ONE form of each segment: with a run-time "first tile" test in it the
    // compiler merges the two forms by copying the 64 O accumulators at the top of every segment.
    using I2 = std::integral_constant<int, 2>;
    auto run = [&](auto act_c) {
        constexpr bool ACT = decltype(act_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the prologue's copies
        bar(true);
        if (grp == 0) {
            // group 0, tile t (stage s = t % 3):  C(t) = P.V(t-1) [stage s+2], QK^T(t) [s]
            //                                      V(t) = copies of K(t+2) [-> s+2]; softmax(t); pre-read V(t) [s], K(t+1) [s+1]
            auto vseg = [&](const int t, auto stg_c) {
                constexpr int STG = decltype(stg_c)::value;
                copies_g0(t + 2, (STG + 2) % 3);
                if (ACT) {
                    softmax(kt_begin + t);
                    preread(std::integral_constant<int, STG>{}, std::integral_constant<int, (STG + 1) % 3>{}, true, t + 1 < T);
                }
                PP_STAMP(1)
                bar(false);
            };
            auto tile = [&](const int t, auto stg_c) {
                constexpr int STG = decltype(stg_c)::value;
                if (ACT) compute(std::integral_constant<int, (STG + 2) % 3>{}, std::integral_constant<int, STG>{}, Yes{}, Yes{}, Yes{});
                PP_STAMP(0)
                bar(true);
                vseg(t, stg_c);
            };
            if (ACT) compute(I2{}, I0{}, No{}, Yes{}, No{});      // QK^T(0)
            bar(true);
            vseg(0, I0{});
            int t = 1;
            for (; t + 2 < T; t += 3) {
                tile(t, I1{});
                tile(t + 1, I2{});
                tile(t + 2, I0{});
            }
            // P.V of the last tile closes the wave: its stage is (T - 1) % 3
            if (t >= T) {
                if (ACT) compute(I0{}, I1{}, Yes{}, No{}, Yes{});
            } else {
                tile(t, I1{});
                if (t + 1 >= T) {
                    if (ACT) compute(I1{}, I2{}, Yes{}, No{}, Yes{});
                } else {
                    tile(t + 1, I2{});
                    if (ACT) compute(I2{}, I0{}, Yes{}, No{}, Yes{});
                }
            }
        } else {
            // group 1, tile t (stage s):  V'(t) = copies of V(t+1) [-> s+1] and K_hi(t+2) [-> s+2]; softmax(t-1); pre-read V(t-1) [s+2], K(t) [s]
            //                             C(t)  = P.V(t-1), QK^T(t)
            auto tile = [&](const int t, auto stg_c) {
                constexpr int STG = decltype(stg_c)::value;
                using SK = std::integral_constant<int, STG>;
                using SV = std::integral_constant<int, (STG + 2) % 3>;
                copies_g1(t + 1, t + 2, (STG + 1) % 3, (STG + 2) % 3);
                if (ACT) {
                    softmax(kt_begin + t - 1);
                    preread(SV{}, SK{}, true, true);
                }
                PP_STAMP(1)
                bar(false);
                if (ACT) compute(SV{}, SK{}, Yes{}, Yes{}, Yes{});
                PP_STAMP(0)
                bar(true);
            };
            copies_g1(1, 2, 1, 2);
            if (ACT) preread(I2{}, I0{}, false, true);
            bar(false);
            if (ACT) compute(I2{}, I0{}, No{}, Yes{}, Yes{});      // QK^T(0)
            bar(true);
            int t = 1;
            for (; t + 2 < T; t += 3) {
                tile(t, I1{});
                tile(t + 1, I2{});
                tile(t + 2, I0{});
            }
            if (t >= T) {
                if (ACT) { softmax(kt_begin + T - 1); compute(I0{}, I1{}, Yes{}, No{}, No{}); }
            } else {
                tile(t, I1{});
                if (t + 1 >= T) {
                    if (ACT) { softmax(kt_begin + T - 1); compute(I1{}, I2{}, Yes{}, No{}, No{}); }
                } else {
                    tile(t + 1, I2{});
                    if (ACT) { softmax(kt_begin + T - 1); compute(I2{}, I0{}, Yes{}, No{}, No{}); }
                }
            }
        }
    };
    if (T > 0) {
        if (wave_idle) run(No{});
        else run(Yes{});
    }
#ifdef ATT_PP_TRACE
    if (lane == 0 && !wave_idle && a.Opart) {
        unsigned long long* tr = reinterpret_cast<unsigned long long*>(a.Opart) + ((size_t)blockIdx.x * 8 + wid) * 4;
        for (int i = 0; i < 4; ++i) tr[i] = tacc[i];
    }
#endif

    if (a.nsplit > 1) {
        if (q < S) {
            const size_t Mtot = (size_t)(a.nseq ? a.nseq : gridDim.x / (nqt * a.nhead * a.nsplit)) * S;
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
                if (!X2) *reinterpret_cast<f16x4*>(a.Olo + ob) = vl;
            }
        }
        if (overflow) atomicOr(a.range_flag, 1);
    }
}

// does this head_dim-128 launch run on the ping-pong kernel?  "attn_pp": 0 = automatic, 1 = always, 2 = never (the two-wave kernel).
// Only the production operand sets exist in this form (one fp16 plane of P in F16X2 / F16MX: "attn_mx" = 0).
inline bool attn_pp_applies(const AttnHArgs& a, int nseq) {
    if (tune().attn_pp == 2 || tune().attn_mx != 0 || tune().attn_pf == 2) return false;
    if (tune().attn_pp == 1) return true;
    return false;
}

template <bool X2, bool MX>
inline void launch_attn_pp_one(const AttnHArgs& a, dim3 grid, int nqt, hipStream_t st) {
    static DevSeen seen;
    const auto kern = &attn_pp_kernel<X2, MX>;
    if (auto once_ = first_use_on_device(seen))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_PP_LDS);
    hipLaunchKernelGGL(kern, grid, dim3(512), ATT_PP_LDS, st, a, nqt);
}

// the ping-pong kernel for a head_dim-128 launch in the production operand sets (F16MX with bf8 K images, F16X2 and F16X3 with the
// default "attn_mx" = 0); the combine pass of a split-KV launch is the caller's
inline void launch_attn_pp(AttnHArgs a, int nseq, hipStream_t st) {
    const int nqt = (a.S + 255) / 256;
    const dim3 grid(nqt * a.nhead * nseq * a.nsplit);
    const unsigned long long x_max = std::max<unsigned long long>(grid.x, (unsigned long long)a.nsplit * ((a.S + 31) / 32));
    a.mq = fast_div_magic(nqt, x_max);
    a.ms = fast_div_magic(a.nsplit, x_max);
    a.mh = fast_div_magic(a.nhead, x_max);
    a.nseq = nseq;
    if (a.x2 && a.K8h) launch_attn_pp_one<true, true>(a, grid, nqt, st);
    else if (a.x2) launch_attn_pp_one<true, false>(a, grid, nqt, st);
    else launch_attn_pp_one<false, false>(a, grid, nqt, st);
}

}  // namespace jmid
