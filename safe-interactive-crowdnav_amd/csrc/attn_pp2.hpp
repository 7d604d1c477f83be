// EXPERIMENT (-DJMID_EXPERIMENTS, knob "attn_pp" = 3): the 8-wave ping-pong once more, built on what this round measured - a wave's softmax
// overlaps matrix work only when it is ANOTHER wave's (attn_sp2.hpp), and the ping-pong of attn_pp.hpp lost its gain to the wait for the copies
// inside its vector segments and to the barrier imbalance that wait caused.  Here:
//   * workgroup = 8 waves = 256 queries; waves w and w + 4 share a SIMD (group 0 / group 1); every wave alternates a COMPUTE segment C(t) - the
//     logits of tile t with P.V of tile t - 1 between them, 20 matrix instructions (attn_sp.hpp's matrix phase) - and a VECTOR segment V(t) -
//     the softmax of tile t; group 1 runs one segment behind group 0, an s_barrier between segments: a SIMD always has one wave in C and one in V;
//   * only group 0 copies (its 256 threads run the 32-key kernel's six copy rounds per tile, one behind each of the first matrix instructions of
//     its compute segment), for the tile AFTER next: a copy has three segments (~3 000 cycles) to land before `s_waitcnt vmcnt(6)` at the end of
//     a vector segment asks for it, and group 1 never waits at all;
//   * K and V^T in four-stage rings (96 KB; one workgroup per CU), stage = tile % 4 a compile-time constant (loop unrolled four times).
// Per accumulator the instructions and their order are the 32-key kernel's: bit-identical (tools/attn_k64_check.hip with PP2=1: S = 33 ... 1 217,
// both operand sets).  MEASURED (profiles/r05_attn_pp2_check.log, 51 sequences of 1 200): 0.2692 ms per launch against 0.2591 (F16X2 0.3242 against
// 0.3087) - 4 % slower, where attn_pp.hpp was 4 % slower too.  With the copies out of the way what remains is the structure itself: every segment
// lasts as long as the slowest of eight waves, twice per tile, and the two free-running workgroups of the shipped kernel lose less to chance than
// this one loses to waiting for its slowest member.  (Cycle stamps perturb this kernel beyond use: every variant of them spills.)
#pragma once

namespace jmid {

constexpr int PP2_VOFF = 4 * SP_KST;               // V^T stages (4096 halfs each) behind the four K stages, at byte 65 536
constexpr size_t ATT_PP2_LDS = size_t(4 * SP_KST + 4 * 4096) * sizeof(half_t);      // 96 KB

template <bool MX>
__global__ __launch_bounds__(512, 1) void attn_pp2_kernel(AttnHArgs a, int nqt) {
    constexpr int HD = 128, KT = 32, NT = 4, NKS = 8;
    args_now_each(a, nqt);
    extern __shared__ __attribute__((aligned(16))) unsigned char att_lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(att_lds_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wid >> 2;      // waves w and w + 4 share a SIMD: group 0 / group 1
    const int l31 = lane & 31, hi = lane >> 5;
    // XCD-aware order: the q-tiles of one (sequence, head) share K/V, keep them on one XCD's L2 (as attn_f16x3_dma_kernel)
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const int sh = fast_div(swz, nqt, a.mq), qt = swz - sh * nqt;
    const int seq = fast_div(sh, a.nhead, a.mh), h = sh - seq * a.nhead;
    const int S = a.S, d = a.d;
    const size_t tok0 = (size_t)seq * S;
    const int q = (qt * 8 + wid) * 32 + l31;
    const int qc = q < S ? q : S - 1;
    const bool wave_idle = (qt * 8 + wid) * 32 >= S;      // keeps copying its share of every tile and meets the barriers, computes nothing

    // Q operands (as attn_f16x3_dma_kernel: raw loads first, conversions after the first copies have been issued)
    f16x8 qh[NKS], ql[MX ? 1 : NKS];
    i32x8 q8h[2], q8l[2];
    i32x4 q8raw[2][4], q8lraw[2][2];
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
            for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    q8raw[blk][c] = __builtin_bit_cast(i32x4, *reinterpret_cast<const f16x8*>(a.Qhi + o8 + 64 * blk + 8 * c));
                q8lraw[blk][0] = *reinterpret_cast<const i32x4*>(a.Q8l + o8 + 64 * blk);
                q8lraw[blk][1] = *reinterpret_cast<const i32x4*>(a.Q8l + o8 + 64 * blk + 16);
            }
        }
    }
    auto q_finish = [&]() {
        if (MX) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    q8h[blk][2 * c] = bf8_of_f16x4(q8raw[blk][c][0], q8raw[blk][c][1]);
                    q8h[blk][2 * c + 1] = bf8_of_f16x4(q8raw[blk][c][2], q8raw[blk][c][3]);
                }
                const i32x4 l0 = q8lraw[blk][0], l1 = q8lraw[blk][1];
                q8l[blk] = i32x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
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
    float m_run = -INFINITY, l_run = 0.f;   // running max in log2 units (Q is pre-scaled by log2(e)/sqrt(hd))

    // DMA sources: wave-uniform base + a per-thread 32-bit offset; the offsets are swapped IN PLACE when the copies reach the sequence's last tile
    const int wid_s = wid & 3;      // (only group 0 - threads 0 ... 255 - copies: the 32-key kernel's six copy rounds per tile)
    const char* const kh_b = reinterpret_cast<const char*>(a.Khi + tok0 * d + h * HD);
    const char* const kl_b = reinterpret_cast<const char*>(a.Klo + tok0 * d + h * HD);
    const char* const k8h_b = reinterpret_cast<const char*>(a.K8h) + (tok0 * d + h * HD);
    const char* const k8l_b = reinterpret_cast<const char*>(a.K8l) + (tok0 * d + h * HD);
    const size_t vt0 = ((size_t)seq * a.nhead + h) * HD * a.Spad;
    const char* const vth_b = reinterpret_cast<const char*>(a.Vthi + vt0);
    const int k_row = tid >> 4, k_c = (tid & 15) ^ (k_row & 15);
    const int v_row = tid >> 2, v_c = (tid & 3) ^ ((v_row >> 2) & 3);
    const int last_tile = (S + KT - 1) / KT - 1;
    const int rows_last = S - last_tile * KT - 1;
    const int chunks_last = a.Spad / 8 - 1 - last_tile * 4;
    auto rowc = [&](int r) { return r < rows_last ? r : rows_last; };
    const unsigned k8sw = (unsigned)(((tid & 7) ^ (((tid >> 3) >> 1) & 7)) << 4);
    unsigned offK16[2] = {(unsigned)(k_row * d + k_c * 8) * 2u, (unsigned)((16 + k_row) * d + k_c * 8) * 2u};
    unsigned offK8 = (unsigned)((tid >> 3) * d) + k8sw;
    unsigned offV = (unsigned)(v_row * a.Spad + v_c * 8) * 2u;
    auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };
    auto to_last = [&]() {
        const int t = opaque(tid);
        offK16[0] = (unsigned)(rowc(t >> 4) * d + k_c * 8) * 2u;
        offK16[1] = (unsigned)(rowc(16 + (t >> 4)) * d + k_c * 8) * 2u;
        offK8 = (unsigned)(rowc(t >> 3) * d) + k8sw;
        offV = (unsigned)((t >> 2) * a.Spad + (v_c < chunks_last ? v_c : chunks_last) * 8) * 2u;
    };
    // copy i of key tile kt into ring stage `stage`: 0, 1 = halves of K_hi; 2, 3 = bf8 images (or halves of K_lo); 4, 5 = halves of V^T_hi
    auto issue_one = [&](int kt, int i, int stage) {
        const char* src;
        half_t* dst;
        if (i < 4) {
            dst = lds + stage * SP_KST + wid_s * 512 + i * 2048;
            if (MX && i >= 2) src = (i == 2 ? k8h_b : k8l_b) + (size_t)kt * (KT * d) + offK8;
            else src = ((i >> 1) ? kl_b : kh_b) + (size_t)kt * (KT * d) * 2 + offK16[i & 1];
        } else {
            dst = lds + PP2_VOFF + stage * 4096 + wid_s * 512 + (i - 4) * 2048;
            src = vth_b + (size_t)(64 * (i - 4)) * a.Spad * 2 + (size_t)kt * 64 + offV;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    // fragment read addresses: one opaque register per distinct per-lane offset, every read register + immediate (attn_k64.hpp)
    unsigned ka[NKS], k8a[4], va[2];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        ka[ks] = 2u * (unsigned)(l31 * 128 + (((2 * ks + hi) ^ (l31 & 15)) << 3));
        asm volatile("" : "+v"(ka[ks]));
    }
#pragma unroll
    for (int i = 0; i < (MX ? 4 : 0); ++i) {      // i = 2 blk + c
        k8a[i] = (unsigned)(l31 * 128 + ((((i >> 1) * 4 + hi * 2 + (i & 1)) ^ ((l31 >> 1) & 7)) << 4));
        asm volatile("" : "+v"(k8a[i]));
    }
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        va[mf] = 2u * (unsigned)(PP2_VOFF + l31 * 32 + (((2 * mf + hi) ^ ((l31 >> 2) & 3)) * 8));
        asm volatile("" : "+v"(va[mf]));
    }

    const int n = last_tile + 1;
    if (grp == 0) {      // tiles 0 and 1
        if (last_tile == 0) to_last();
#pragma unroll
        for (int c = 0; c < 6; ++c) issue_one(0, c, 0);
        if (n > 1) {
            if (last_tile == 1) to_last();
#pragma unroll
            for (int c = 0; c < 6; ++c) issue_one(1, c, 1);
        }
    }
    q_finish();

    f16x8 ph[2];      // P of the previous tile (fp16), carried into the next compute segment
    f32x16 sm;        // logits of the current tile, carried from its compute segment into its vector segment
    // compute segment of tile kt: logits(kt) with P.V(kt - 1) between them; group 0 also issues the copies of tile kt + 2
    auto compute = [&](const int kt_in, auto stg_c, auto first_c) {
        constexpr int STG = decltype(stg_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int PSTG = (STG + 3) % 4, CSTG = (STG + 2) % 4;
        const int kt = __builtin_amdgcn_readfirstlane(kt_in);
        const bool copies = grp == 0 && kt + 2 < n;
        if (copies && kt + 2 == last_tile) to_last();
        if (wave_idle) {
            if (copies) {
#pragma unroll
                for (int c = 0; c < 6; ++c) issue_one(kt + 2, c, CSTG);
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sm[r] = 0.f;
        constexpr int KB = STG * (SP_KST * 2), VB = PSTG * 8192;
        auto kread = [&](int ks) { return *reinterpret_cast<const f16x8*>(att_lds_raw + ka[ks] + KB); };
        auto vread = [&](int i) { return *reinterpret_cast<const f16x8*>(att_lds_raw + va[i >> 2] + (VB + (i & 3) * 2048)); };
        constexpr int PFD = 2;
        f16x8 kf[NKS], vf[2 * NT];
#pragma unroll
        for (int i = 0; i < PFD; ++i) {
            kf[i] = kread(i);
            if (!FIRST) vf[i] = vread(i);
        }
        if (MX) {
            auto k8op = [&](int img, int blk) {
                const i32x4 c0 = *reinterpret_cast<const i32x4*>(att_lds_raw + k8a[2 * blk] + (KB + 8192 + img * 4096));
                const i32x4 c1 = *reinterpret_cast<const i32x4*>(att_lds_raw + k8a[2 * blk + 1] + (KB + 8192 + img * 4096));
                return i32x8{c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
            };
            i32x8 k8h_op, k8l_op;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks + PFD < NKS) {
                    kf[ks + PFD] = kread(ks + PFD);
                    if (!FIRST) vf[ks + PFD] = vread(ks + PFD);
                }
                if (ks == 6) k8h_op = k8op(0, 0);
                if (ks == 7) k8l_op = k8op(1, 0);
                sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qh[ks], sm, 0, 0, 0);
                if (!FIRST) ot[ks & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[ks], ph[ks >> 2], ot[ks & 3], 0, 0, 0);
                if (copies && ks < 6) issue_one(kt + 2, ks, CSTG);
                __builtin_amdgcn_sched_barrier(0);
            }
            sm = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k8h_op, q8l[0], sm, 1, 1, 0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            k8h_op = k8op(0, 1);
            sm = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k8l_op, q8h[0], sm, 1, 1, 0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            k8l_op = k8op(1, 1);
            sm = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k8h_op, q8l[1], sm, 1, 1, 0, 0, 0, 0);
            sm = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k8l_op, q8h[1], sm, 1, 1, 0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            auto klread = [&](int ks) { return *reinterpret_cast<const f16x8*>(att_lds_raw + ka[ks] + (KB + 8192)); };
            f16x8 kl_c = klread(0);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                f16x8 kl_n = kl_c;
                if (ks + 1 < NKS) kl_n = klread(ks + 1);
                if (ks + PFD < NKS) {
                    kf[ks + PFD] = kread(ks + PFD);
                    if (!FIRST) vf[ks + PFD] = vread(ks + PFD);
                }
                sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qh[ks], sm, 0, 0, 0);
                if (!FIRST) ot[ks & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[ks], ph[ks >> 2], ot[ks & 3], 0, 0, 0);
                sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], ql[MX ? 0 : ks], sm, 0, 0, 0);
                sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl_c, qh[ks], sm, 0, 0, 0);
                if (copies && ks < 6) issue_one(kt + 2, ks, CSTG);
                __builtin_amdgcn_sched_barrier(0);
                kl_c = kl_n;
            }
        }
    };
    // vector segment of tile kt: its softmax (the 32-key kernel's); O holds every P.V up to tile kt - 1
    auto vector = [&](const int kt) {
        if (wave_idle) return;
        if (kt == n - 1) {       // only the sequence's last tile can hold keys past S
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
            for (int nn = 0; nn < NT; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[nn][r] *= alpha;
        }
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            u32x4 hq;
#pragma unroll
            for (int i = 0; i < 4; ++i) hq[i] = pk_f16_rne(sm[8 * mf + 2 * i], sm[8 * mf + 2 * i + 1]);
            ph[mf] = __builtin_bit_cast(f16x8, hq);
        }
    };
    auto bar = [&]() {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // one tile of a wave's timeline: C(kt) | V(kt) |   (| = s_barrier); group 0 makes sure, before the barrier that ends V(kt), that its copies
    // of tile kt + 1 have landed (those of tile kt + 2, issued in C(kt), may still fly)
    auto tile = [&](const int kt, auto stg_c, auto first_c) {
        compute(kt, stg_c, first_c);
        bar();
        vector(kt);
        if (grp == 0) {
            if (kt + 2 < n) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        bar();
    };
    {
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>;
        using S3 = std::integral_constant<int, 3>;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // Q; group 0: tiles 0 and 1
        bar();
        if (grp == 1) bar();                                      // group 1 runs one segment behind
        tile(0, S0{}, std::true_type{});
        int kt = 1;
        for (; kt + 3 < n; kt += 4) {
            tile(kt, S1{}, std::false_type{});
            tile(kt + 1, S2{}, std::false_type{});
            tile(kt + 2, S3{}, std::false_type{});
            tile(kt + 3, S0{}, std::false_type{});
        }
        if (kt < n) tile(kt, S1{}, std::false_type{});
        if (kt + 1 < n) tile(kt + 1, S2{}, std::false_type{});
        if (kt + 2 < n) tile(kt + 2, S3{}, std::false_type{});
        // the last tile's P.V (its V^T stage: (n - 1) % 4; nobody overwrites it any more)
        if (!wave_idle) {
            const int vb = ((n - 1) % 4) * 8192;
#pragma unroll
            for (int i = 0; i < 2 * NT; ++i) {
                const f16x8 v = *reinterpret_cast<const f16x8*>(att_lds_raw + va[i >> 2] + vb + (i & 3) * 2048);
                ot[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v, ph[i >> 2], ot[i & 3], 0, 0, 0);
            }
        }
        if (grp == 0) bar();                                      // (the barrier count of group 1)
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
                f16x4 vh;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = ot[n][4 * r4 + e] * inv;
                    half_t hh, ll;
                    split_f32(v, hh, ll);
                    overflow |= !(fabsf(v) <= kHalfMax);
                    vh[e] = hh;
                }
                *reinterpret_cast<f16x4*>(a.Ohi + blk_index(orow, h * HD + c0, d)) = vh;      // (F16X2 / F16MX: out_proj reads O_hi only)
            }
        }
        if (overflow) atomicOr(a.range_flag, 1);
    }
}

inline bool attn_pp2_applies(const AttnHArgs& a) {
    if (!a.x2 || a.nsplit != 1 || tune().attn_mx == 1 || tune().attn_pf == 2) return false;
    if (a.K8h && !a.Q8l) return false;
    return tune().attn_pp == 3;
}

inline void launch_attn_pp2(AttnHArgs a, int nseq, hipStream_t st) {
    const int nqt8 = (a.S + 255) / 256;
    const dim3 grid(nqt8 * a.nhead * nseq);
    a.mq = fast_div_magic(nqt8, grid.x);
    a.mh = fast_div_magic(a.nhead, grid.x);
    if (a.K8h) {
        static DevSeen seen;
        const auto kern = &attn_pp2_kernel<true>;
        if (auto once_ = first_use_on_device(seen))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kern, grid, dim3(512), ATT_PP2_LDS, st, a, nqt8);
    } else {
        static DevSeen seen;
        const auto kern = &attn_pp2_kernel<false>;
        if (auto once_ = first_use_on_device(seen))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kern, grid, dim3(512), ATT_PP2_LDS, st, a, nqt8);
    }
}

}  // namespace jmid
