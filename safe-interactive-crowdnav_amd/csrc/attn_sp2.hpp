// EXPERIMENT (-DJMID_EXPERIMENTS, knob "attn_sp" = 2): head-dim-128 attention of F16MX / F16X2 (one fp16 plane of P, no key split) as a full
// in-wave software pipeline: iteration j issues the matrix instructions of the LOGITS of tile j + 1 and of P.V of tile j - 1 (20 instructions,
// 768 matrix cycles) with the SOFTMAX of tile j - ~90 vector instructions, cut into 17 atoms - placed in the gaps between them, so that a wave's
// vector work runs in the shadow of its own matrix work instead of its partner's.  Two score accumulators and two P planes alternate (the loop
// is unrolled six times: 3 ring stages x 2 parities, every register index and LDS offset a compile-time constant); K in a three-stage ring
// (copies two tiles ahead), V^T in a three-stage ring (one tile ahead; 72 KB per workgroup).  One barrier per tile.
// Per accumulator the instructions and their order are the 32-key kernel's, and O's rescale by tile j's alpha runs after P.V(j - 1) and before
// P.V(j): bit-identical.
#pragma once

namespace jmid {

template <bool MX>
__global__ __launch_bounds__(256, 2) void attn_sp2_kernel(AttnHArgs a, int nqt) {
    constexpr int HD = 128, KT = 32, NT = 4, NKS = 8;
    args_now_each(a, nqt);
    extern __shared__ __attribute__((aligned(16))) unsigned char att_lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(att_lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // XCD-aware order: the q-tiles of one (sequence, head) share K/V, keep them on one XCD's L2 (as attn_f16x3_dma_kernel)
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const int sh = fast_div(swz, nqt, a.mq), qt = swz - sh * nqt;
    const int seq = fast_div(sh, a.nhead, a.mh), h = sh - seq * a.nhead;
    const int S = a.S, d = a.d;
    const size_t tok0 = (size_t)seq * S;
    const int q = (qt * 4 + wid) * 32 + l31;
    const int qc = q < S ? q : S - 1;
    const bool wave_idle = (qt * 4 + wid) * 32 >= S;      // keeps copying its share of every tile and meets the barriers, computes nothing

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

    // DMA sources: wave-uniform base + a per-thread 32-bit offset; each stream's offsets are swapped IN PLACE when it reaches the sequence's last tile
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);
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
    auto k_to_last = [&]() {
        const int t = opaque(tid);
        offK16[0] = (unsigned)(rowc(t >> 4) * d + k_c * 8) * 2u;
        offK16[1] = (unsigned)(rowc(16 + (t >> 4)) * d + k_c * 8) * 2u;
        offK8 = (unsigned)(rowc(t >> 3) * d) + k8sw;
    };
    auto v_to_last = [&]() { offV = (unsigned)((opaque(tid) >> 2) * a.Spad + (v_c < chunks_last ? v_c : chunks_last) * 8) * 2u; };
    // K copy i of key tile kt (0, 1 = halves of K_hi; 2, 3 = bf8 images or halves of K_lo) / V^T copy i (halves of V^T_hi) into a ring stage
    auto issue_k = [&](int kt, int i, int stage) {
        if (i == 0 && kt == last_tile) k_to_last();      // (uniform)
        half_t* dst = lds + stage * SP_KST + wid_s * 512 + i * 2048;
        const char* src;
        if (MX && i >= 2) src = (i == 2 ? k8h_b : k8l_b) + (size_t)kt * (KT * d) + offK8;
        else src = ((i >> 1) ? kl_b : kh_b) + (size_t)kt * (KT * d) * 2 + offK16[i & 1];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto issue_v = [&](int kt, int i, int stage) {
        if (i == 0 && kt == last_tile) v_to_last();
        half_t* dst = lds + SP_VOFF + stage * 4096 + wid_s * 512 + i * 2048;
        const char* src = vth_b + (size_t)(64 * i) * a.Spad * 2 + (size_t)kt * 64 + offV;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
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
        va[mf] = 2u * (unsigned)(SP_VOFF + l31 * 32 + (((2 * mf + hi) ^ ((l31 >> 2) & 3)) * 8));
        asm volatile("" : "+v"(va[mf]));
    }

    const int n = last_tile + 1;      // key tiles (the launcher sends sequences of >= 3 tiles here)
    // prologue: K(0), V^T(0), K(1)
#pragma unroll
    for (int c = 0; c < 4; ++c) issue_k(0, c, 0);
#pragma unroll
    for (int c = 0; c < 2; ++c) issue_v(0, c, 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) issue_k(1, c, 1);
    q_finish();

    f32x16 sm[2];      // logits / P (fp32) of tile j (index j & 1) and of tile j + 1
    f16x8 ph[2][2];    // fp16 plane of P: tile j - 1 ([(j - 1) & 1]) and tile j
    // the logits of one tile out of ring stage KS into `acc`: the 32-key kernel's instruction sequence; slot(i) runs behind matrix instruction i
    // (i = 0 ... 11) - that is where the caller puts its P.V instructions, its copies and its softmax atoms
    auto logits = [&](auto ks_c, f32x16& acc, auto&& slot) {
        constexpr int KB = decltype(ks_c)::value * (SP_KST * 2);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        auto kread = [&](int ks) { return *reinterpret_cast<const f16x8*>(att_lds_raw + ka[ks] + KB); };
        constexpr int PFD = 2;
        f16x8 kf[NKS];
#pragma unroll
        for (int i = 0; i < PFD; ++i) kf[i] = kread(i);
        if (MX) {
            auto k8op = [&](int img, int blk) {
                const i32x4 c0 = *reinterpret_cast<const i32x4*>(att_lds_raw + k8a[2 * blk] + (KB + 8192 + img * 4096));
                const i32x4 c1 = *reinterpret_cast<const i32x4*>(att_lds_raw + k8a[2 * blk + 1] + (KB + 8192 + img * 4096));
                return i32x8{c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
            };
            i32x8 k8h_op, k8l_op;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks + PFD < NKS) kf[ks + PFD] = kread(ks + PFD);
                if (ks == 6) k8h_op = k8op(0, 0);
                if (ks == 7) k8l_op = k8op(1, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qh[ks], acc, 0, 0, 0);
                slot(ks);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k8h_op, q8l[0], acc, 1, 1, 0, 0, 0, 0);
            k8h_op = k8op(0, 1);
            slot(8);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k8l_op, q8h[0], acc, 1, 1, 0, 0, 0, 0);
            k8l_op = k8op(1, 1);
            slot(9);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k8h_op, q8l[1], acc, 1, 1, 0, 0, 0, 0);
            slot(10);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k8l_op, q8h[1], acc, 1, 1, 0, 0, 0, 0);
            slot(11);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            auto klread = [&](int ks) { return *reinterpret_cast<const f16x8*>(att_lds_raw + ka[ks] + (KB + 8192)); };
            f16x8 kl_c = klread(0);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                f16x8 kl_n = kl_c;
                if (ks + 1 < NKS) kl_n = klread(ks + 1);
                if (ks + PFD < NKS) kf[ks + PFD] = kread(ks + PFD);
                // twelve slots over eight steps, in order: step ks holds slots [12 ks / 8, 12 (ks + 1) / 8)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qh[ks], acc, 0, 0, 0);
                slot((12 * ks) / 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], ql[MX ? 0 : ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl_c, qh[ks], acc, 0, 0, 0);
                if ((12 * (ks + 1)) / 8 - (12 * ks) / 8 == 2) slot((12 * ks) / 8 + 1);
                __builtin_amdgcn_sched_barrier(0);
                kl_c = kl_n;
            }
        }
    };
    // the softmax of one tile (the 32-key kernel's arithmetic, instruction for instruction) as 17 atoms, run in order
    struct SoftmaxState { float tmax, m_new, alpha, psum; bool rescale; };
    auto atom = [&](int i, f32x16& s, f16x8 (&p)[2], SoftmaxState& st) {
        if (i == 0) {
            st.tmax = s[0];
#pragma unroll
            for (int r = 1; r < 8; ++r) st.tmax = fmaxf(st.tmax, s[r]);
        } else if (i == 1) {
#pragma unroll
            for (int r = 8; r < 16; ++r) st.tmax = fmaxf(st.tmax, s[r]);
        } else if (i == 2) {
            float x0, x1;
            half_swap(st.tmax, x0, x1);
            st.tmax = fmaxf(x0, x1);
        } else if (i == 3) {
            st.m_new = att_lazy_max(m_run, st.tmax);
            st.alpha = __builtin_amdgcn_exp2f(m_run - st.m_new);
            st.rescale = !__all(st.m_new == m_run);
            st.psum = 0.f;
        } else if (i < 12) {
#pragma unroll
            for (int r = 2 * (i - 4); r < 2 * (i - 4) + 2; ++r) {
                s[r] = __builtin_amdgcn_exp2f(s[r] - st.m_new);
                st.psum += s[r];
            }
        } else if (i == 12) {
            float x0, x1;
            half_swap(st.psum, x0, x1);
            st.psum = x0 + x1;
            l_run = fmaf(l_run, st.alpha, st.psum);
            m_run = st.m_new;
        } else {
            const int mf = (i - 13) >> 1, i0 = 2 * ((i - 13) & 1);
            u32x4 hq = __builtin_bit_cast(u32x4, p[mf]);
            hq[i0] = pk_f16_rne(s[8 * mf + 2 * i0], s[8 * mf + 2 * i0 + 1]);
            hq[i0 + 1] = pk_f16_rne(s[8 * mf + 2 * i0 + 2], s[8 * mf + 2 * i0 + 3]);
            p[mf] = __builtin_bit_cast(f16x8, hq);
        }
    };
    auto rescale_o = [&](const float alpha) {
#pragma unroll
        for (int nn = 0; nn < NT; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[nn][r] *= alpha;
    };
    constexpr int NA = 17;
    // iteration j: logits(j + 1) + P.V(j - 1) with softmax(j) in their shadow.  J3 = j % 3, PAR = j & 1 (compile time); FIRST: j = 0 (no P.V)
    auto iter = [&](const int j_in, auto j3_c, auto par_c, auto first_c) {
        constexpr int J3 = decltype(j3_c)::value, PAR = decltype(par_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int KS = (J3 + 1) % 3, VS = (J3 + 2) % 3, KN = (J3 + 2) % 3, VN = (J3 + 1) % 3;
        const int j = __builtin_amdgcn_readfirstlane(j_in);
        const bool kmore = j + 2 < n;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // K(j + 1) and V^T(j) have landed (this wave's share)
        __builtin_amdgcn_s_barrier();                      // ... everybody's; everybody is through iteration j - 1
        __builtin_amdgcn_sched_barrier(0);
        auto copy = [&](int c) {      // K(j + 2) into the stage K(j - 1) left, V^T(j + 1) into the stage V^T(j - 2) left
            if (c < 4) {
                if (kmore) issue_k(j + 2, c, KN);
            } else {
                issue_v(j + 1, c - 4, VN);
            }
        };
        if (wave_idle) {
#pragma unroll
            for (int c = 0; c < 6; ++c) copy(c);
            return;
        }
        SoftmaxState st;
        constexpr int NSLOT = FIRST ? 12 : 20;
        int done = 0;
        auto atoms_upto = [&](int slot_idx) {      // the atoms that belong behind matrix instruction `slot_idx` of NSLOT
#pragma unroll
            for (int i = (slot_idx * NA) / NSLOT; i < ((slot_idx + 1) * NA) / NSLOT; ++i) atom(i, sm[PAR], ph[PAR], st);
            (void)done;
        };
        constexpr int VB = VS * 8192;
        auto vread = [&](int i) { return *reinterpret_cast<const f16x8*>(att_lds_raw + va[i >> 2] + (VB + (i & 3) * 2048)); };
        f16x8 vf[2 * NT];
        if (!FIRST) {
            vf[0] = vread(0);
            vf[1] = vread(1);
        }
        logits(std::integral_constant<int, KS>{}, sm[PAR ^ 1], [&](int i) {
            if (FIRST) {
                if (i < 6) copy(i);
                atoms_upto(i);
            } else if (i < 8) {
                // behind logits instruction i: P.V instruction i of tile j - 1, then the atoms of both slots
                if (i + 2 < 2 * NT) vf[i + 2] = vread(i + 2);
                if (i < 6) copy(i);
                atoms_upto(2 * i);
                ot[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[i], ph[PAR ^ 1][i >> 2], ot[i & 3], 0, 0, 0);
                atoms_upto(2 * i + 1);
            } else {
                atoms_upto(8 + i);
            }
        });
        if (st.rescale) rescale_o(st.alpha);      // after every P.V(j - 1) instruction, before any of P.V(j)
    };
    // the last tile: its softmax, then P.V(n - 2) and P.V(n - 1) (run-time ring stages; once per wave)
    auto finish = [&](auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (wave_idle) return;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if ((n - 1) * KT + frag_row(r, hi) >= S) sm[PAR][r] = -INFINITY;      // only the last tile can hold keys past S
        SoftmaxState st;
#pragma unroll
        for (int i = 0; i < NA; ++i) atom(i, sm[PAR], ph[PAR], st);
        const int vb1 = ((n - 2) % 3) * 8192, vb2 = ((n - 1) % 3) * 8192;
#pragma unroll
        for (int i = 0; i < 2 * NT; ++i) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(att_lds_raw + va[i >> 2] + vb1 + (i & 3) * 2048);
            ot[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v, ph[PAR ^ 1][i >> 2], ot[i & 3], 0, 0, 0);
        }
        if (st.rescale) rescale_o(st.alpha);
#pragma unroll
        for (int i = 0; i < 2 * NT; ++i) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(att_lds_raw + va[i >> 2] + vb2 + (i & 3) * 2048);
            ot[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v, ph[PAR][i >> 2], ot[i & 3], 0, 0, 0);
        }
    };
    {
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        // logits of tile 0 (nothing to overlap them with)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (!wave_idle) logits(I0{}, sm[0], [&](int) {});
        iter(0, I0{}, I0{}, std::true_type{});
        int j = 1;
        for (; j + 5 <= n - 2; j += 6) {
            iter(j, I1{}, I1{}, std::false_type{});
            iter(j + 1, I2{}, I0{}, std::false_type{});
            iter(j + 2, I0{}, I1{}, std::false_type{});
            iter(j + 3, I1{}, I0{}, std::false_type{});
            iter(j + 4, I2{}, I1{}, std::false_type{});
            iter(j + 5, I0{}, I0{}, std::false_type{});
        }
        if (j <= n - 2) iter(j, I1{}, I1{}, std::false_type{});
        if (j + 1 <= n - 2) iter(j + 1, I2{}, I0{}, std::false_type{});
        if (j + 2 <= n - 2) iter(j + 2, I0{}, I1{}, std::false_type{});
        if (j + 3 <= n - 2) iter(j + 3, I1{}, I0{}, std::false_type{});
        if (j + 4 <= n - 2) iter(j + 4, I2{}, I1{}, std::false_type{});
        if ((n - 1) & 1) finish(I1{});
        else finish(I0{});
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

inline bool attn_sp2_applies(const AttnHArgs& a) {
    if (!a.x2 || a.nsplit != 1 || tune().attn_mx == 1 || tune().attn_pf == 2) return false;
    if (a.K8h && !a.Q8l) return false;
    return tune().attn_sp == 2 && a.S > 96;      // (at least three key tiles)
}

inline void launch_attn_sp2(const AttnHArgs& a, int nseq, int nqt, hipStream_t st) {
    const dim3 grid(nqt * a.nhead * nseq);
    const size_t ldsb = tune().attn_one_wg ? 160 * 1024 : ATT_SP_LDS;
    if (a.K8h) {
        static DevSeen seen;
        const auto kern = &attn_sp2_kernel<true>;
        if (auto once_ = first_use_on_device(seen))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kern, grid, dim3(256), ldsb, st, a, nqt);
    } else {
        static DevSeen seen;
        const auto kern = &attn_sp2_kernel<false>;
        if (auto once_ = first_use_on_device(seen))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kern, grid, dim3(256), ldsb, st, a, nqt);
    }
}

}  // namespace jmid
