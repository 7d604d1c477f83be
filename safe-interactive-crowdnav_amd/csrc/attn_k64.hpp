// EXPERIMENT (-DJMID_EXPERIMENTS, knob "attn_k64" = 1): head-dim-128 attention with 64-key tiles (F16MX and F16X2, launches without a key split).
// The idea: the per-tile costs of the 32-key kernel that do not scale with the keys - the wait for the copies (~280 of 2 670 cycles per 32 keys
// in its cycle stamps), a barrier - once per 64 keys, with every copy issued a whole tile ahead:
//   * a tile is two 32-key rounds of attn_f16x3_dma_kernel<X2, MX, P1, PF> - logits, softmax, logits, softmax - whose two P.V products are
//     deferred behind ONE wait for V^T and ONE barrier; K(T + 1) has a whole tile to land;
//   * when the SECOND round moves the reference maximum (rare: attn_f16x3.hpp::att_lazy_max), its rescale of O runs behind the first round's
//     P.V - the 32-key kernel's order.
// So the kernel is bit-identical to the 32-key kernel for every input (tools/attn_k64_check.hip: S = 40 ... 1217, both operand sets, logits
// scaled until every tile rescales; tests/test_gpu_parity.py::test_attention_on_64_key_tiles_equals_the_32_key_kernel).
// MEASURED (51 sequences of 1 200, profiles/r05_attn_k64_check.log): F16MX 0.2628 ms per launch against 0.2523, F16X2 0.3009 against 0.2994 -
// no gain.  The same log holds the measurement that explains it (knob "attn_one_wg": ONE workgroup per CU, a wave alone on its SIMD): the
// 32-key kernel needs 0.3073 ms that way, the 64-key kernel 0.3077.  A wave alone runs a 32-key tile in ~1 630 cycles (768 of matrix
// instructions + ~600 of softmax + ~250 of waits: strictly serial), the second wave per SIMD adds 22 % - what it fills are exactly the
// waits and barriers this kernel removes, so removing them buys nothing; what limits the pair is that both waves want the same unit at the
// same time 40 % of the time.  What the way here taught about hipcc (docs/NOTEBOOK.md section 10): a loop counter kept in a VECTOR register
// turns every copy's address into 64-bit vector arithmetic (readfirstlane it); a run-time ring stage costs three vector instructions per
// fragment read (opaque per-lane address registers + immediates instead); an `if` that reorders two phases makes hipcc copy all 64
// accumulators twice per tile (keep the phases straight-line, make only the rescale conditional); the key mask in the loop body pins 16
// registers (compile it into the last tile only).
#pragma once

namespace jmid {

constexpr int K64_SUB = 8192;                      // halfs per 32-key K sub-tile: K_hi 8 KB + (bf8 images 4 + 4 KB | K_lo 8 KB)
constexpr int K64_KSTAGE = 2 * K64_SUB;            // halfs per K stage (64 keys)
constexpr int K64_VOFF = 2 * K64_KSTAGE;           // V^T (two sub-tiles of 4096 halfs) behind the two K stages, at byte 65 536: the K reads are a per-lane
                                                   // address register + an immediate below 64 KB, the V^T reads' registers hold the 64 KB themselves
constexpr size_t ATT_K64_LDS = size_t(K64_VOFF + 2 * 4096) * sizeof(half_t);      // 80 KB

template <bool MX>
__global__ __launch_bounds__(256, 2) void attn_k64_kernel(AttnHArgs a, int nqt) {
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

    // DMA sources: wave-uniform base (plane pointer + sub-tile stride) + a per-thread 32-bit offset, two offset sets (regular; the
    // sequence's last 32-key sub-tile, rows past S / chunks past Spad clamped to valid memory - those keys are masked)
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);
    const char* const kh_b = reinterpret_cast<const char*>(a.Khi + tok0 * d + h * HD);
    const char* const kl_b = reinterpret_cast<const char*>(a.Klo + tok0 * d + h * HD);
    const char* const k8h_b = reinterpret_cast<const char*>(a.K8h) + (tok0 * d + h * HD);
    const char* const k8l_b = reinterpret_cast<const char*>(a.K8l) + (tok0 * d + h * HD);
    const size_t vt0 = ((size_t)seq * a.nhead + h) * HD * a.Spad;
    const char* const vth_b = reinterpret_cast<const char*>(a.Vthi + vt0);
    const int k_row = tid >> 4, k_c = (tid & 15) ^ (k_row & 15);
    const int v_row = tid >> 2, v_c = (tid & 3) ^ ((v_row >> 2) & 3);
    const int last32 = (S + KT - 1) / KT - 1;                     // index of the last 32-key sub-tile
    const int rows_last = S - last32 * KT - 1;                    // its highest valid row
    const int chunks_last = a.Spad / 8 - 1 - last32 * 4;          // highest valid 16-byte chunk of its V^T rows
    auto rowc = [&](int r) { return r < rows_last ? r : rows_last; };
    const unsigned k8sw = (unsigned)(((tid & 7) ^ (((tid >> 3) >> 1) & 7)) << 4);
    // two offset sets as in the 32-key kernel, swapped IN PLACE by a uniform branch when a stream reaches the sequence's last sub-tile (the K
    // stream and the V^T stream each run through the sub-tiles in order, so each swaps once)
    unsigned offK16[2] = {(unsigned)(k_row * d + k_c * 8) * 2u, (unsigned)((16 + k_row) * d + k_c * 8) * 2u};
    unsigned offK8 = (unsigned)((tid >> 3) * d) + k8sw;
    unsigned offV = (unsigned)(v_row * a.Spad + v_c * 8) * 2u;
    auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };      // (keeps the last set out of the loop's registers)
    auto k_to_last = [&]() {
        const int t = opaque(tid);
        offK16[0] = (unsigned)(rowc(t >> 4) * d + k_c * 8) * 2u;
        offK16[1] = (unsigned)(rowc(16 + (t >> 4)) * d + k_c * 8) * 2u;
        offK8 = (unsigned)(rowc(t >> 3) * d) + k8sw;
    };
    auto v_to_last = [&]() { offV = (unsigned)((opaque(tid) >> 2) * a.Spad + (v_c < chunks_last ? v_c : chunks_last) * 8) * 2u; };
    // K copy i (0, 1: the halves of K_hi; 2, 3: the bf8 images of K_hi / K_lo, or the halves of K_lo) of sub-tile u of 64-key tile T
    auto issue_k = [&](int T, int u, int i, int stage) {
        int j = 2 * T + u;
        if (i == 0 && j == last32) k_to_last();        // (uniform)
        j = j > last32 ? last32 : j;                   // a sub-tile past the sequence copies the last one again: valid memory, every key masked
        half_t* dst = lds + stage * K64_KSTAGE + u * K64_SUB + wid_s * 512 + i * 2048;
        const char* src;
        if (MX && i >= 2) src = (i == 2 ? k8h_b : k8l_b) + (size_t)j * (KT * d) + offK8;
        else src = ((i >> 1) ? kl_b : kh_b) + (size_t)j * (KT * d) * 2 + offK16[i & 1];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    // V^T copy i (rows 0-63 / 64-127 of the hi plane) of sub-tile u of tile T
    auto issue_v = [&](int T, int u, int i) {
        int j = 2 * T + u;
        if (i == 0 && j == last32) v_to_last();
        j = j > last32 ? last32 : j;
        half_t* dst = lds + K64_VOFF + u * 4096 + wid_s * 512 + i * 2048;
        const char* src = vth_b + (size_t)(64 * i) * a.Spad * 2 + (size_t)j * 64 + offV;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    // fragment read addresses (bytes from the start of LDS; the swizzles of attn_f16x3_dma_kernel): one register per distinct per-lane offset,
    // made opaque - every read is then register + immediate (stage, sub-tile, plane), and hipcc neither re-derives an address per read (three
    // vector instructions each) nor keeps one register per (stage, sub-tile) combination
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
        va[mf] = 2u * (unsigned)(K64_VOFF + l31 * 32 + (((2 * mf + hi) ^ ((l31 >> 2) & 3)) * 8));
        asm volatile("" : "+v"(va[mf]));
    }

    const int nt64 = (last32 + 2) / 2;       // 64-key tiles
#pragma unroll
    for (int c = 0; c < 8; ++c) issue_k(0, c >> 2, c & 3, 0);
    q_finish();

#ifdef ATT_K64_TRACE      // tools/attn_k64_check.hip: cycles per phase, accumulated per wave, written to a.Opart behind the loop
    unsigned tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = (unsigned)__builtin_amdgcn_s_memtime();      // (32-bit, scalar registers)
#define K64_STAMP(i)                                                  \
    {                                                                 \
        __builtin_amdgcn_sched_barrier(0);                            \
        const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime(); \
        tacc[i] += now_ - tprev;                                      \
        tprev = now_;                                                 \
        __builtin_amdgcn_sched_barrier(0);                            \
    }
#else
#define K64_STAMP(i)
#endif
    auto tile = [&](const int T_in, auto stg_c, auto more_c) {
        constexpr int STG = decltype(stg_c)::value;      // the K ring stage of tile T
        constexpr bool MORE = decltype(more_c)::value;
        // (hipcc keeps the loop counter in a vector register otherwise: every copy's address then takes 64-bit vector arithmetic and its
        // uniform tests become exec-masked branches)
        const int T = __builtin_amdgcn_readfirstlane(T_in);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of K(T) has landed
        K64_STAMP(0)      // wait for K(T)
        __builtin_amdgcn_s_barrier();                      // A: ... and everybody else's; everybody is through P.V of tile T - 1
        __builtin_amdgcn_sched_barrier(0);
        K64_STAMP(1)      // barrier A
        // this tile's copies in issue order: V^T(T) (needed behind the softmax), then K(T + 1) (needed at the next barrier A)
        auto copy = [&](int c) {
            if (c < 4) issue_v(T, c >> 1, c & 1);
            else if (MORE) issue_k(T + 1, (c - 4) >> 2, (c - 4) & 3, 1 - STG);
        };
        auto wait_v = [&]() {      // V^T(T) has landed; the K copies behind it may stay in flight
            if (MORE) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                  // B
            __builtin_amdgcn_sched_barrier(0);
        };
        if (wave_idle) {
#pragma unroll
            for (int c = 0; c < 12; ++c) copy(c);
            wait_v();
            return;
        }
        // ---- logits of sub-tile u (32 keys): the instruction sequence of attn_f16x3_dma_kernel<MX, P1, PF>; copy c0 + ks goes out behind step ks ----
        auto logits = [&](const int u, const int c0, f32x16& sm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sm[r] = 0.f;
            constexpr int KU = STG * (K64_KSTAGE * 2);       // + u * 16 KB: byte offset of the sub-tile
            auto kread = [&](int ks) { return *reinterpret_cast<const f16x8*>(att_lds_raw + ka[ks] + (KU + u * (K64_SUB * 2))); };
            if (MX) {
                constexpr int PFD = 3;
                auto k8read = [&](int img, int blk, int c) {
                    return *reinterpret_cast<const i32x4*>(att_lds_raw + k8a[2 * blk + c] + (KU + u * (K64_SUB * 2) + 8192 + img * 4096));
                };
                f16x8 kf[NKS];
                // the bf8 operands go through ONE set of 16 registers (the 32-key kernel holds both 64-deep blocks at once: 16 registers this
                // kernel does not have next to the first round's P): block 1's image is read into an operand's registers as soon as the
                // instruction that used them has been issued - the two bf8 instructions of block 0 (128 cycles) cover most of that read
                i32x8 k8h_op, k8l_op;
                auto k8op = [&](int img, int blk) {
                    const i32x4 c0 = k8read(img, blk, 0), c1 = k8read(img, blk, 1);
                    return i32x8{c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                };
#pragma unroll
                for (int i = 0; i < PFD; ++i) kf[i] = kread(i);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    if (ks + PFD < NKS) kf[ks + PFD] = kread(ks + PFD);
                    if (ks == 6) k8h_op = k8op(0, 0);
                    if (ks == 7) k8l_op = k8op(1, 0);
                    sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qh[ks], sm, 0, 0, 0);
                    if (c0 + ks < 12) copy(c0 + ks);
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
                auto klread = [&](int ks) { return *reinterpret_cast<const f16x8*>(att_lds_raw + ka[ks] + (KU + u * (K64_SUB * 2) + 8192)); };
                f16x8 kh_c = kread(0), kl_c = klread(0);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    f16x8 kh_n = kh_c, kl_n = kl_c;
                    if (ks + 1 < NKS) {
                        kh_n = kread(ks + 1);
                        kl_n = klread(ks + 1);
                    }
                    sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh_c, qh[ks], sm, 0, 0, 0);
                    sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh_c, ql[MX ? 0 : ks], sm, 0, 0, 0);
                    sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl_c, qh[ks], sm, 0, 0, 0);
                    if (c0 + ks < 12) copy(c0 + ks);
                    __builtin_amdgcn_sched_barrier(0);
                    kh_c = kh_n;
                    kl_c = kl_n;
                }
            }
            if (!MORE && 2 * T + u >= last32) {       // only the sequence's last sub-tile can hold keys past S (one past it: all of them) - both in its LAST tile
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((2 * T + u) * KT + frag_row(r, hi) >= S) sm[r] = -INFINITY;
            }
        };
        // one softmax round of the 32-key kernel without its O rescale: new (lazy) reference maximum, alpha, P = 2^(s - m_new), row sums, the fp16
        // plane of P.  Returns alpha; `rescale` (uniform): some row of the wave has a new reference - O has to be multiplied by alpha before this
        // sub-tile's P.V goes into it (for every other row alpha is exactly 1)
        auto softmax = [&](f32x16& sm, f16x8 (&ph)[2], bool& rescale) {
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
            rescale = !__all(m_new == m_run);
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
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                u32x4 hq;
#pragma unroll
                for (int i = 0; i < 4; ++i) hq[i] = pk_f16_rne(sm[8 * mf + 2 * i], sm[8 * mf + 2 * i + 1]);
                ph[mf] = __builtin_bit_cast(f16x8, hq);
            }
            return alpha;
        };
        auto rescale_o = [&](const float alpha) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[n][r] *= alpha;
        };
        // O += P.V of sub-tile u: step i = 4 mf + n, fragments three steps ahead (the 32-key kernel's order per accumulator)
        auto pv = [&](const int u, const f16x8 (&ph)[2]) {
            constexpr int PFD = 3;
            auto vread = [&](int i) { return *reinterpret_cast<const f16x8*>(att_lds_raw + va[i >> 2] + (u * 8192 + (i & 3) * 2048)); };
            f16x8 vf[2 * NT];
#pragma unroll
            for (int i = 0; i < PFD; ++i) vf[i] = vread(i);
#pragma unroll
            for (int i = 0; i < 2 * NT; ++i) {
                if (i + PFD < 2 * NT) vf[i + PFD] = vread(i + PFD);
                ot[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[i], ph[i >> 2], ot[i & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        f16x8 pa[2], pb[2];
        bool resc_a, resc_b;
        float alpha_b;
        {
            f32x16 sm;
            logits(0, 0, sm);
            K64_STAMP(2)      // logits (both sub-tiles)
            const float alpha_a = softmax(sm, pa, resc_a);
            if (resc_a) rescale_o(alpha_a);      // O holds every earlier tile's P.V: the 32-key kernel's order
            K64_STAMP(3)      // softmax (both)
        }
        {
            f32x16 sm;
            logits(1, 8, sm);
            K64_STAMP(2)
            alpha_b = softmax(sm, pb, resc_b);
            K64_STAMP(3)
        }
        wait_v();
        K64_STAMP(4)      // wait for V^T(T) + barrier B
        pv(0, pa);
        if (resc_b) rescale_o(alpha_b);          // (rare) the second round's rescale: only after the first round's P.V is in O
        pv(1, pb);
        K64_STAMP(5)      // P.V (both)
    };
    {
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        int T = 0;
        for (; T + 2 < nt64; T += 2) {
            tile(T, S0{}, std::true_type{});
            tile(T + 1, S1{}, std::true_type{});
        }
        if (T + 1 < nt64) {
            tile(T, S0{}, std::true_type{});
            tile(T + 1, S1{}, std::false_type{});
        } else {
            tile(T, S0{}, std::false_type{});
        }
    }

#ifdef ATT_K64_TRACE
    if (a.Opart && lane == 0 && !wave_idle) {
        unsigned long long* t = reinterpret_cast<unsigned long long*>(a.Opart) + ((size_t)blockIdx.x * 4 + wid) * 8;
        for (int i = 0; i < 6; ++i) t[i] = tacc[i];
    }
#endif
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

// F16MX (bf8 K images + the bf8 image of Q_lo) and F16X2 launches with one fp16 plane of P and no key split
inline bool attn_k64_applies(const AttnHArgs& a) {
    if (!a.x2 || a.nsplit != 1 || tune().attn_mx == 1 || tune().attn_pf == 2) return false;
    if (a.K8h && !a.Q8l) return false;      // ("attn_mx" = 3: Q_lo as an fp16 plane - the 32-key kernel's A/B path)
    return tune().attn_k64 == 1;
}

inline void launch_attn_k64(const AttnHArgs& a, int nseq, int nqt, hipStream_t st) {
    const dim3 grid(nqt * a.nhead * nseq);
    if (a.K8h) {
        static DevSeen seen;
        const auto kern = &attn_k64_kernel<true>;
        if (auto once_ = first_use_on_device(seen))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kern, grid, dim3(256), tune().attn_one_wg ? 160 * 1024 : ATT_K64_LDS, st, a, nqt);
    } else {
        static DevSeen seen;
        const auto kern = &attn_k64_kernel<false>;
        if (auto once_ = first_use_on_device(seen))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kern, grid, dim3(256), tune().attn_one_wg ? 160 * 1024 : ATT_K64_LDS, st, a, nqt);
    }
}

}  // namespace jmid
