// attn_pp_kernel (8-wave ping-pong workgroups, attn_pp.hpp) against attn_f16x3_dma_kernel<MX, P1, PF> (two independent 4-wave workgroups
// per CU) on the same random planes: bitwise comparison of the O_hi plane, then the time per launch of each.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops -w \
//         -DJMID_DIAGNOSTICS -DJMID_EXPERIMENTS -I safe-interactive-crowdnav_amd/csrc -I include tools/attn_pp_check.hip -o build/attn_pp_check
//   build/attn_pp_check [nseq = 51] [S = 1200] [reps = 20]      (-DJMID_DIAGNOSTICS: the experiment is compiled in that flavour only;
//   -DAQ_TRACE: cycles per phase of the one-wave kernel; AQ_TRACE_TWO_WAVE=1 in the environment: of the two-wave kernel)
#include "attn_f16x3.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
using namespace jmid;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int nseq = argc > 1 ? atoi(argv[1]) : 51, S = argc > 2 ? atoi(argv[2]) : 1200, reps = argc > 3 ? atoi(argv[3]) : 20;
    const int mode = argc > 4 ? atoi(argv[4]) : 0;      // 0 = F16MX operands (bf8 K images), 1 = F16X2 (fp16 K_lo, Q_lo), 2 = F16X3 (+ V^T_lo, P_lo)
    const int d = 512, nhead = 4, hd = 128, Spad = vt_spad(S);
    const size_t M = (size_t)nseq * S, Mpad = (M + 127) / 128 * 128 + 128;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<half_t> qh(Mpad * d), kh(Mpad * d), vt((size_t)nseq * d * Spad), ql(Mpad * d), kl(Mpad * d), vtl((size_t)nseq * d * Spad);
    std::vector<unsigned char> q8l(Mpad * d), k8h(Mpad * d), k8l(Mpad * d);
    auto top = [](half_t v) { return (unsigned char)((__builtin_bit_cast(unsigned short, v) + 0x80u) >> 8); };
    for (size_t i = 0; i < qh.size(); ++i) {
        const float q = nd(rng) * 0.35f, k = nd(rng);
        qh[i] = (half_t)q; kh[i] = (half_t)k;
        ql[i] = (half_t)(q - (float)qh[i]);
        kl[i] = (half_t)(k - (float)kh[i]);
        q8l[i] = top((half_t)(q - (float)qh[i]));
        k8h[i] = top(kh[i]);
        k8l[i] = top((half_t)(k - (float)kh[i]));
    }
    for (size_t i = 0; i < vt.size(); ++i) { const float v = nd(rng); vt[i] = (half_t)v; vtl[i] = (half_t)(v - (float)vt[i]); }
    half_t *dQ, *dK, *dV, *dO[2], *dQl, *dKl, *dVl, *dOl;
    unsigned char *dQ8, *dK8h, *dK8l;
    int* flag;
    const size_t oelems = blk_plane_elems(M, d) + 128 * d;
    CK(hipMalloc(&dQ, qh.size() * 2)); CK(hipMalloc(&dK, kh.size() * 2)); CK(hipMalloc(&dV, vt.size() * 2));
    CK(hipMalloc(&dQl, ql.size() * 2)); CK(hipMalloc(&dKl, kl.size() * 2)); CK(hipMalloc(&dVl, vtl.size() * 2));
    CK(hipMemcpy(dQl, ql.data(), ql.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dKl, kl.data(), kl.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dVl, vtl.data(), vtl.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&dQ8, q8l.size())); CK(hipMalloc(&dK8h, k8h.size())); CK(hipMalloc(&dK8l, k8l.size()));
    CK(hipMalloc(&dO[0], oelems * 2)); CK(hipMalloc(&dO[1], oelems * 2)); CK(hipMalloc(&dOl, oelems * 2)); CK(hipMalloc(&flag, 4));
    CK(hipMemcpy(dQ, qh.data(), qh.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dK, kh.data(), kh.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dV, vt.data(), vt.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dQ8, q8l.data(), q8l.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dK8h, k8h.data(), k8h.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dK8l, k8l.data(), k8l.size(), hipMemcpyHostToDevice));
    CK(hipMemset(flag, 0, 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    Tuning tn[2];
    tn[0].attn_pp = 2;       // the two-wave kernel
    tn[1].attn_pp = 1;       // the ping-pong kernel
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int round = 0; round < 3; ++round)      // alternating: the chip's clock follows its power budget, a kernel timed first runs faster
    for (int v = 0; v < 2; ++v) {
        if (round == 0) CK(hipMemset(dO[v], 0, oelems * 2));
        AttnHArgs a{dQ, reinterpret_cast<half_t*>(dQ8), dK, nullptr, dV, nullptr, dO[v], nullptr, S, Spad, d, nhead, 1.f, flag, 1, nullptr, nullptr, 1,
                    dK8h, dK8l, dQ8};
        if (mode >= 1) { a.Qlo = dQl; a.Klo = dKl; a.K8h = a.K8l = a.Q8l = nullptr; }
        if (mode == 2) { a.Vtlo = dVl; a.Olo = dOl; a.x2 = 0; }
        TuneScope ts(&tn[v]);
        CK(launch_attn_f16x3(a, nseq, hd, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CK(launch_attn_f16x3(a, nseq, hd, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double fl = 4.0 * nseq * (double)S * S * d;
        printf("%s: %.4f ms per launch  (%.0f TFLOP/s algorithmic)\n", v ? "8-wave ping-pong workgroups  " : "two 4-wave workgroups per CU ", ms / reps,
               fl / (ms / reps * 1e-3) * 1e-12);
    }
#ifdef ATT_PP_TRACE
    {
        const int nwg = ((S + 255) / 256) * nhead * nseq;
        unsigned long long* tr;
        CK(hipMalloc(&tr, (size_t)nwg * 8 * 4 * 8));
        CK(hipMemset(tr, 0, (size_t)nwg * 8 * 4 * 8));
        AttnHArgs a{dQ, reinterpret_cast<half_t*>(dQ8), dK, nullptr, dV, nullptr, dO[1], nullptr, S, Spad, d, nhead, 1.f, flag, 1,
                    reinterpret_cast<float*>(tr), nullptr, 1, dK8h, dK8l, dQ8};
        if (mode >= 1) { a.Qlo = dQl; a.Klo = dKl; a.K8h = a.K8l = a.Q8l = nullptr; }
        if (mode == 2) { a.Vtlo = dVl; a.Olo = dOl; a.x2 = 0; }
        TuneScope ts(&tn[1]);
        for (int rep = 0; rep < 3; ++rep) CK(launch_attn_f16x3(a, nseq, hd, st));
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> h((size_t)nwg * 8 * 4);
        CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
        double sum[2][4] = {{0}}; int nw[2] = {0, 0};
        for (int w = 0; w < nwg * 8; ++w) { if (!h[(size_t)w * 4]) continue; const int g = (w & 7) >> 2; ++nw[g]; for (int i = 0; i < 4; ++i) sum[g][i] += (double)h[(size_t)w * 4 + i]; }
        const char* nm[4] = {"compute segment", "vector segment (work)", "wait vmcnt", "barrier"};
        const int nt = (S + 31) / 32;
        for (int g = 0; g < 2; ++g) {
            double tot = 0;
            for (int i = 0; i < 4; ++i) tot += sum[g][i];
            printf("trace, group %d: %d active waves, %d key tiles; cycles per wave per tile:\n", g, nw[g], nt);
            for (int i = 0; i < 4; ++i) printf("  %-24s %8.1f  %5.1f %%\n", nm[i], sum[g][i] / nw[g] / nt, 100 * sum[g][i] / tot);
            printf("  total %.1f cycles per tile\n", tot / nw[g] / nt);
        }
    }
#endif
    if (getenv("AQ_TRACE_TWO_WAVE")) {      // cycles per phase of the two-wave kernel's key-tile loop (its TRACE instantiation, F16MX operands)
        const int nqt = (S + 127) / 128, nblk = nqt * nhead * nseq;
        unsigned long long* tr;
        CK(hipMalloc(&tr, (size_t)nblk * 4 * 12 * 8));
        const auto kern = &attn_f16x3_dma_kernel<true, true, true, false, true, true>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_DMA_LDS));
        AttnHArgs a{dQ, reinterpret_cast<half_t*>(dQ8), dK, nullptr, dV, nullptr, dO[0], nullptr, S, Spad, d, nhead, 1.f, flag, 1, nullptr, nullptr, 1,
                    dK8h, dK8l, dQ8};
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), ATT_DMA_LDS, st, a, nqt, 0, tr);
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> t((size_t)nblk * 4 * 12);
        CK(hipMemcpy(t.data(), tr, t.size() * 8, hipMemcpyDeviceToHost));
        const char* names[8] = {"prologue", "wait vmcnt", "barrier", "issue DMA", "QK^T mfma", "softmax", "PV mfma", "setup"};
        const int idx[8] = {0, 1, 2, 3, 4, 5, 6, 9};
        double sum[8] = {0};
        size_t nw = 0;
        for (size_t w = 0; w < (size_t)nblk * 4; ++w) {
            if (!t[w * 12 + 4]) continue;      // idle wave
            ++nw;
            for (int i = 0; i < 8; ++i) sum[i] += (double)t[w * 12 + idx[i]];
        }
        const int nt = (S + 31) / 32;
        double tot = 0;
        for (int i = 1; i < 7; ++i) tot += sum[i];
        printf("two-wave kernel, traced: %zu active waves; cycles per wave per key tile (32 queries):\n", nw);
        for (int i = 1; i < 7; ++i) printf("  %-12s %8.1f  %5.1f %%\n", names[i], sum[i] / nw / nt, 100 * sum[i] / tot);
        printf("  total %.1f per tile;  setup %.0f + prologue %.0f cycles per wave\n", tot / nw / nt, sum[7] / nw, sum[0] / nw);
    }
    std::vector<half_t> o0(oelems), o1(oelems);
    CK(hipMemcpy(o0.data(), dO[0], oelems * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o1.data(), dO[1], oelems * 2, hipMemcpyDeviceToHost));
    size_t ndiff = 0;
    int qhist[64] = {0}, chist[8] = {0}, shown = 0;
    double maxd = 0;
    for (size_t m = 0; m < M; ++m)
        for (int c = 0; c < d; ++c) {
            const size_t o = blk_index((int)m, c, d);
            if (__builtin_bit_cast(unsigned short, o0[o]) != __builtin_bit_cast(unsigned short, o1[o])) {
                ++ndiff;
                ++qhist[(m % S) % 64];
                ++chist[(c % 128) / 16];
                const double dd = fabs((double)(float)o0[o] - (double)(float)o1[o]);
                if (dd > maxd) maxd = dd;
                if (shown < 6) { printf("  token %zu (q %zu of its sequence) col %d: %g vs %g\n", m, m % S, c, (float)o0[o], (float)o1[o]); ++shown; }
            }
        }
    printf("differing elements: %zu of %zu, max |d| %g\n", ndiff, M * d, maxd);
    if (ndiff) {
        printf("  by query %% 64:");
        for (int i = 0; i < 64; ++i) printf(" %d", qhist[i]);
        printf("\n  by (head dim %% 128) / 16:");
        for (int i = 0; i < 8; ++i) printf(" %d", chist[i]);
        printf("\n");
    }
    int f = 0;
    CK(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost));
    printf("range flag %d\n", f);
    return ndiff != 0;
}
