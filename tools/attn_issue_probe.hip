// Where should the LDS-DMA copies of the next key tile go out?  The shipped F16MX attention kernel (attn_f16x3_dma_kernel<MX, P1, PF>) compiled
// with -DATT_ISSUE_AT=0 (what ships: one copy per QK^T step), 1 (all six between the last QK^T instruction and the softmax) or 2 (the same,
// after all eight V^T fragments of the tile are in registers: no LDS access of the wave while its copies fly) on the same random planes:
// time per launch, cycles per phase (the TRACE instantiation) and a checksum of the O plane (must not depend on the variant).
//   for v in 0 1 2; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops -w \
//       -DATT_ISSUE_AT=$v -I safe-interactive-crowdnav_amd/csrc -I include tools/attn_issue_probe.hip -o build/attn_issue_probe$v; done
//   build/attn_issue_probe$v [nseq = 51] [S = 1200] [reps = 20]
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
    const int nqt = (S + 127) / 128, nblk = nqt * nhead * nseq;
    AttnHArgs a{dQ, reinterpret_cast<half_t*>(dQ8), dK, nullptr, dV, nullptr, dO[0], nullptr, S, Spad, d, nhead, 1.f, flag, 1, nullptr, nullptr, 1,
                dK8h, dK8l, dQ8};
    const auto kern = &attn_f16x3_dma_kernel<false, true, true, false, true, true>;
    const auto kern_t = &attn_f16x3_dma_kernel<true, true, true, false, true, true>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_DMA_LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_t), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_DMA_LDS));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), ATT_DMA_LDS, st, a, nqt, 0, (unsigned long long*)nullptr);
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), ATT_DMA_LDS, st, a, nqt, 0, (unsigned long long*)nullptr);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = 4.0 * nseq * (double)S * S * d;
    printf("ATT_ISSUE_AT=%d: %.4f ms per launch  (%.0f TFLOP/s algorithmic)\n", ATT_ISSUE_AT, ms / reps, flops / (ms / reps) / 1e9);
    {
        unsigned long long* tr;
        CK(hipMalloc(&tr, (size_t)nblk * 4 * 12 * 8));
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern_t, dim3(nblk), dim3(256), ATT_DMA_LDS, st, a, nqt, 0, tr);
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
        printf("  traced: %zu active waves; cycles per wave per key tile:", nw);
        for (int i = 1; i < 7; ++i) printf("  %s %.0f", names[i], sum[i] / nw / nt);
        printf("  total %.0f\n", tot / nw / nt);
    }
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), ATT_DMA_LDS, st, a, nqt, 0, (unsigned long long*)nullptr);
    CK(hipStreamSynchronize(st));
    std::vector<half_t> o0(oelems);
    CK(hipMemcpy(o0.data(), dO[0], oelems * 2, hipMemcpyDeviceToHost));
    unsigned long long h = 1469598103934665603ull;
    for (size_t m = 0; m < M; ++m)
        for (int c = 0; c < d; ++c) h = (h ^ __builtin_bit_cast(unsigned short, o0[blk_index((int)m, c, d)])) * 1099511628211ull;
    printf("  O plane checksum %016llx\n", h);
    return 0;
}
