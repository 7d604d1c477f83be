// One-scene attention launch (S = 1200, 4 heads, split-KV 6: 240 workgroups) under the in-kernel trace of attn_f16x3_dma_kernel:
// cycles per phase of the key-tile loop, wave lifetime, and the launch + combine pair's wall time in a warm back-to-back chain.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I safe-interactive-crowdnav_amd/csrc tools/attn_small_trace.hip -o build/attn_small_trace
#include "attn_f16x3.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace jmid;
int main(int argc, char** argv) {
    const int nseq = argc > 1 ? atoi(argv[1]) : 1, S = 1200, ns = argc > 2 ? atoi(argv[2]) : 6;
    const int d = 512, nhead = 4, HD = 128, Spad = vt_spad(S);
    const size_t M = (size_t)nseq * S;
    auto dalloc = [&](size_t bytes, int fill) { void* p; hipMalloc(&p, bytes); hipMemset(p, fill, bytes); return p; };
    AttnHArgs a{};
    a.Qhi = (half_t*)dalloc(M * d * 2, 0x2c); a.Qlo = (half_t*)dalloc(M * d * 2, 0x10);
    a.Khi = (half_t*)dalloc(M * d * 2, 0x38); a.Klo = (half_t*)dalloc(M * d * 2, 0x30);
    a.Vthi = (half_t*)dalloc((size_t)nseq * nhead * HD * Spad * 2, 0x3a); a.Vtlo = (half_t*)dalloc((size_t)nseq * nhead * HD * Spad * 2, 0x10);
    a.Ohi = (half_t*)dalloc(blk_plane_elems(M, d) * 2, 0); a.Olo = (half_t*)dalloc(blk_plane_elems(M, d) * 2, 0);
    a.K8h = (unsigned char*)a.Klo; a.K8l = a.K8h + M * d; a.Q8l = (unsigned char*)a.Qlo;
    a.S = S; a.Spad = Spad; a.d = d; a.nhead = nhead; a.scale = 1.f; a.nsplit = ns; a.x2 = 1;
    a.Opart = (float*)dalloc((size_t)ns * M * d * 4, 0); a.MLpart = (float*)dalloc((size_t)ns * M * nhead * 2 * 4, 0);
    a.range_flag = (int*)dalloc(4, 0);
    const int nqt = (S + 127) / 128, nblk = nqt * nhead * nseq * ns;
    if (argc <= 3 || atoi(argv[3])) {      // the reciprocals launch_attn_f16x3 hands the kernel (third argument 0: run-time divisions, as before round 4)
        a.mq = fast_div_magic(nqt, nblk); a.ms = fast_div_magic(ns, nblk); a.mh = fast_div_magic(nhead, nblk); a.nseq = nseq;
    }
    unsigned long long* trace = (unsigned long long*)dalloc((size_t)nblk * 4 * 12 * 8, 0);
    auto kt = &attn_f16x3_dma_kernel<true, true, true, false, true, true>;
    auto kp = &attn_f16x3_dma_kernel<false, true, true, false, true, true>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_DMA_LDS);
    hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_DMA_LDS);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 300;
    auto chain = [&](bool traced, bool combine) {
        for (int i = 0; i < reps; ++i) {
            hipLaunchKernelGGL(traced ? kt : kp, dim3(nblk), dim3(256), ATT_DMA_LDS, 0, a, nqt, 0, traced ? trace : nullptr);
            if (combine && ns > 1) hipLaunchKernelGGL(attn_combine_kernel, dim3(std::min<size_t>((M * (d / 4) + 255) / 256, 2048)), dim3(256), 0, 0, a, M, 128);
        }
    };
    float ms_a, ms_ac, ms_t;
    chain(false, false);
    hipEventRecord(e0); chain(false, false); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_a, e0, e1);
    hipEventRecord(e0); chain(false, true); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_ac, e0, e1);
    hipEventRecord(e0); chain(true, false); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_t, e0, e1);
    std::vector<unsigned long long> t((size_t)nblk * 4 * 12);
    hipMemcpy(t.data(), trace, t.size() * 8, hipMemcpyDeviceToHost);
    const char* names[7] = {"prologue", "wait vmcnt", "barrier", "issue DMA", "QK^T mfma", "softmax+split", "PV mfma"};
    double sum[7] = {0}, life = 0, life_rt = 0;
    unsigned long long r0 = ~0ull, r1 = 0;
    const size_t nw = (size_t)nblk * 4;
    for (size_t w = 0; w < nw; ++w) {
        for (int i = 0; i < 7; ++i) sum[i] += (double)t[w * 12 + i];
        life += (double)(t[w * 12 + 8] - t[w * 12 + 7]);
        life_rt += (double)(t[w * 12 + 11] - t[w * 12 + 10]);
        r0 = std::min(r0, t[w * 12 + 10]); r1 = std::max(r1, t[w * 12 + 11]);
    }
    const double tiles = (double)((S + 31) / 32) / ns;
    printf("nseq=%d S=%d nsplit=%d: %d workgroups; per launch in a warm chain: attention %.2f us, attention + combine %.2f us, traced %.2f us\n", nseq, S, ns,
           nblk, ms_a * 1e3 / reps, ms_ac * 1e3 / reps, ms_t * 1e3 / reps);
    printf("first wave start -> last wave's loop end %.2f us; wave lifetime (start -> loop end) avg %.2f us = %.0f cycles (%.0f MHz), %.2f key tiles per wave\n",
           (double)(r1 - r0) * 0.01, life_rt / nw * 0.01, life / nw, life / life_rt * 100.0, tiles);
    double pre = 0;
    for (size_t w = 0; w < nw; ++w) pre += (double)t[w * 12 + 9];
    printf("  %-14s %8.0f cycles/wave (entry -> first copy issued: arguments, tile arithmetic, Q loads requested)\n", "setup", pre / nw);
    for (int i = 0; i < 7; ++i)
        printf("  %-14s %8.0f cycles/wave  %7.1f cycles/tile  %5.1f %%\n", names[i], sum[i] / nw, sum[i] / nw / (i ? tiles : 1), 100.0 * sum[i] / life);
    return 0;
}
