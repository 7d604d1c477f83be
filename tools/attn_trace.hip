// Poor man's thread trace of the attention DMA kernel: per-wave s_memtime deltas per phase of the key-tile loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I safe-interactive-crowdnav_amd/csrc tools/attn_trace.hip -o build/attn_trace
#include "attn_f16x3.hpp"
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
#include <map>
using namespace jmid;
int main(int argc, char** argv) {
    const int nseq = argc > 1 ? atoi(argv[1]) : 51, S = argc > 2 ? atoi(argv[2]) : 1200, abl = argc > 3 ? atoi(argv[3]) : 0;
    const int d = 512, nhead = 4, HD = 128, Spad = vt_spad(S);
    const size_t M = (size_t)nseq * S;
    // operands as the pipeline produces them: x ~ N(0, sigma) split into hi = fp16(x), lo = fp16(x - hi)
    auto alloc_pair = [&](size_t n, float sigma, half_t** ph, half_t** pl) {
        std::vector<_Float16> h(n), l(n);
        for (size_t i = 0; i < n; ++i) {
            float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
            float x = sigma * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
            h[i] = (_Float16)x;
            l[i] = (_Float16)(x - (float)h[i]);
        }
        hipMalloc(ph, n * 2); hipMemcpy(*ph, h.data(), n * 2, hipMemcpyHostToDevice);
        hipMalloc(pl, n * 2); hipMemcpy(*pl, l.data(), n * 2, hipMemcpyHostToDevice);
    };
    AttnHArgs a{};
    half_t *qh, *ql, *kh, *kl, *vh, *vl, *oh, *ol;
    alloc_pair(M * d, 1.4426950f / sqrtf(128.f) * 1.5f, &qh, &ql);   // Q arrives pre-scaled by log2(e)/sqrt(hd)
    alloc_pair(M * d, 1.0f, &kh, &kl);
    alloc_pair((size_t)nseq * nhead * HD * Spad, 1.0f, &vh, &vl);
    alloc_pair(blk_plane_elems(M, d), 1.0f, &oh, &ol);
    a.Qhi = qh; a.Qlo = ql; a.Khi = kh; a.Klo = kl; a.Vthi = vh; a.Vtlo = vl; a.Ohi = oh; a.Olo = ol;
    a.S = S; a.Spad = Spad; a.d = d; a.nhead = nhead; a.scale = 1.f; a.nsplit = 1;
    hipMalloc(&a.range_flag, 4); hipMemset(a.range_flag, 0, 4);
    const int nqt = (S + 127) / 128, nblk = nqt * nhead * nseq;
    unsigned long long* trace; hipMalloc(&trace, (size_t)nblk * 4 * 12 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f16x3_dma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_DMA_LDS);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f16x3_dma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_DMA_LDS);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms_plain = 0, ms_trace = 0;
    const int nrep = argc > 4 ? atoi(argv[4]) : 3;     // more repetitions = warm clocks / steady-state power management
    for (int rep = 0; rep < nrep; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(attn_f16x3_dma_kernel<false>, dim3(nblk), dim3(256), ATT_DMA_LDS, 0, a, nqt, abl, (unsigned long long*)nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_plain, e0, e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(attn_f16x3_dma_kernel<true>, dim3(nblk), dim3(256), ATT_DMA_LDS, 0, a, nqt, abl, trace);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_trace, e0, e1);
    }
    std::vector<unsigned long long> t((size_t)nblk * 4 * 12);
    hipMemcpy(t.data(), trace, t.size() * 8, hipMemcpyDeviceToHost);
    const char* names[7] = {"prologue", "wait vmcnt", "barrier", "issue DMA", "QK^T mfma", "softmax+split", "PV mfma"};
    double sum[7] = {0}; double life = 0, life_rt = 0; unsigned long long tmin = ~0ull, tmax = 0;
    const size_t nw = (size_t)nblk * 4;
    std::map<unsigned, int> per_cu;
    for (size_t w = 0; w < nw; ++w) {
        for (int i = 0; i < 7; ++i) sum[i] += (double)t[w * 12 + i];
        life += (double)(t[w * 12 + 8] - t[w * 12 + 7]);
        life_rt += (double)(t[w * 12 + 11] - t[w * 12 + 10]);
        tmin = std::min(tmin, t[w * 12 + 7]); tmax = std::max(tmax, t[w * 12 + 8]);
    }
    const int ntiles = (S + 31) / 32;
    printf("nseq=%d S=%d abl=%d: %d workgroups, kernel %.1f us plain, %.1f us traced; s_memtime span %.0f ticks -> %.1f MHz\n", nseq, S, abl,
           nblk, ms_plain * 1e3, ms_trace * 1e3, (double)(tmax - tmin), (double)(tmax - tmin) / (ms_trace * 1e3));
    printf("wave lifetime avg %.0f ticks = %.1f us (100 MHz counter) -> shader clock %.0f MHz   (%d tiles -> %.0f ticks/tile)\n", life / nw,
           life_rt / nw / 100.0, life / life_rt * 100.0, ntiles, life / nw / ntiles);
    for (int i = 0; i < 7; ++i)
        printf("  %-14s %9.0f ticks/wave  %7.1f ticks/tile  %5.1f %%\n", names[i], sum[i] / nw, sum[i] / nw / (i ? ntiles : 1), 100.0 * sum[i] / life);
    // first workgroup's wave 0 for a feel of variance
    for (int w = 0; w < 4; ++w) {
        printf("  wg0 wave%d:", w);
        for (int i = 0; i < 7; ++i) printf(" %7llu", t[w * 12 + i]);
        printf("  hwid %llx\n", t[w * 12 + 9]);
    }
    return 0;
}
