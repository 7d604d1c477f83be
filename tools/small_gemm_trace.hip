// Where does a small-launch GEMM spend its time?  Wall-clock stamps (s_memrealtime, 100 MHz) of every workgroup of
// gemm_small_kernel at its phase boundaries, for the shapes of a one-scene step (M = 1200): entry, ring primed (all look-ahead
// copies issued), first stage landed, K loop done, epilogue stores issued, stores retired.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DJMID_SMALL_TRACE -I safe-interactive-crowdnav_amd/csrc tools/small_gemm_trace.hip -o build/small_gemm_trace
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "attn_f16x3.hpp"
#include "gemm_small.hpp"
using namespace jmid;

__global__ void dirty_kernel(unsigned* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0x11111111u;
}

template <int EPI, int OUT, int WC>
void run(const char* name, int M, int N, int K) {
    using C = SmCfg<SM_MX, WC>;
    const size_t pa = blk_plane_elems(M, K), pw = blk_plane_elems(N, K);
    half_t *ah, *wh, *ch;
    unsigned char* w8;
    float *bias, *Cc;
    hipMalloc(&ah, pa * 2); hipMalloc(&wh, pw * 2); hipMalloc(&w8, (size_t)N * K); hipMalloc(&bias, N * 4);
    hipMalloc(&Cc, (size_t)M * N * 4); hipMalloc(&ch, blk_plane_elems(M, N) * 2 * 8);
    hipMemset(ah, 0x11, pa * 2); hipMemset(wh, 0x12, pw * 2); hipMemset(w8, 0x13, (size_t)N * K); hipMemset(bias, 0, N * 4);
    int* flag; hipMalloc(&flag, 4); hipMemset(flag, 0, 4);
    GemmHArgs g{};
    g.Ahi = ah; g.Alo = ah; g.Whi = wh; g.Wlo = wh; g.W8 = w8; g.bias = bias; g.C = Cc; g.ldc = N; g.M = M; g.N = N; g.K = K;
    g.Chi = ch; g.Clo = ch + blk_plane_elems(M, N); g.Khi = ch + 2 * blk_plane_elems(M, N); g.Klo = ch + 3 * blk_plane_elems(M, N);
    g.Vthi = ch + 4 * blk_plane_elems(M, N); g.Vtlo = ch + 5 * blk_plane_elems(M, N);
    g.d = 512; g.hd = 128; g.S = M; g.Spad = vt_spad(M); g.vt_direct = 1; g.qscale = 0.1f; g.range_flag = flag; g.x2 = 1;
    g.K8h = reinterpret_cast<unsigned char*>(ch + 6 * blk_plane_elems(M, N)); g.K8l = g.K8h + (size_t)M * 512;
    g.Q8l = g.K8l + (size_t)M * 512;
    unsigned* cnt; hipMalloc(&cnt, 1024); hipMemset(cnt, 0, 1024);
    float *gam; hipMalloc(&gam, 2048); hipMemset(gam, 0, 2048);
    g.ln_gamma = gam; g.ln_beta = gam; g.ln_xh = ch; g.ln_xl = ch + blk_plane_elems(M, N); g.ln_xl8 = reinterpret_cast<unsigned char*>(g.ln_xl);
    g.ln_cnt = cnt; g.ln_eps = 1e-5f; g.ln_no_lo = 0;
    const int ntm = (M + 63) / 64, ntn = N / C::BN, nwg = ntm * ntn;
    unsigned long long* tr;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_small_kernel<EPI, OUT, SM_MX, WC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)C::LDS_BYTES);
    hipMalloc(&tr, (size_t)nwg * 64 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_small_trace), &tr, sizeof(tr));
    for (int abl : {JMID_SMALL_ABL}) {
        const int pn = 1, reps = 300;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto chain = [&](int n) {
            for (int it = 0; it < n; ++it) {
                // a dependent predecessor that dirties the A plane (as the real chain does), so the operands are not L2-warm
                hipLaunchKernelGGL(dirty_kernel, dim3(256), dim3(256), 0, 0, reinterpret_cast<unsigned*>(ah), pa / 2);
                hipLaunchKernelGGL((gemm_small_kernel<EPI, OUT, SM_MX, WC>), dim3(nwg), dim3(C::NT), C::LDS_BYTES, 0, g, ntm, ntn, ntn / pn, 4, 0u, 0u);
            }
        };
        chain(reps);                 // warm clocks
        hipEventRecord(e0);
        chain(reps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> t((size_t)nwg * 64);
        hipMemcpy(t.data(), tr, t.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < nwg; ++b) t0 = std::min(t0, t[b * 64]);
        double avg[9] = {0}, mx[9] = {0}, mhz = 0;
        for (int b = 0; b < nwg; ++b) {
            for (int i = 0; i < 9; ++i) {
                if (i >= 6 && t[b * 64 + i] < t[b * 64 + 5]) continue;      // (not a last arriver: stale stamp)
                const double v = (double)(t[b * 64 + i] - t0) * 0.01;     // us
                avg[i] += v / nwg;
                mx[i] = std::max(mx[i], v);
            }
            mhz += (double)(t[b * 64 + 8 + 3] - t[b * 64 + 8 + 2]) / ((double)(t[b * 64 + 3] - t[b * 64 + 2]) * 0.01) / nwg;
        }
        printf("%-12s abl=%-2d M=%d N=%d K=%d WC=%d KB=%d NS=%d wgs=%d  pair %.2f us, %.0f MHz | us since first entry (avg / max): entry %.2f/%.2f  primed %.2f/%.2f  first stage %.2f/%.2f  "
               "K loop done %.2f/%.2f  stores issued %.2f/%.2f  retired %.2f/%.2f  tail: loads in %.2f, stats %.2f, done %.2f/%.2f\n",
               name, abl, M, N, K, WC, C::KB, C::NS, nwg, ms * 1e3 / reps, mhz, avg[0], mx[0], avg[1], mx[1], avg[2], mx[2], avg[3], mx[3], avg[4], mx[4], avg[5], mx[5], avg[7] * nwg / std::max(1, OUT == OUT_LN ? (M + 63) / 64 : nwg), avg[8] * nwg / std::max(1, OUT == OUT_LN ? (M + 63) / 64 : nwg), avg[6] * nwg / std::max(1, OUT == OUT_LN ? (M + 63) / 64 : nwg), mx[6]);
    }
    hipFree(ah); hipFree(wh); hipFree(w8); hipFree(bias); hipFree(Cc); hipFree(ch); hipFree(tr); hipFree(flag);
}

int main() {
    run<EPI_BIAS, OUT_F32, 2>("out_proj", 1200, 512, 512);
    run<EPI_BIAS, OUT_LN, 2>("out_proj+LN", 1200, 512, 512);
    run<EPI_BIAS, OUT_LN, 2>("linear2+LN", 1200, 512, 1024);
    run<EPI_BIAS, OUT_F32, 2>("linear2", 1200, 512, 1024);
    run<EPI_BIAS_RELU, OUT_SPLIT, 4>("linear1", 1200, 1024, 512);
    run<EPI_BIAS, OUT_QKV, 4>("in_proj", 1200, 1536, 512);
    return 0;
}
