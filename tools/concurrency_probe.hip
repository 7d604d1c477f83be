// Does embed_kernel give the same bits when it runs concurrently with the attention kernel on another stream?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I safe-interactive-crowdnav_amd/csrc tools/concurrency_probe.hip -o build/concurrency_probe
#include "attn_f16x3.hpp"
#include "elementwise.hpp"
#include "gemm_f16x3.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace jmid;
// a pure MFMA loop (no LDS, no memory traffic) and a pure LDS-DMA loop as alternative co-runners
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, const f16x8* frags) {
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = frags[(i * 256 + threadIdx.x) % 2048]; b[i] = frags[((i + 4) * 256 + threadIdx.x) % 2048]; }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(j + it) & 3], b[j], acc[j], 0, 0, 0);
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void valu_loop(float* out, int iters) {
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    for (int it = 0; it < iters; ++it) { y = fmaf(y, 1.0001f, x); x = fmaf(x, 0.9999f, 1e-4f); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}
int main(int argc, char** argv) {
    const int elds = argc > 6 ? atoi(argv[6]) : 0;    // unused dynamic LDS requested by embed_kernel
    const int alds = argc > 7 ? atoi(argv[7]) : 0;    // extra (unused) dynamic LDS of the attention co-runner: 32768 -> one workgroup per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(&embed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    const int nseq = 8, S = 1200, d = 512, nhead = 4, HD = 128, Spad = vt_spad(S), T = 12, A = 5, K = 20;
    const size_t M = (size_t)nseq * S;
    auto dev_rand_h = [&](size_t n, float sc) {
        std::vector<_Float16> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = (_Float16)(sc * ((rand() & 1023) - 512) / 512.0f);
        half_t* p; hipMalloc(&p, n * 2); hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice); return p;
    };
    auto dev_rand_f = [&](size_t n, float sc) {
        std::vector<float> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = sc * ((rand() & 1023) - 512) / 512.0f;
        float* p; hipMalloc(&p, n * 4); hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice); return p;
    };
    AttnHArgs a{};
    a.Qhi = dev_rand_h(M * d, 0.2f); a.Qlo = dev_rand_h(M * d, 1e-4f); a.Khi = dev_rand_h(M * d, 1.f); a.Klo = dev_rand_h(M * d, 4e-4f);
    a.Vthi = dev_rand_h((size_t)nseq * nhead * HD * Spad, 1.f); a.Vtlo = dev_rand_h((size_t)nseq * nhead * HD * Spad, 4e-4f);
    a.Ohi = dev_rand_h(blk_plane_elems(M, d), 1.f); a.Olo = dev_rand_h(blk_plane_elems(M, d), 1.f);
    a.S = S; a.Spad = Spad; a.d = d; a.nhead = nhead; a.scale = 1.f; a.nsplit = 1;
    hipMalloc(&a.range_flag, 4); hipMemset(a.range_flag, 0, 4);
    // embed operands
    const int hyp_ld = 1796, EA = nseq * A;
    EmbedArgs e{};
    e.x = dev_rand_f(M * 2, 1.f); e.W1 = dev_rand_f(d * 2, 0.5f); e.b1 = dev_rand_f(d, 0.5f); e.pe = dev_rand_f(24 * d, 1.f);
    e.hyp = dev_rand_f((size_t)EA * hyp_ld, 1.f); e.thyp = dev_rand_f(hyp_ld, 1.f);
    e.X = nullptr; e.M = (int)M; e.d = d; e.hyp_ld = hyp_ld; e.goff = 0; e.boff = d; e.rmap = RowMap{T, A, K * A};
    hipMalloc(&e.Xh, blk_plane_elems(M, d) * 2); hipMalloc(&e.Xl, blk_plane_elems(M, d) * 2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f16x3_dma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_DMA_LDS + alds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f16x3_dma_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_DMA_LDS + alds);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    // consumer of the embed output on the same stream: C = X . W^T (256x128 LDS-DMA GEMM, fp32 out)
    GemmHArgs g{};
    g.Ahi = e.Xh; g.Alo = e.Xl; g.M = (int)M; g.N = 1536; g.K = d;
    g.Whi = dev_rand_h(blk_plane_elems(1536, d), 8.f); g.Wlo = dev_rand_h(blk_plane_elems(1536, d), 4e-3f);
    g.bias = dev_rand_f(1536, 1.f); g.ldc = 1536; g.range_flag = a.range_flag;
    hipMalloc(&g.C, M * 1536 * 4);
    std::vector<float> cref(M * 1536), ccur(M * 1536);
    const int nqt = (S + 127) / 128, nblk = nqt * nhead * nseq;
    const long total = (long)M * (d / 4);
    const int eblocks = (int)std::min<long>((total + 255) / 256, 4096);
    const size_t pe = blk_plane_elems(M, d);
    std::vector<_Float16> ref_h(pe), ref_l(pe), cur_h(pe), cur_l(pe);
    std::vector<_Float16> oref(pe), ocur(pe);
    // reference: each kernel alone
    hipLaunchKernelGGL(embed_kernel, dim3(eblocks), dim3(256), elds, s2, e);
    (void)launch_gemm_h_dma256<EPI_BIAS, OUT_F32>(g, s2);
    hipDeviceSynchronize();
    hipMemcpy(ref_h.data(), e.Xh, pe * 2, hipMemcpyDeviceToHost); hipMemcpy(ref_l.data(), e.Xl, pe * 2, hipMemcpyDeviceToHost);
    hipMemcpy(cref.data(), g.C, M * 1536 * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(attn_f16x3_dma_kernel<false>, dim3(nblk), dim3(256), ATT_DMA_LDS, s1, a, nqt, 0, (unsigned long long*)nullptr);
    hipDeviceSynchronize(); hipMemcpy(oref.data(), a.Ohi, pe * 2, hipMemcpyDeviceToHost);
    const int mode = argc > 1 ? atoi(argv[1]) : 0; const int abl = argc > 2 ? atoi(argv[2]) : 0; const int niter = argc > 3 ? atoi(argv[3]) : 100; const int use_gemm = argc > 4 ? atoi(argv[4]) : 1; const int copy_c = argc > 5 ? atoi(argv[5]) : 1;   // co-runner: 0 attention, 1 pure MFMA loop, 2 pure VALU loop, 3 nothing
    float* sink; hipMalloc(&sink, 2048 * 256 * 4);
    f16x8* fr = reinterpret_cast<f16x8*>(dev_rand_h(2048 * 8, 1.f));
    auto corunner = [&]() {
        if (mode == 0) hipLaunchKernelGGL(attn_f16x3_dma_kernel<false>, dim3(nblk), dim3(256), ATT_DMA_LDS + alds, s1, a, nqt, abl, (unsigned long long*)nullptr);
        else if (mode == 6) hipLaunchKernelGGL((attn_f16x3_dma_kernel<false, true>), dim3(nblk), dim3(256), ATT_DMA_LDS + alds, s1, a, nqt, abl, (unsigned long long*)nullptr);
        else if (mode == 5) hipLaunchKernelGGL(attn_f16x3_kernel<128>, dim3((S + 127) / 128, nhead, nseq), dim3(256), 0, s1, a);
        else if (mode == 1) hipLaunchKernelGGL(mfma_loop, dim3(1024), dim3(256), 0, s1, sink, 800, fr);
        else if (mode == 2) hipLaunchKernelGGL(valu_loop, dim3(2048), dim3(256), 0, s1, sink, 20000);
    };
    int bad_embed = 0, bad_attn = 0;
    for (int it = 0; it < niter; ++it) {
        hipMemsetAsync(e.Xh, 0xff, pe * 2, s2); hipMemsetAsync(e.Xl, 0xff, pe * 2, s2); hipDeviceSynchronize();
        corunner();
        hipLaunchKernelGGL(embed_kernel, dim3(eblocks), dim3(256), elds, s2, e);
        if (use_gemm) (void)launch_gemm_h_dma256<EPI_BIAS, OUT_F32>(g, s2);
        corunner();
        hipDeviceSynchronize();
        if (copy_c) hipMemcpy(ccur.data(), g.C, M * 1536 * 4, hipMemcpyDeviceToHost);
        if (copy_c) { size_t nc = 0; for (size_t i = 0; i < M * 1536; ++i) nc += memcmp(&ccur[i], &cref[i], 4) != 0;
          if (nc && 0) printf("iter %d: GEMM output (consumer of embed on the same stream) differs in %zu elements\n", it, nc); }
        hipMemcpy(cur_h.data(), e.Xh, pe * 2, hipMemcpyDeviceToHost); hipMemcpy(cur_l.data(), e.Xl, pe * 2, hipMemcpyDeviceToHost);
        hipMemcpy(ocur.data(), a.Ohi, pe * 2, hipMemcpyDeviceToHost);
        size_t nb = 0, first = 0;
        for (size_t i = 0; i < (size_t)M * d; ++i) {   // only rows < M matter; blocked index covers them all for M % 128 == 0
            if (memcmp(&cur_h[i], &ref_h[i], 2) || memcmp(&cur_l[i], &ref_l[i], 2)) { if (!nb) first = i; ++nb; }
        }
        size_t na = 0;
        for (size_t i = 0; i < (size_t)M * d; ++i) na += memcmp(&ocur[i], &oref[i], 2) != 0;
        if (nb && bad_embed < 3) {
            // decode the blocked index of the differing elements: tile (rb, kb), row r, stored chunk, element
            size_t shown = 0; printf("  iter %d: %zu differing elements\n", it, nb);
            for (size_t i = 0; i < (size_t)M * d && shown < 72; ++i)
                if (memcmp(&cur_h[i], &ref_h[i], 2) || memcmp(&cur_l[i], &ref_l[i], 2)) {
                    const size_t tile = i / 4096, in = i % 4096; const int r = in / 32, pos = in % 32;
                    const int rb = tile / (d / 32), kb = tile % (d / 32);
                    const int chunk = (pos >> 3) ^ ((r >> 2) & 3);
                    unsigned short hb, lb, rhb, rlb; memcpy(&hb, &cur_h[i], 2); memcpy(&lb, &cur_l[i], 2); memcpy(&rhb, &ref_h[i], 2); memcpy(&rlb, &ref_l[i], 2);
                    printf("    idx %zu: row %d col %d  hi %04x (ref %04x)  lo %04x (ref %04x)   hi+lo %.7f ref %.7f\n", i, rb * 128 + r, kb * 32 + chunk * 8 + (pos & 7),
                           hb, rhb, lb, rlb, (float)cur_h[i] + (float)cur_l[i], (float)ref_h[i] + (float)ref_l[i]);
                    ++shown;
                }
        }
        if (nb) { ++bad_embed; if (bad_embed <= 0) printf("iter %d: embed differs in %zu elements (first index %zu, value %f vs %f)\n", it, nb, first, (float)cur_h[first], (float)ref_h[first]); }
        if (na) { ++bad_attn; if (bad_attn <= 0) printf("iter %d: attention output differs in %zu elements\n", it, na); }
    }
    printf("attn +LDS %d | embed LDS %d | gemm %d copyC %d | mode %d abl %d: concurrent runs with a different embed result: %d / %d (attention differs: %d)\n", alds, elds, use_gemm, copy_c, mode, abl, bad_embed, niter, bad_attn);
    return 0;
}
