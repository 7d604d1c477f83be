// Second round of the co-residency hunt (docs/NOTEBOOK.md section 3): with the packed-fp32 instructions gone, which kernel next to which
// co-runner still computes differently?  Victim: the production embed_kernel (built without packed fp32, like the library).
// Co-runners: the attention kernel, the F16MX GEMM + LayerNorm kernel (64- and 128-row tiles), the F16MX 256 x 256 GEMM.
// Finding: the F16MX GEMM + LayerNorm kernel with 64-row tiles (166 VGPRs: waves of another kernel fit on its SIMDs) computed
// tile-wide 1-ulp differences next to out_ddim_kernel<true> as long as its fp8 correction ran through the SCALED MFMA, which is an
// instruction pair (v_mfma_ld_scale_b32 + v_mfma_scale_f32_32x32x64_f8f6f4) - with both scales 2^0, i.e. the same arithmetic
// (-DJMID_PROBE_SCALED_MFMA rebuilds that: 20 of 20 runs differ); with the unscaled v_mfma_f32_32x32x64_f8f6f4: 0.  The 128-row
// kernel (250 VGPRs, nothing else fits on its SIMDs) never differed, nor did any kernel next to embed_kernel or attention.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -Xclang -target-feature -Xclang -packed-fp32-ops -I safe-interactive-crowdnav_amd/csrc tools/concurrency_probe9.hip -o build/concurrency_probe9
//   ... -DJMID_PROBE_SCALED_MFMA ... -o build/concurrency_probe9_scaled
#include "attn_f16x3.hpp"
#include "gemm_ln_f16x3.hpp"
#include "elementwise.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
using namespace jmid;

int main(int argc, char** argv) {
    const int niter = argc > 1 ? atoi(argv[1]) : 300;
    const int nseq = 32, S = 1200, d = 512, nhead = 4, HD = 128, Spad = vt_spad(S), T = 12, A = 5, K = 20;
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
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    // co-runner 1: attention
    AttnHArgs a{};
    a.Qhi = dev_rand_h(M * d, 0.2f); a.Qlo = dev_rand_h(M * d, 1e-4f); a.Khi = dev_rand_h(M * d, 1.f); a.Klo = dev_rand_h(M * d, 4e-4f);
    a.Vthi = dev_rand_h((size_t)nseq * nhead * HD * Spad, 1.f); a.Vtlo = dev_rand_h((size_t)nseq * nhead * HD * Spad, 4e-4f);
    a.Ohi = dev_rand_h(blk_plane_elems(M, d), 1.f); a.Olo = dev_rand_h(blk_plane_elems(M, d), 1.f);
    a.S = S; a.Spad = Spad; a.d = d; a.nhead = nhead; a.scale = 1.f; a.nsplit = 1;
    hipMalloc(&a.range_flag, 4); hipMemset(a.range_flag, 0, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f16x3_dma_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_DMA_LDS);
    const int nqt = (S + 127) / 128, nblk = nqt * nhead * nseq;
    auto attn = [&]() { hipLaunchKernelGGL((attn_f16x3_dma_kernel<false, true>), dim3(nblk), dim3(256), ATT_DMA_LDS, s1, a, nqt, 0, (unsigned long long*)nullptr); };
    // co-runner 2 / 3: F16MX GEMM + LayerNorm (K = 512)
    GemmLnArgs gl{};
    gl.Ahi = dev_rand_h(blk_plane_elems(M, 512), 1.f); gl.Alo = gl.Ahi;
    gl.W16hi = dev_rand_h((size_t)512 * 512, 0.05f); gl.W16lo = gl.W16hi;
    gl.W8 = (unsigned char*)dev_rand_h((size_t)512 * 512 / 2, 0.5f);
    gl.bias = dev_rand_f(512, 0.1f); gl.gamma = dev_rand_f(512, 1.f); gl.beta = dev_rand_f(512, 0.1f);
    gl.Xh = dev_rand_h(blk_plane_elems(M, 512), 1.f); gl.Xl = dev_rand_h(blk_plane_elems(M, 512), 1e-4f);
    gl.M = (int)M; gl.K = 512; gl.eps = 1e-5f; gl.range_flag = a.range_flag; gl.x2 = 1;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ln_mx_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GLNX_LDS_BYTES);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ln_mx_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GLNX_LDS_BYTES);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ln_f16x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GLN_LDS_BYTES);
    auto ln64 = [&]() { hipLaunchKernelGGL(gemm_ln_mx_kernel<2>, dim3((int)(M / 64)), dim3(512), GLNX_LDS_BYTES, s1, gl, (int)(M / 64)); };
    auto ln128 = [&]() { hipLaunchKernelGGL(gemm_ln_mx_kernel<4>, dim3((int)(M / 128)), dim3(512), GLNX_LDS_BYTES, s1, gl, (int)(M / 128)); };
    auto ln64x2 = [&]() { hipLaunchKernelGGL(gemm_ln_f16x3_kernel<true>, dim3((int)(M / 64)), dim3(512), GLN_LDS_BYTES, s1, gl, (int)(M / 64)); };
    // victim: embed_kernel as the library launches it
    const int hyp_ld = 1796, EA = nseq * A;
    EmbedArgs e{};
    e.x = dev_rand_f(M * 2, 1.f); e.W1 = dev_rand_f(d * 2, 0.5f); e.b1 = dev_rand_f(d, 0.5f); e.pe = dev_rand_f(24 * d, 1.f);
    e.hyp = dev_rand_f((size_t)EA * hyp_ld, 1.f); e.thyp = dev_rand_f(hyp_ld, 1.f);
    e.X = nullptr;
    hipMalloc(&e.Xh, blk_plane_elems(M, d) * 2); hipMalloc(&e.Xl, blk_plane_elems(M, d) * 2);
    e.M = (int)M; e.d = d; e.hyp_ld = hyp_ld; e.goff = 0; e.boff = d; e.rmap = RowMap{T, A, K * A};
    const size_t pe = blk_plane_elems(M, d);
    std::vector<unsigned short> ref(pe * 2), cur(pe * 2);
    const int eblocks = (int)std::min<long>(((long)M * (d / 4) + 255) / 256, 256L * 16);
    auto victim = [&]() { hipLaunchKernelGGL(embed_kernel, dim3(eblocks), dim3(256), 0, s2, e); };
    auto fetch = [&](std::vector<unsigned short>& v) { hipMemcpy(v.data(), e.Xh, pe * 2, hipMemcpyDeviceToHost); hipMemcpy(v.data() + pe, e.Xl, pe * 2, hipMemcpyDeviceToHost); };
    auto run = [&](const char* name, auto co) {
        victim(); hipDeviceSynchronize(); fetch(ref);
        int bad = 0; size_t nel = 0, not_q3 = 0;
        for (int it = 0; it < niter; ++it) {
            hipMemsetAsync(e.Xh, 0xff, pe * 2, s2); hipMemsetAsync(e.Xl, 0xff, pe * 2, s2); hipDeviceSynchronize();
            co(); victim(); co();
            hipDeviceSynchronize();
            fetch(cur);
            size_t dd = 0;
            for (size_t i = 0; i < pe * 2; ++i)
                if (cur[i] != ref[i]) { ++dd; }
            if (dd) { ++bad; nel += dd; }
        }
        printf("%-40s %3d / %d runs differ, %zu halfs\n", name, bad, niter, nel);
        fflush(stdout);
    };
    // the other way round: the GEMM + LayerNorm kernels as victims (they rewrite X in place: restore it before every run)
    half_t *x0h, *x0l; hipMalloc(&x0h, pe * 2); hipMalloc(&x0l, pe * 2);
    hipMemcpy(x0h, gl.Xh, pe * 2, hipMemcpyDeviceToDevice); hipMemcpy(x0l, gl.Xl, pe * 2, hipMemcpyDeviceToDevice);
    std::vector<unsigned short> ref2(pe * 2), cur2(pe * 2);
    auto fetch2 = [&](std::vector<unsigned short>& v) { hipMemcpy(v.data(), gl.Xh, pe * 2, hipMemcpyDeviceToHost); hipMemcpy(v.data() + pe, gl.Xl, pe * 2, hipMemcpyDeviceToHost); };
    auto restore = [&]() { hipMemcpyAsync(gl.Xh, x0h, pe * 2, hipMemcpyDeviceToDevice, s1); hipMemcpyAsync(gl.Xl, x0l, pe * 2, hipMemcpyDeviceToDevice, s1); hipDeviceSynchronize(); };
    auto run2 = [&](const char* name, auto vict, auto co) {
        restore(); vict(); hipDeviceSynchronize(); fetch2(ref2);
        int bad = 0; size_t nel = 0;
        for (int it = 0; it < niter; ++it) {
            restore();
            co(); vict(); co();
            hipDeviceSynchronize();
            fetch2(cur2);
            size_t dd = 0;
            for (size_t i = 0; i < pe * 2; ++i) dd += cur2[i] != ref2[i];
            if (dd && bad == 0) {       // where?  plane index -> (row, col) of the blocked layout: [rb][kt][128 rows][32], 16-byte chunks swizzled
                size_t rows_bad[8] = {0}, shown = 0; std::vector<int> per_tile(M / 64, 0);
                for (size_t i = 0; i < pe; ++i)
                    if (cur2[i] != ref2[i]) {
                        const size_t img = i / 4096, r = (i % 4096) / 32; const size_t rb = img / 16;
                        const int row = (int)(rb * 128 + r);
                        per_tile[row / 64]++;
                        if (shown++ < 6) printf("      hi plane idx %zu: row %d (tile %d, row-in-tile %d), k-image %zu, got %04x want %04x\n", i, row, row / 64, row % 64, img % 16, cur2[i], ref2[i]);
                    }
                int nt = 0; for (int v : per_tile) nt += v > 0;
                printf("      tiles with differences: %d of %zu; first few: ", nt, per_tile.size());
                int c = 0; for (size_t t = 0; t < per_tile.size() && c < 24; ++t) if (per_tile[t]) { printf("%zu(%d) ", t, per_tile[t]); ++c; }
                printf("\n");
            }
            if (dd) { ++bad; nel += dd; }
        }
        printf("%-70s %3d / %d runs differ, %zu halfs\n", name, bad, niter, nel);
        fflush(stdout);
    };
    auto attn2 = [&]() { hipLaunchKernelGGL((attn_f16x3_dma_kernel<false, true>), dim3(nblk), dim3(256), ATT_DMA_LDS, s2, a, nqt, 0, (unsigned long long*)nullptr); };

    // out_ddim_kernel<true> (output layer + DDIM update + next embedding: the row-wise kernel of every step of a chunk)
    OutArgs oa{};
    oa.Y4 = dev_rand_f(M * 128, 1.f); oa.Wo = dev_rand_f(2 * 128, 0.2f); oa.bo = dev_rand_f(2, 0.1f); oa.hyp = e.hyp; oa.thyp = e.thyp;
    float* xs = dev_rand_f(M * 2, 1.f); float* xw; hipMalloc(&xw, M * 2 * 4);
    oa.x = xw; oa.e_out = nullptr; oa.M = (int)M; oa.dl = 128; oa.hyp_ld = hyp_ld; oa.goff = 1024; oa.boff = 1026;
    oa.c_e = 0.3f; oa.c_x = 0.9f; oa.n_x = 1.01f; oa.n_e = -0.2f; oa.rmap = e.rmap;
    EmbedArgs en = e; en.x = xw;
    auto outk = [&]() { hipLaunchKernelGGL(out_ddim_kernel<true>, dim3((int)((M + 3) / 4)), dim3(256), 0, s2, oa, en); };
    std::vector<unsigned short> ref3(pe * 2), cur3(pe * 2); std::vector<float> rx(M * 2), cx(M * 2);
    auto run3 = [&](const char* name, auto co) {
        hipMemcpy(xw, xs, M * 2 * 4, hipMemcpyDeviceToDevice); outk(); hipDeviceSynchronize(); fetch(ref3); hipMemcpy(rx.data(), xw, M * 2 * 4, hipMemcpyDeviceToHost);
        int bad = 0; size_t nel = 0, nx = 0, not_q3 = 0;
        for (int it = 0; it < niter; ++it) {
            hipMemcpy(xw, xs, M * 2 * 4, hipMemcpyDeviceToDevice);
            hipMemsetAsync(e.Xh, 0xff, pe * 2, s2); hipMemsetAsync(e.Xl, 0xff, pe * 2, s2); hipDeviceSynchronize();
            co(); outk(); co();
            hipDeviceSynchronize();
            fetch(cur3); hipMemcpy(cx.data(), xw, M * 2 * 4, hipMemcpyDeviceToHost);
            size_t dd = 0;
            for (size_t i = 0; i < pe * 2; ++i) dd += cur3[i] != ref3[i];
            size_t dx = 0;
            for (size_t i = 0; i < M * 2; ++i) dx += memcmp(&cx[i], &rx[i], 4) != 0;
            if (dd || dx) { ++bad; nel += dd; nx += dx; }
        }
        printf("victim out_ddim<true>, co-runner %-40s %3d / %d runs differ, %zu plane halfs, %zu x values\n", name, bad, niter, nel, nx);
        fflush(stdout);
    };

    run2("victim F16MX LN 64-row, co-runner out_ddim<true>", ln64, outk);
    run2("victim F16MX LN 128-row, co-runner out_ddim<true>", ln128, outk);
    run2("victim F16X2 LN 64-row, co-runner out_ddim<true>", ln64x2, outk);
    run2("victim F16MX LN 64-row, co-runner embed_kernel", ln64, victim);
    run2("victim F16MX LN 64-row, co-runner attention", ln64, attn2);
    return 0;
    run("no co-runner", [&]() {});
    run("attention", attn);
    run("F16MX GEMM + LayerNorm, 64-row tiles", ln64);
    run("F16MX GEMM + LayerNorm, 128-row tiles", ln128);
    run("F16X2 GEMM + LayerNorm, 64-row tiles", ln64x2);
    return 0;
}
