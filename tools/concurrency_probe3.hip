// Bisect of embed_kernel as the victim of the attention kernel's QK^T section (see concurrency_probe.hip).
#include "attn_f16x3.hpp"
#include "elementwise.hpp"
#include "gemm_f16x3.hpp"
static jmid::GemmHArgs G; static int GKIND = 1;
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace jmid;
// FLAGS: 1 = no sigmoid, 2 = fixed hyp row (no ea()), 4 = no pe, 8 = fp32 output only (no split), 16 = no W1/x part, 32 = rcp sigmoid
template <int FLAGS>
__global__ __launch_bounds__(256) void embed_var(EmbedArgs a, float* out32) {
    const int d4 = a.d >> 2;
    const long total = (long)a.M * d4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / d4), j = (int)(idx % d4) * 4;
        const float x0 = a.x[2 * (size_t)m], x1 = a.x[2 * (size_t)m + 1];
        const int t = m % a.rmap.T;
        const float* hrow = a.hyp + ((FLAGS & 2) ? 0 : (size_t)a.rmap.ea(m) * a.hyp_ld);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = j + e;
            const float lin = (FLAGS & 16) ? 1.0f : a.W1[2 * c] * x0 + a.W1[2 * c + 1] * x1 + a.b1[c];
            const float ga = hrow[a.goff + c] + a.thyp[a.goff + c];
            const float gate = (FLAGS & 1) ? ga : (FLAGS & 32) ? __builtin_amdgcn_rcpf(1.0f + expf(-ga)) : sigmoidf_(ga);
            const float bias = hrow[a.boff + c] + a.thyp[a.boff + c];
            o[e] = lin * gate + bias + ((FLAGS & 4) ? 0.f : a.pe[(size_t)t * a.d + c]);
        }
        if (FLAGS & 8) {
            *reinterpret_cast<f32x4*>(out32 + (size_t)m * a.d + j) = o;
        } else {
            f16x4 vh, vl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                half_t hh, ll;
                split_f32(o[e], hh, ll);
                vh[e] = hh;
                vl[e] = ll;
            }
            const size_t ob = blk_index(m, j, a.d);
            *reinterpret_cast<f16x4*>(a.Xh + ob) = vh;
            *reinterpret_cast<f16x4*>(a.Xl + ob) = vl;
        }
    }
}
template <int FLAGS>
void run(AttnHArgs a, int nqt, int nblk, EmbedArgs e, float* out32, size_t M, int d, hipStream_t s1, hipStream_t s2, int niter) {
    const size_t pe = blk_plane_elems(M, d);
    const size_t nbytes = (FLAGS & 8) ? M * d * 4 : pe * 2;
    void* buf = (FLAGS & 8) ? (void*)out32 : (void*)e.Xh;
    std::vector<char> ref(nbytes), cur(nbytes);
    const long total = (long)M * (d / 4);
    const int eblocks = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(embed_var<FLAGS>, dim3(eblocks), dim3(256), 0, s2, e, out32); hipDeviceSynchronize();
    hipMemcpy(ref.data(), buf, nbytes, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int it = 0; it < niter; ++it) {
        hipMemsetAsync(buf, 0xff, nbytes, s2); hipDeviceSynchronize();
        hipLaunchKernelGGL((attn_f16x3_dma_kernel<false, true>), dim3(nblk), dim3(256), ATT_DMA_LDS, s1, a, nqt, 0, (unsigned long long*)nullptr);
        hipLaunchKernelGGL(embed_var<FLAGS>, dim3(eblocks), dim3(256), 0, s2, e, out32);
        if (GKIND == 1) (void)launch_gemm_h_dma256<EPI_BIAS, OUT_F32>(G, s2);
        else if (GKIND == 2) (void)launch_gemm_h_cfg<2, 2, EPI_BIAS, OUT_F32>(G, s2);
        else if (GKIND == 3) (void)launch_gemm_h_dma256x256<EPI_BIAS, OUT_F32>(G, s2);
        hipLaunchKernelGGL((attn_f16x3_dma_kernel<false, true>), dim3(nblk), dim3(256), ATT_DMA_LDS, s1, a, nqt, 0, (unsigned long long*)nullptr);
        hipDeviceSynchronize();
        hipMemcpy(cur.data(), buf, nbytes, hipMemcpyDeviceToHost);
        bad += memcmp(cur.data(), ref.data(), nbytes) != 0;
    }
    printf("embed variant flags %2d: %d / %d concurrent runs differ\n", FLAGS, bad, niter);
}
int main(int argc, char** argv) {
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
    const int hyp_ld = 1796, EA = nseq * A;
    EmbedArgs e{};
    e.x = dev_rand_f(M * 2, 1.f); e.W1 = dev_rand_f(d * 2, 0.5f); e.b1 = dev_rand_f(d, 0.5f); e.pe = dev_rand_f(24 * d, 1.f);
    e.hyp = dev_rand_f((size_t)EA * hyp_ld, 1.f); e.thyp = dev_rand_f(hyp_ld, 1.f);
    e.X = nullptr; e.M = (int)M; e.d = d; e.hyp_ld = hyp_ld; e.goff = 0; e.boff = d; e.rmap = RowMap{T, A, K * A};
    hipMalloc(&e.Xh, blk_plane_elems(M, d) * 2); hipMalloc(&e.Xl, blk_plane_elems(M, d) * 2);
    float* out32; hipMalloc(&out32, M * d * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f16x3_dma_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_DMA_LDS);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    const int nqt = (S + 127) / 128, nblk = nqt * nhead * nseq, niter = argc > 1 ? atoi(argv[1]) : 300;
    GKIND = argc > 2 ? atoi(argv[2]) : 1;
    G.Ahi = e.Xh; G.Alo = e.Xl; G.M = (int)M; G.N = 1536; G.K = d;
    G.Whi = dev_rand_h(blk_plane_elems(1536, d), 8.f); G.Wlo = dev_rand_h(blk_plane_elems(1536, d), 4e-3f);
    G.bias = dev_rand_f(1536, 1.f); G.ldc = 1536; G.range_flag = a.range_flag;
    hipMalloc(&G.C, M * 1536 * 4);
    printf("consumer GEMM kind %d (0 none, 1 dma256, 2 register-staged 128x128, 3 dma256x256)\n", GKIND);
    run<0>(a, nqt, nblk, e, out32, M, d, s1, s2, niter);
    run<1 | 2 | 4 | 16>(a, nqt, nblk, e, out32, M, d, s1, s2, niter);
    run<1>(a, nqt, nblk, e, out32, M, d, s1, s2, niter);     // no sigmoid
    run<2>(a, nqt, nblk, e, out32, M, d, s1, s2, niter);     // fixed hyp row (no integer divisions for the row map)
    run<4>(a, nqt, nblk, e, out32, M, d, s1, s2, niter);     // no positional table
    run<8>(a, nqt, nblk, e, out32, M, d, s1, s2, niter);     // fp32 output, no split
    run<16>(a, nqt, nblk, e, out32, M, d, s1, s2, niter);    // no W1 / x part
    run<1 | 2>(a, nqt, nblk, e, out32, M, d, s1, s2, niter);
    return 0;
}
