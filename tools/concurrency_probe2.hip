// Which instruction class of a victim kernel goes wrong next to the attention kernel's QK^T section?
#include "attn_f16x3.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace jmid;
template <int KIND>
__global__ __launch_bounds__(256) void victim(const float* x, float* y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        float r;
        if (KIND == 0) r = 1.0f / (1.0f + expf(-v));                               // IEEE division + expf
        else if (KIND == 1) r = __builtin_amdgcn_rcpf(1.0f + expf(-v));               // v_rcp_f32 + expf
        else if (KIND == 2) r = 1.0f / (1.0f + v * v);                               // IEEE division only
        else if (KIND == 3) r = expf(-v);                                            // expf only
        else { r = v; for (int k = 0; k < 16; ++k) r = fmaf(r, 0.999f, 0.001f * v); }   // plain FMAs
        y[i] = r;
    }
}
template <int KIND>
int run(AttnHArgs a, int nqt, int nblk, const float* x, float* y, size_t n, hipStream_t s1, hipStream_t s2, int niter) {
    std::vector<float> ref(n), cur(n);
    hipLaunchKernelGGL(victim<KIND>, dim3(4096), dim3(256), 0, s2, x, y, n); hipDeviceSynchronize();
    hipMemcpy(ref.data(), y, n * 4, hipMemcpyDeviceToHost);
    int bad = 0; size_t nel = 0;
    for (int it = 0; it < niter; ++it) {
        hipMemsetAsync(y, 0xff, n * 4, s2); hipDeviceSynchronize();
        hipLaunchKernelGGL((attn_f16x3_dma_kernel<false, true>), dim3(nblk), dim3(256), ATT_DMA_LDS, s1, a, nqt, 0, (unsigned long long*)nullptr);
        hipLaunchKernelGGL(victim<KIND>, dim3(4096), dim3(256), 0, s2, x, y, n);
        hipLaunchKernelGGL((attn_f16x3_dma_kernel<false, true>), dim3(nblk), dim3(256), ATT_DMA_LDS, s1, a, nqt, 0, (unsigned long long*)nullptr);
        hipDeviceSynchronize();
        hipMemcpy(cur.data(), y, n * 4, hipMemcpyDeviceToHost);
        size_t d = 0; for (size_t i = 0; i < n; ++i) d += memcmp(&cur[i], &ref[i], 4) != 0;
        if (d) { ++bad; nel += d; }
    }
    printf("victim kind %d: %d / %d concurrent runs differ (%zu elements in total)\n", KIND, bad, niter, nel);
    return bad;
}
int main(int argc, char** argv) {
    const int nseq = 8, S = 1200, d = 512, nhead = 4, HD = 128, Spad = vt_spad(S);
    const size_t M = (size_t)nseq * S;
    auto dev_rand_h = [&](size_t n, float sc) {
        std::vector<_Float16> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = (_Float16)(sc * ((rand() & 1023) - 512) / 512.0f);
        half_t* p; hipMalloc(&p, n * 2); hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice); return p;
    };
    AttnHArgs a{};
    a.Qhi = dev_rand_h(M * d, 0.2f); a.Qlo = dev_rand_h(M * d, 1e-4f); a.Khi = dev_rand_h(M * d, 1.f); a.Klo = dev_rand_h(M * d, 4e-4f);
    a.Vthi = dev_rand_h((size_t)nseq * nhead * HD * Spad, 1.f); a.Vtlo = dev_rand_h((size_t)nseq * nhead * HD * Spad, 4e-4f);
    a.Ohi = dev_rand_h(blk_plane_elems(M, d), 1.f); a.Olo = dev_rand_h(blk_plane_elems(M, d), 1.f);
    a.S = S; a.Spad = Spad; a.d = d; a.nhead = nhead; a.scale = 1.f; a.nsplit = 1;
    hipMalloc(&a.range_flag, 4); hipMemset(a.range_flag, 0, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f16x3_dma_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_DMA_LDS);
    const size_t n = (size_t)4 << 20;
    std::vector<float> hx(n); for (size_t i = 0; i < n; ++i) hx[i] = 4.0f * ((rand() & 65535) - 32768) / 32768.0f;
    float *x, *y; hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    const int nqt = (S + 127) / 128, nblk = nqt * nhead * nseq, niter = argc > 1 ? atoi(argv[1]) : 200;
    run<0>(a, nqt, nblk, x, y, n, s1, s2, niter);
    run<1>(a, nqt, nblk, x, y, n, s1, s2, niter);
    run<2>(a, nqt, nblk, x, y, n, s1, s2, niter);
    run<3>(a, nqt, nblk, x, y, n, s1, s2, niter);
    run<4>(a, nqt, nblk, x, y, n, s1, s2, niter);
    return 0;
}
