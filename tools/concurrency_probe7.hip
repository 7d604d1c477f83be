// Follow-up to concurrency_probe6: every disturbed float of embed_kernel is component 2 of a 4-wide result in lanes 48-63, whatever
// the arithmetic - i.e. what arrives wrong could be the THIRD DWORD of a 16-byte table load of the last 16-lane group.
// Victims that do nothing but sum cache-resident table loads of one width (embed_kernel's access pattern: every wave reads the same
// few KB of tables, 16 contiguous bytes per lane), next to the F16X2 attention kernel:
//   width 16: global_load_dwordx4    width 8: global_load_dwordx2    width 4: global_load_dword
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -I safe-interactive-crowdnav_amd/csrc tools/concurrency_probe7.hip -o build/concurrency_probe7
#include "attn_f16x3.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace jmid;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int W>
__global__ __launch_bounds__(256) void victim(const float* tab, float* y, int M, int d, int ntab) {
    const int d4 = d >> 2;
    const long total = (long)M * d4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / d4), j = (int)(idx % d4) * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < ntab; ++k) {
            const float* p = tab + (size_t)k * 24 * d + (size_t)((m + 5 * k) % 24) * d + j;     // 24 rows of d floats per table
            if (W == 16) {
                acc += *reinterpret_cast<const f32x4*>(p);
            } else if (W == 8) {
                const f32x2 a = *reinterpret_cast<const volatile f32x2*>(p), b = *reinterpret_cast<const volatile f32x2*>(p + 2);
                acc += f32x4{a[0], a[1], b[0], b[1]};
            } else {
                const volatile float* q = p;
                acc += f32x4{q[0], q[1], q[2], q[3]};
            }
        }
        *reinterpret_cast<f32x4*>(y + (size_t)m * d + j) = acc;
    }
}

int main(int argc, char** argv) {
    const int niter = argc > 1 ? atoi(argv[1]) : 2000;
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
    const int nqt = (S + 127) / 128, nblk = nqt * nhead * nseq;
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    auto attn = [&]() { hipLaunchKernelGGL((attn_f16x3_dma_kernel<false, true>), dim3(nblk), dim3(256), ATT_DMA_LDS, s1, a, nqt, 0, (unsigned long long*)nullptr); };
    const int ntab = 6;
    std::vector<float> ht((size_t)ntab * 24 * d);
    for (auto& v : ht) v = ((rand() & 1023) - 512) / 512.0f;
    float* tab; hipMalloc(&tab, ht.size() * 4); hipMemcpy(tab, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
    float* y; hipMalloc(&y, M * d * 4);
    const long total = (long)M * (d / 4);
    const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
    std::vector<float> ref(M * d), cur(M * d);
    auto run = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize(); hipMemcpy(ref.data(), y, M * d * 4, hipMemcpyDeviceToHost);
        int bad = 0; size_t nel = 0, hist[4] = {0, 0, 0, 0}, quad[4] = {0, 0, 0, 0};
        for (int it = 0; it < niter; ++it) {
            hipMemsetAsync(y, 0xff, M * d * 4, s2); hipDeviceSynchronize();
            attn(); launch(); attn();
            hipDeviceSynchronize();
            hipMemcpy(cur.data(), y, M * d * 4, hipMemcpyDeviceToHost);
            size_t dd = 0;
            for (size_t i = 0; i < M * d; ++i)
                if (memcmp(&cur[i], &ref[i], 4)) { ++dd; ++hist[i % 4]; ++quad[((i / 4) % 64) / 16]; }
            if (dd) { ++bad; nel += dd; }
        }
        printf("%-40s %3d / %d runs differ, %zu floats; by component %zu %zu %zu %zu; by 16-lane group %zu %zu %zu %zu\n", name, bad, niter, nel,
               hist[0], hist[1], hist[2], hist[3], quad[0], quad[1], quad[2], quad[3]);
        fflush(stdout);
    };
    run("16-byte table loads (dwordx4)", [&]() { hipLaunchKernelGGL(victim<16>, dim3(blocks), dim3(256), 0, s2, tab, y, (int)M, d, ntab); });
    run("8-byte table loads (dwordx2)", [&]() { hipLaunchKernelGGL(victim<8>, dim3(blocks), dim3(256), 0, s2, tab, y, (int)M, d, ntab); });
    run("4-byte table loads (dword)", [&]() { hipLaunchKernelGGL(victim<4>, dim3(blocks), dim3(256), 0, s2, tab, y, (int)M, d, ntab); });
    run("16-byte table loads (dwordx4), again", [&]() { hipLaunchKernelGGL(victim<16>, dim3(blocks), dim3(256), 0, s2, tab, y, (int)M, d, ntab); });
    return 0;
}
