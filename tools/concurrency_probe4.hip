// Where does the wrong value of a bystander kernel come from?  Victims next to the attention kernel (two streams):
//   copy     y[i] = x[i] with x[i] = i (16-byte loads and stores): a wrong word names the index that arrived instead
//   compute  y[i] = f(i), no loads at all (integer + float VALU chain on the thread's own index)
//   tables   y[i] = sum of eight 16-byte loads from small tables that stay in the vector L1
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -I safe-interactive-crowdnav_amd/csrc tools/concurrency_probe4.hip -o build/concurrency_probe4
#include "attn_f16x3.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace jmid;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void victim_copy(const u32x4* x, u32x4* y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i];
}
// eight small tables that stay in the CU's vector L1 (as embed_kernel's bias / gate / positional tables do)
__global__ __launch_bounds__(256) void victim_tables(const u32x4* x, u32x4* y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        u32x4 acc = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += x[k * 1024 + ((i + 37 * k) & 127)];
        y[i] = acc;
    }
}
__global__ __launch_bounds__(256) void victim_compute(unsigned* y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u;
        float f = (float)(h >> 8) * (1.0f / 16777216.0f);
#pragma unroll
        for (int k = 0; k < 24; ++k) { f = fmaf(f, 0.999f, 0.25f); h = (h ^ (h >> 13)) * 0x5bd1e995u; }
        y[i] = h ^ __float_as_uint(f);
    }
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
    std::vector<unsigned> hx(n), ref(n), cur(n);
    for (size_t i = 0; i < n; ++i) hx[i] = (unsigned)i;
    unsigned *x, *y; hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    const int nqt = (S + 127) / 128, nblk = nqt * nhead * nseq, niter = argc > 1 ? atoi(argv[1]) : 400;
    for (int kind = 0; kind < 3; ++kind) {
        auto launch = [&]() {
            if (kind == 0) hipLaunchKernelGGL(victim_copy, dim3(2048), dim3(256), 0, s2, (const u32x4*)x, (u32x4*)y, n / 4);
            else if (kind == 1) hipLaunchKernelGGL(victim_compute, dim3(2048), dim3(256), 0, s2, y, n);
            else hipLaunchKernelGGL(victim_tables, dim3(2048), dim3(256), 0, s2, (const u32x4*)x, (u32x4*)y, n / 4);
        };
        launch(); hipDeviceSynchronize(); hipMemcpy(ref.data(), y, n * 4, hipMemcpyDeviceToHost);
        int bad = 0, shown = 0;
        for (int it = 0; it < niter; ++it) {
            hipMemsetAsync(y, 0xee, n * 4, s2); hipDeviceSynchronize();
            hipLaunchKernelGGL((attn_f16x3_dma_kernel<false, true>), dim3(nblk), dim3(256), ATT_DMA_LDS, s1, a, nqt, 0, (unsigned long long*)nullptr);
            launch();
            hipLaunchKernelGGL((attn_f16x3_dma_kernel<false, true>), dim3(nblk), dim3(256), ATT_DMA_LDS, s1, a, nqt, 0, (unsigned long long*)nullptr);
            hipDeviceSynchronize();
            hipMemcpy(cur.data(), y, n * 4, hipMemcpyDeviceToHost);
            size_t dcount = 0;
            for (size_t i = 0; i < n; ++i)
                if (cur[i] != ref[i]) {
                    if (shown < 40) { printf("  %s iter %d: y[%zu] = 0x%08x (expected 0x%08x)\n", kind == 0 ? "copy" : kind == 1 ? "compute" : "tables", it, i, cur[i], ref[i]); ++shown; }
                    ++dcount;
                }
            bad += dcount != 0;
        }
        printf("victim %s: %d / %d concurrent runs differ\n", kind == 0 ? "copy (x[i] = i)" : kind == 1 ? "compute (no loads)" : "tables (eight L1-resident 16-byte loads per thread)", bad, niter);
    }
    return 0;
}
