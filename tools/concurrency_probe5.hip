// Which instruction class of embed_kernel is the one that goes wrong next to the attention kernel (docs/NOTEBOOK.md section 3)?
// Victims built from the classes its ISA holds and the round-1 synthetic victims did not: packed fp32 VALU (v_pk_*_f32),
// integer division by run-time values (RowMap::ea: v_rcp_iflag / v_mul_hi chains), the hi/lo fp16 split, and the real
// embed_kernel as the control.  Co-runner: the F16X2 attention instance (the one with the higher disturbance rate).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -I safe-interactive-crowdnav_amd/csrc tools/concurrency_probe5.hip -o build/concurrency_probe5
#include "attn_f16x3.hpp"
#include "elementwise.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace jmid;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void victim(const float* x, float* y, size_t n4, int T, int A, int KA) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
        f32x4 r;
        if (KIND == 0) {            // packed fp32 chain: v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32
            f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                a = a * f32x2{0.999f, 1.001f} + b;
                b = b * f32x2{1.0005f, 0.9995f} + a * f32x2{0.25f, -0.25f};
                asm volatile("" : "+v"(a), "+v"(b));
            }
            r = f32x4{a[0], a[1], b[0], b[1]};
        } else if (KIND == 1) {     // integer divisions by run-time values, as RowMap::ea and m % T
            const int m = (int)(i % 600000);
            const int row = m / T, e = row / KA, ag = row % A, t = m % T;
            r = f32x4{(float)(e * A + ag), (float)t, v[0] + (float)row, v[1]};
        } else if (KIND == 2) {     // hi / lo fp16 split (split_f32) of four values
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                half_t h, l;
                split_f32(v[e] * 1.37f, h, l);
                r[e] = (float)h + (float)l;
            }
        } else {                    // scalar fp32 chain (round-1 control: never disturbed)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float q = v[e];
                for (int k = 0; k < 12; ++k) q = fmaf(q, 0.999f, 0.001f * v[e]);
                r[e] = q;
            }
        }
        *reinterpret_cast<f32x4*>(y + 4 * i) = r;
    }
}

int main(int argc, char** argv) {
    const int niter = argc > 1 ? atoi(argv[1]) : 400;
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
    hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f16x3_dma_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_DMA_LDS);
    const int nqt = (S + 127) / 128, nblk = nqt * nhead * nseq;
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    auto attn = [&]() { hipLaunchKernelGGL((attn_f16x3_dma_kernel<false, true>), dim3(nblk), dim3(256), ATT_DMA_LDS, s1, a, nqt, 0, (unsigned long long*)nullptr); };
    const size_t n4 = (size_t)1 << 20;
    float* x = dev_rand_f(4 * n4, 2.f); float* y; hipMalloc(&y, (4 * n4 + M * d) * 4);   // embed_kernel writes M * d floats
    std::vector<float> ref(4 * n4), cur(4 * n4);
    auto run = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize(); hipMemcpy(ref.data(), y, 4 * n4 * 4, hipMemcpyDeviceToHost);
        int bad = 0; size_t nel = 0;
        for (int it = 0; it < niter; ++it) {
            hipMemsetAsync(y, 0xff, 4 * n4 * 4, s2); hipDeviceSynchronize();
            attn(); launch(); attn();
            hipDeviceSynchronize();
            hipMemcpy(cur.data(), y, 4 * n4 * 4, hipMemcpyDeviceToHost);
            size_t dd = 0; for (size_t i = 0; i < 4 * n4; ++i) dd += memcmp(&cur[i], &ref[i], 4) != 0;
            if (dd) {
                ++bad; nel += dd;
                if (bad <= 4) {
                    printf("  run %d: %zu differing floats:", it, dd);
                    size_t shown = 0;
                    for (size_t i = 0; i < 4 * n4 && shown < 40; ++i)
                        if (memcmp(&cur[i], &ref[i], 4)) { printf(" [%zu: row %zu col %zu e%zu  %.6g vs %.6g]", i, i / 512, i % 512, i % 4, cur[i], ref[i]); ++shown; }
                    printf("\n");
                }
            }
        }
        printf("%-44s %d / %d concurrent runs differ (%zu elements)\n", name, bad, niter, nel);
    };
    run("packed fp32 VALU chain (v_pk_*_f32)", [&]() { hipLaunchKernelGGL(victim<0>, dim3(4096), dim3(256), 0, s2, x, y, n4, T, A, K * A); });
    run("integer divisions by run-time values", [&]() { hipLaunchKernelGGL(victim<1>, dim3(4096), dim3(256), 0, s2, x, y, n4, T, A, K * A); });
    run("hi / lo fp16 split", [&]() { hipLaunchKernelGGL(victim<2>, dim3(4096), dim3(256), 0, s2, x, y, n4, T, A, K * A); });
    run("scalar fp32 chain (control)", [&]() { hipLaunchKernelGGL(victim<3>, dim3(4096), dim3(256), 0, s2, x, y, n4, T, A, K * A); });
    // the real embed_kernel (control: expected to differ in ~1-4 % of the runs)
    const int hyp_ld = 1796, EA = nseq * A;
    EmbedArgs e{};
    e.x = dev_rand_f(M * 2, 1.f); e.W1 = dev_rand_f(d * 2, 0.5f); e.b1 = dev_rand_f(d, 0.5f); e.pe = dev_rand_f(24 * d, 1.f);
    e.hyp = dev_rand_f((size_t)EA * hyp_ld, 1.f); e.thyp = dev_rand_f(hyp_ld, 1.f);
    e.X = y; e.M = (int)M; e.d = d; e.hyp_ld = hyp_ld; e.goff = 0; e.boff = d; e.rmap = RowMap{T, A, K * A};
    e.Xh = nullptr; e.Xl = nullptr;
    const long total = (long)M * (d / 4);
    const int eblocks = (int)std::min<long>((total + 255) / 256, 4096);
    run("embed_kernel, fp32 output (control)", [&]() { hipLaunchKernelGGL(embed_kernel, dim3(eblocks), dim3(256), 0, s2, e); });
    return 0;
}
