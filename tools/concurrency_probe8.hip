// Follow-up to concurrency_probe6 / 7.  In embed_kernel's ISA the disturbed component (e = 2 of the 4-wide result) is the LOW result
// of   v_pk_add_f32 v[20:21], v[20:21], v[22:23] op_sel:[0,1] op_sel_hi:[1,0]   - a packed-fp32 add whose operand selects are
// CROSSED (the low result reads the high register of src1) - fed by v_pk_mul_f32 ... op_sel:[1,0] op_sel_hi:[0,1].  Victims made of
// exactly these instructions (inline asm), next to the F16X2 attention kernel, with the real embed_kernel as the control:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -I safe-interactive-crowdnav_amd/csrc tools/concurrency_probe8.hip -o build/concurrency_probe8
#include "attn_f16x3.hpp"
#include "elementwise.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace jmid;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void victim(const float* x, float* y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
        f32x2 a = {v[0], v[1]}, b = {v[2], v[3]}, c = {v[1], v[2]};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f32x2 t, u;
            if (KIND == 0) {          // crossed selects, as in embed_kernel
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(c), "v"(a));
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(u) : "v"(c), "v"(b));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(a) : "v"(t), "v"(u));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(b) : "v"(u), "v"(t));
            } else {                  // the same chain with straight selects
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(c), "v"(a));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(u) : "v"(c), "v"(b));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a) : "v"(t), "v"(u));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(b) : "v"(u), "v"(t));
            }
            a = a * 0.5f;
            b = b * 0.5f;
        }
        *reinterpret_cast<f32x4*>(y + 4 * i) = f32x4{a[0], a[1], b[0], b[1]};
    }
}

// alternative co-runners: a pure MFMA loop (no LDS, no memory traffic) and a pure VALU loop
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
    const int niter = argc > 1 ? atoi(argv[1]) : 1500;
    const int corun = argc > 2 ? atoi(argv[2]) : 0;      // 0 attention (F16X2 instance), 1 pure MFMA loop, 2 pure VALU loop, 3 nothing
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
    float* sink; hipMalloc(&sink, 2048 * 256 * 4);
    f16x8* fr = reinterpret_cast<f16x8*>(dev_rand_h(2048 * 8, 1.f));
    auto attn = [&]() {
        if (corun == 0) hipLaunchKernelGGL((attn_f16x3_dma_kernel<false, true>), dim3(nblk), dim3(256), ATT_DMA_LDS, s1, a, nqt, 0, (unsigned long long*)nullptr);
        else if (corun == 1) hipLaunchKernelGGL(mfma_loop, dim3(1024), dim3(256), 0, s1, sink, 800, fr);
        else if (corun == 2) hipLaunchKernelGGL(valu_loop, dim3(2048), dim3(256), 0, s1, sink, 20000);
    };
    printf("co-runner: %s\n", corun == 0 ? "attention kernel (F16X2 instance)" : corun == 1 ? "pure MFMA loop" : corun == 2 ? "pure VALU loop" : "none");
    const size_t n4 = M * d / 4;
    float* x = dev_rand_f(4 * n4, 1.f); float* y; hipMalloc(&y, 4 * n4 * 4);
    std::vector<float> ref(4 * n4), cur(4 * n4);
    auto run = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize(); hipMemcpy(ref.data(), y, 4 * n4 * 4, hipMemcpyDeviceToHost);
        int bad = 0; size_t nel = 0, hist[4] = {0, 0, 0, 0}, quad[4] = {0, 0, 0, 0};
        for (int it = 0; it < niter; ++it) {
            hipMemsetAsync(y, 0xff, 4 * n4 * 4, s2); hipDeviceSynchronize();
            attn(); launch(); attn();
            hipDeviceSynchronize();
            hipMemcpy(cur.data(), y, 4 * n4 * 4, hipMemcpyDeviceToHost);
            size_t dd = 0;
            for (size_t i = 0; i < 4 * n4; ++i)
                if (memcmp(&cur[i], &ref[i], 4)) { ++dd; ++hist[i % 4]; ++quad[((i / 4) % 64) / 16]; }
            if (dd) { ++bad; nel += dd; }
        }
        printf("%-46s %3d / %d runs differ, %zu floats; by component %zu %zu %zu %zu; by 16-lane group %zu %zu %zu %zu\n", name, bad, niter, nel,
               hist[0], hist[1], hist[2], hist[3], quad[0], quad[1], quad[2], quad[3]);
        fflush(stdout);
    };
    const int hyp_ld = 1796, EA = nseq * A;
    EmbedArgs e{};
    e.x = dev_rand_f(M * 2, 1.f); e.W1 = dev_rand_f(d * 2, 0.5f); e.b1 = dev_rand_f(d, 0.5f); e.pe = dev_rand_f(24 * d, 1.f);
    e.hyp = dev_rand_f((size_t)EA * hyp_ld, 1.f); e.thyp = dev_rand_f(hyp_ld, 1.f);
    e.X = y; e.M = (int)M; e.d = d; e.hyp_ld = hyp_ld; e.goff = 0; e.boff = d; e.rmap = RowMap{T, A, K * A};
    const int eblocks = (int)std::min<long>(((long)n4 + 255) / 256, 4096);
    run("embed_kernel (control)", [&]() { hipLaunchKernelGGL(embed_kernel, dim3(eblocks), dim3(256), 0, s2, e); });
    run("packed fp32, CROSSED op_sel (asm)", [&]() { hipLaunchKernelGGL(victim<0>, dim3(4096), dim3(256), 0, s2, x, y, n4); });
    run("packed fp32, straight selects (asm)", [&]() { hipLaunchKernelGGL(victim<1>, dim3(4096), dim3(256), 0, s2, x, y, n4); });
    run("embed_kernel (control), again", [&]() { hipLaunchKernelGGL(embed_kernel, dim3(eblocks), dim3(256), 0, s2, e); });
    return 0;
}
