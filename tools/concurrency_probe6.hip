// Source-level bisect of embed_kernel as the victim of the attention kernel (docs/NOTEBOOK.md section 3): the disturbance is always
// ONE component (e = 2) of the 4-wide result in lanes 48-63 of a wave holding a plausible stale-looking value - a VALU result
// not written for the last 16-lane group.  Which construct of the kernel makes it vulnerable?
//   bit 0  32-bit index arithmetic (no 64-bit division: no EXEC-masked slow path in the loop head)
//   bit 1  sigmoid as rcp(1 + exp2(-x log2 e)) (no range clamps: no v_cmp / v_cndmask through VCC, no v_div_fmas)
//   bit 2  one element per thread iteration instead of a grid-stride loop
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -I safe-interactive-crowdnav_amd/csrc tools/concurrency_probe6.hip -o build/concurrency_probe6
#include "attn_f16x3.hpp"
#include "elementwise.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace jmid;

template <int FLAGS>
__device__ __forceinline__ float sig(float x) {
    if (FLAGS & 2) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
    return sigmoidf_(x);
}
template <int FLAGS>
__device__ __forceinline__ void store_var(const EmbedArgs& a, int m, int j, int t, float x0, float x1, const float* hrow) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = j + e;
        const float lin = a.W1[2 * c] * x0 + a.W1[2 * c + 1] * x1 + a.b1[c];
        const float gate = sig<FLAGS>(hrow[a.goff + c] + a.thyp[a.goff + c]);
        const float bias = hrow[a.boff + c] + a.thyp[a.boff + c];
        o[e] = lin * gate + bias + a.pe[(size_t)t * a.d + c];
    }
    *reinterpret_cast<f32x4*>(a.X + (size_t)m * a.d + j) = o;
}
template <int FLAGS>
__global__ __launch_bounds__(256) void embed_var(EmbedArgs a) {
    const int d4 = a.d >> 2;
    if (FLAGS & 4) {
        const int idx = blockIdx.x * blockDim.x + threadIdx.x;
        if (idx >= a.M * d4) return;
        const int m = idx / d4, j = (idx % d4) * 4;
        store_var<FLAGS>(a, m, j, m % a.rmap.T, a.x[2 * (size_t)m], a.x[2 * (size_t)m + 1], a.hyp + (size_t)a.rmap.ea(m) * a.hyp_ld);
    } else if (FLAGS & 1) {
        const int total = a.M * d4;
        for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
            const int m = idx / d4, j = (idx % d4) * 4;
            store_var<FLAGS>(a, m, j, m % a.rmap.T, a.x[2 * (size_t)m], a.x[2 * (size_t)m + 1], a.hyp + (size_t)a.rmap.ea(m) * a.hyp_ld);
        }
    } else {
        const long total = (long)a.M * d4;
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
            const int m = (int)(idx / d4), j = (int)(idx % d4) * 4;
            store_var<FLAGS>(a, m, j, m % a.rmap.T, a.x[2 * (size_t)m], a.x[2 * (size_t)m + 1], a.hyp + (size_t)a.rmap.ea(m) * a.hyp_ld);
        }
    }
}

int main(int argc, char** argv) {
    const int niter = argc > 1 ? atoi(argv[1]) : 1500;
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
    const int hyp_ld = 1796, EA = nseq * A;
    EmbedArgs e{};
    e.x = dev_rand_f(M * 2, 1.f); e.W1 = dev_rand_f(d * 2, 0.5f); e.b1 = dev_rand_f(d, 0.5f); e.pe = dev_rand_f(24 * d, 1.f);
    e.hyp = dev_rand_f((size_t)EA * hyp_ld, 1.f); e.thyp = dev_rand_f(hyp_ld, 1.f);
    hipMalloc(&e.X, M * d * 4);
    e.M = (int)M; e.d = d; e.hyp_ld = hyp_ld; e.goff = 0; e.boff = d; e.rmap = RowMap{T, A, K * A};
    const long total = (long)M * (d / 4);
    const int eblocks = (int)std::min<long>((total + 255) / 256, 4096), fblocks = (int)((total + 255) / 256);
    std::vector<float> ref(M * d), cur(M * d);
    auto run = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize(); hipMemcpy(ref.data(), e.X, M * d * 4, hipMemcpyDeviceToHost);
        int bad = 0; size_t nel = 0, not_e2 = 0, not_q3 = 0;
        for (int it = 0; it < niter; ++it) {
            hipMemsetAsync(e.X, 0xff, M * d * 4, s2); hipDeviceSynchronize();
            attn(); launch(); attn();
            hipDeviceSynchronize();
            hipMemcpy(cur.data(), e.X, M * d * 4, hipMemcpyDeviceToHost);
            size_t dd = 0;
            for (size_t i = 0; i < M * d; ++i)
                if (memcmp(&cur[i], &ref[i], 4)) { ++dd; not_e2 += (i % 4) != 2; not_q3 += ((i / 4) % 64) < 48; }
            if (dd) { ++bad; nel += dd; }
        }
        printf("%-62s %3d / %d runs differ, %zu floats (%zu not component 2, %zu not in lanes 48-63)\n", name, bad, niter, nel, not_e2, not_q3);
        fflush(stdout);
    };
    run("0: embed_kernel as shipped (64-bit loop, IEEE sigmoid)", [&]() { hipLaunchKernelGGL(embed_var<0>, dim3(eblocks), dim3(256), 0, s2, e); });
    run("1: 32-bit index arithmetic", [&]() { hipLaunchKernelGGL(embed_var<1>, dim3(eblocks), dim3(256), 0, s2, e); });
    run("2: rcp / exp2 sigmoid (no VCC clamps, no v_div_fmas)", [&]() { hipLaunchKernelGGL(embed_var<2>, dim3(eblocks), dim3(256), 0, s2, e); });
    run("3: both", [&]() { hipLaunchKernelGGL(embed_var<3>, dim3(eblocks), dim3(256), 0, s2, e); });
    run("4: one element per thread (no loop), IEEE sigmoid", [&]() { hipLaunchKernelGGL(embed_var<4>, dim3(fblocks), dim3(256), 0, s2, e); });
    run("6: one element per thread, rcp / exp2 sigmoid", [&]() { hipLaunchKernelGGL(embed_var<6>, dim3(fblocks), dim3(256), 0, s2, e); });
    run("0: embed_kernel as shipped, again", [&]() { hipLaunchKernelGGL(embed_var<0>, dim3(eblocks), dim3(256), 0, s2, e); });
    return 0;
}
