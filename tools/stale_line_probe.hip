// Is the run-to-run variation seen with two chunk lanes (docs/NOTEBOOK.md section 3) a MEMORY-VISIBILITY effect rather than a
// corrupted register?  The round-1 probes saw "one 16-lane group of one register" of a row-wise kernel go wrong next to
// the attention kernel: 16 lanes x 8-byte stores = exactly one 128-byte L2 line, and the errors in the real pipeline
// were 1e-4..1e-2 (a value of the PREVIOUS denoise step looks like that; a corrupted register would not).
//
// Stream A: for it = 1..N:  writer(buf, it)  ->  checker(buf, it, counters)      (producer / consumer, same stream)
// Stream B: a co-runner kept busy the whole time (attention LDS-DMA kernel, or a plain spin kernel)
// writer stores tag `it` with 8-byte stores in the blocked panel layout (as embed_kernel / the LayerNorm epilogues do);
// checker reads 16-byte pieces with a different grid (so the workgroup -> XCD assignment differs from the writer's) and
// counts words that are != it, classifying them as STALE (== it - 1, the previous iteration's line) or OTHER.
//   mode bits: 1 = co-runner is the attention DMA kernel, 2 = co-runner is a spin kernel, 4 = writer ends with a
//   device-scope release fence, 8 = checker starts with a device-scope acquire fence, 16 = checker loads with sc1
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -I safe-interactive-crowdnav_amd/csrc tools/stale_line_probe.hip -o build/stale_line_probe
#include "attn_f16x3.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace jmid;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void writer(unsigned* buf, size_t n2, unsigned tag, int fence) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x)
        *reinterpret_cast<u32x2*>(buf + 2 * i) = u32x2{tag, tag ^ (unsigned)i};
    if (fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}
__global__ __launch_bounds__(256) void checker(const unsigned* buf, size_t n2, unsigned tag, unsigned long long* cnt, int fence, int sc1) {
    if (fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    unsigned long long stale = 0, other = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        u32x2 v;
        if (sc1) {
            const unsigned* p = buf + 2 * i;
            asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        } else {
            v = *reinterpret_cast<const u32x2*>(buf + 2 * i);
        }
        if (v[0] != tag || v[1] != (tag ^ (unsigned)i)) {
            if (v[0] == tag - 1 && v[1] == ((tag - 1) ^ (unsigned)i)) ++stale; else ++other;
        }
    }
    if (stale) atomicAdd(cnt, stale);
    if (other) atomicAdd(cnt + 1, other);
}
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    for (int it = 0; it < iters; ++it) { y = fmaf(y, 1.0001f, x); x = fmaf(x, 0.9999f, 1e-4f); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 1, niter = argc > 2 ? atoi(argv[2]) : 2000;
    const int wgrid = argc > 3 ? atoi(argv[3]) : 3000, cgrid = argc > 4 ? atoi(argv[4]) : 1111;
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
    hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f16x3_dma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_DMA_LDS);
    const int nqt = (S + 127) / 128, nblk = nqt * nhead * nseq;
    const size_t n2 = (size_t)M * d / 4;            // 8-byte elements of one fp16 plane
    unsigned* buf; hipMalloc(&buf, n2 * 8); hipMemset(buf, 0, n2 * 8);
    unsigned long long* cnt; hipMalloc(&cnt, 16); hipMemset(cnt, 0, 16);
    float* sink; hipMalloc(&sink, 4096 * 256 * 4);
    hipStream_t sA, sB; hipStreamCreate(&sA); hipStreamCreate(&sB);
    hipEvent_t done; hipEventCreate(&done);
    for (int it = 1; it <= niter; ++it) {
        if (mode & 1) hipLaunchKernelGGL(attn_f16x3_dma_kernel<false>, dim3(nblk), dim3(256), ATT_DMA_LDS, sB, a, nqt, 0, (unsigned long long*)nullptr);
        if (mode & 2) hipLaunchKernelGGL(spin, dim3(512 + (it * 37) % 512), dim3(256), 0, sB, sink, 3000);
        hipLaunchKernelGGL(writer, dim3(wgrid + (it % 7)), dim3(256), 0, sA, buf, n2, (unsigned)it, (mode & 4) ? 1 : 0);
        hipLaunchKernelGGL(checker, dim3(cgrid + (it % 5)), dim3(256), 0, sA, buf, n2, (unsigned)it, cnt, (mode & 8) ? 1 : 0, (mode & 16) ? 1 : 0);
        if (it % 64 == 0) hipStreamSynchronize(sA);   // keep the queues bounded
    }
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost);
    printf("mode %2d  iters %d  writer grid ~%d  checker grid ~%d :  STALE (previous iteration) 8-byte words %llu   OTHER %llu   of %zu per iteration\n",
           mode, niter, wgrid, cgrid, h[0], h[1], n2);
    return 0;
}
