// What sets the ~3.3 us between dependent kernels of a one-scene chain?  Period of a warm chain of EMPTY kernels (the body returns at
// entry) against grid size, threads per workgroup, dynamic LDS, argument bytes, a dirty cache line per workgroup, and the same chain
// replayed as a hipGraph.    hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/launch_gap_probe.hip -o build/launch_gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Big { unsigned w[76]; };      // the small GEMM's argument block is 76 dwords
__global__ void k_empty(int* p) { if (p == (int*)1) *p = 0; }
__global__ void k_empty_big(Big b, int* p) { if (p == (int*)1) *p = b.w[75]; }
__global__ void k_dirty(int* p) { if (threadIdx.x == 0) p[blockIdx.x * 32] = blockIdx.x; }                 // one dirty 128-byte line per workgroup
__global__ void k_dirty_big(float4* p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = float4{1.f, 2.f, 3.f, 4.f}; }
__global__ void k_spin(int* p, int ticks, int store) {       // every wave stays `ticks` x 10 ns; optionally one 16-byte store per thread at the end
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(1);
    if (store) reinterpret_cast<float4*>(p)[blockIdx.x * blockDim.x + threadIdx.x] = float4{1.f, 2.f, 3.f, 4.f};
}
static float chain_us(const std::function<void(hipStream_t)>& launch, int reps, hipStream_t st) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < reps; ++i) launch(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) launch(st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
#include <functional>
int main() {
    hipStream_t st; hipStreamCreate(&st);
    int* buf; hipMalloc(&buf, 64 << 20); hipMemset(buf, 0, 64 << 20);
    const int reps = 2000;
    hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("chain of empty kernels, us per launch:\n");
    for (int grid : {1, 64, 240, 1024, 4096})
        for (int thr : {64, 256, 512})
            for (int lds : {0, 48 * 1024, 144 * 1024}) {
                if (grid > 240 && (thr != 256 || lds == 48 * 1024)) continue;
                const float us = chain_us([&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(thr), lds, s, buf); }, reps, st);
                printf("  grid %5d x %3d threads, %3d KB LDS: %.2f\n", grid, thr, lds / 1024, us);
            }
    Big b{};
    printf("  grid 240 x 256, 304-byte argument block: %.2f\n", chain_us([&](hipStream_t s) { hipLaunchKernelGGL(k_empty_big, dim3(240), dim3(256), 0, s, b, buf); }, reps, st));
    printf("  grid 240 x 256, one dirty line per workgroup: %.2f\n", chain_us([&](hipStream_t s) { hipLaunchKernelGGL(k_dirty, dim3(240), dim3(256), 0, s, buf); }, reps, st));
    for (int mb : {1, 4, 16}) {
        const int n = mb * (1 << 20) / 16;
        printf("  grid 240 x 256 writing %2d MB: %.2f\n", mb, chain_us([&](hipStream_t s) { hipLaunchKernelGGL(k_dirty_big, dim3(240), dim3(256), 0, s, (float4*)buf, n); }, reps / 4, st));
    }
    // alternating two different kernels (instruction cache)
    printf("  alternating two kernels (240 x 256): %.2f\n", chain_us([&](hipStream_t s) {
               hipLaunchKernelGGL(k_empty, dim3(240), dim3(256), 0, s, buf);
               hipLaunchKernelGGL(k_dirty, dim3(240), dim3(256), 0, s, buf); }, reps, st) / 2);
    // the same chain as a graph of 50 nodes
    {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_empty, dim3(240), dim3(256), 0, st, buf);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        printf("  hipGraph of 50 empty kernels (240 x 256): %.2f per node\n", chain_us([&](hipStream_t s) { hipGraphLaunch(ge, s); }, 100, st) / 50);
    }
    // kernels that LAST 5 us (the host is then ahead of the queue): period - 5 us = what the GPU spends between two dependent kernels,
    // in a stream and in a graph, with and without 1 MB of stores to flush
    hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int lds : {48 * 1024, 144 * 1024})
        for (int thr : {256, 512})
            printf("  5 us kernels (240 x %d, %d KB LDS): period %.2f in a stream\n", thr, lds / 1024,
                   chain_us([&](hipStream_t s) { hipLaunchKernelGGL(k_spin, dim3(240), dim3(thr), lds, s, buf, 500, 0); }, reps / 4, st));
    for (int store : {0, 1}) {
        const float us = chain_us([&](hipStream_t s) { hipLaunchKernelGGL(k_spin, dim3(240), dim3(256), 0, s, buf, 500, store); }, reps / 4, st);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_spin, dim3(240), dim3(256), 0, st, buf, 500, store);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        const float ug = chain_us([&](hipStream_t s) { hipGraphLaunch(ge, s); }, 40, st) / 50;
        printf("  5 us kernels (240 x 256%s): period %.2f in a stream, %.2f per node in a graph\n", store ? ", 1 MB of stores" : "", us, ug);
    }
    // two streams, independent chains (what two chunks in flight see)
    {
        hipStream_t s2; hipStreamCreate(&s2);
        hipEvent_t e0, e1, f1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&f1);
        for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(k_empty, dim3(120), dim3(256), 0, st, buf); hipLaunchKernelGGL(k_empty, dim3(120), dim3(256), 0, s2, buf); }
        hipDeviceSynchronize();
        hipEventRecord(e0, st);
        for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(k_empty, dim3(120), dim3(256), 0, st, buf); hipLaunchKernelGGL(k_empty, dim3(120), dim3(256), 0, s2, buf); }
        hipEventRecord(f1, s2); hipStreamWaitEvent(st, f1, 0); hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  two streams, each a chain of empty kernels (120 x 256): %.2f per launch pair\n", ms * 1e3f / reps);
    }
    return 0;
}
