// LDS-DMA fill-rate probe: how many bytes/s can the chip move global -> LDS with global_load_lds_dwordx4?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half_t;
template <int WAVES, int INFLIGHT>   // tiles of 8 x 1 KB per wave in flight
__global__ __launch_bounds__(WAVES * 64) void k(const char* src, size_t bytes_per_block, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wid = tid >> 6;
    const char* base = src + (size_t)blockIdx.x * bytes_per_block;
    const size_t tile = (size_t)WAVES * 8 * 1024;
    const int ntile = bytes_per_block ? (int)(bytes_per_block / tile) : 2;
    auto issue = [&](int t) {
        const char* s = base + (size_t)(t % ntile) * tile + wid * 8192 + (tid & 63) * 16;
        char* d = lds + ((t % (INFLIGHT + 1)) * WAVES + wid) * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + i * 1024),
                                             (__attribute__((address_space(3))) void*)(d + i * 1024), 16, 0, 0);
    };
    for (int t = 0; t < INFLIGHT; ++t) issue(t);
    for (int t = 0; t < iters; ++t) {
        issue(t + INFLIGHT);
        if (INFLIGHT == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if (INFLIGHT == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        if (INFLIGHT == 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) sink[blockIdx.x] = lds[0];
}
template <int WAVES, int INFLIGHT>
void run(const char* name, const char* buf, size_t bytes_per_block, int blocks, int iters) {
    float* sink; hipMalloc(&sink, blocks * 4);
    size_t lds = (size_t)(INFLIGHT + 1) * WAVES * 8192;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<WAVES, INFLIGHT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<WAVES, INFLIGHT>), dim3(blocks), dim3(WAVES * 64), lds, 0, buf, bytes_per_block, 10, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WAVES, INFLIGHT>), dim3(blocks), dim3(WAVES * 64), lds, 0, buf, bytes_per_block, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = (double)blocks * iters * WAVES * 8192;
    printf("%-44s %7.3f ms  %6.2f TB/s  (%5.1f GB/s per CU at %d blocks)\n", name, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256, blocks);
    hipFree(sink);
}
int main() {
    char* buf; size_t total = (size_t)2 << 30; hipMalloc(&buf, total); hipMemset(buf, 1, total);
    // L2-resident: every block re-reads the same small region (64 KB / block -> 16 MB total for 256 blocks)
    run<4, 2>("4 waves, 2 tiles in flight, 64 KB/block (L2)", buf, 65536, 256, 4000);
    // every workgroup streams the SAME region (stride 0): the GEMM's W panel / attention's K,V pattern
    run<4, 2>("SHARED 64 KB region, all 256 blocks", buf, 0, 256, 4000);
    run<4, 2>("SHARED 64 KB region, 512 blocks", buf, 0, 512, 4000);
    run<4, 3>("4 waves, 3 tiles in flight, 64 KB/block (L2)", buf, 65536, 256, 4000);
    run<8, 2>("8 waves, 2 tiles in flight, 128 KB/block (L2)", buf, 131072, 256, 2000);
    run<4, 2>("4 waves x2 blocks/CU, 64 KB/block (L2)", buf, 65536, 512, 4000);
    // Infinity-cache resident: 512 KB per block = 128 MB
    run<4, 2>("4 waves, 2 in flight, 512 KB/block (MALL)", buf, 524288, 256, 4000);
    run<8, 2>("8 waves, 2 in flight, 512 KB/block (MALL)", buf, 524288, 256, 2000);
    // HBM streaming: 8 MB per block = 2 GB
    run<4, 2>("4 waves, 2 in flight, 8 MB/block (HBM)", buf, (size_t)8 << 20, 256, 4000);
    run<8, 2>("8 waves, 2 in flight, 8 MB/block (HBM)", buf, (size_t)8 << 20, 256, 2000);
    return 0;
}
