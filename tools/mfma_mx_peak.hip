// What do the block-scaled MX matrix instructions sustain on FRESH RANDOM operands (power-limited clock), next to the
// f16 MFMA the split products use now?  Planning probe for docs/NOTEBOOK.md section 7 (correction terms on a narrower format).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_mx_peak.hip -o build/mfma_mx_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// FMT: 0 fp8 e4m3, 2 fp6 e2m3, 4 fp4 e2m1 (operand registers beyond the format's width are ignored by the hardware)
template <int FMT>
__global__ __launch_bounds__(256) void k_mx(float* out, int iters, const i32x8* frags, unsigned long long* clk) {
    i32x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = frags[(i * 256 + threadIdx.x) % 2048]; b[i] = frags[((i + 4) * 256 + threadIdx.x) % 2048]; }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(j + u) & 3], b[(j + 2 * u + 1) & 3], acc[j], FMT, FMT, 0,
                                                                         0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { atomicAdd(clk, t1 - t0); atomicAdd(clk + 1, r1 - r0); }
}
__global__ __launch_bounds__(256) void k_f16(float* out, int iters, const i32x8* frags, unsigned long long* clk) {
    f16x8 a[4], b[4];
    const f16x8* fr = reinterpret_cast<const f16x8*>(frags);
    for (int i = 0; i < 4; ++i) { a[i] = fr[(i * 256 + threadIdx.x) % 4096]; b[i] = fr[((i + 4) * 256 + threadIdx.x) % 4096]; }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(j + u) & 3], b[(j + 2 * u + 1) & 3], acc[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { atomicAdd(clk, t1 - t0); atomicAdd(clk + 1, r1 - r0); }
}
template <typename K>
void run(const char* name, K kern, int kdepth, const i32x8* fr) {
    const int blocks = 256 * 8, iters = 4096;
    float* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    unsigned long long* clk; hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 60; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, fr, clk);   // warm clocks
    hipDeviceSynchronize(); hipMemset(clk, 0, 16);
    hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, fr, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const double nm = 10.0 * blocks * 4 * iters * 4;
    printf("%-34s %8.1f TFLOP/s   shader clock %5.0f MHz   (%.2f ms)\n", name, nm * 2.0 * 32 * 32 * kdepth / ms / 1e9,
           (double)c[0] / (double)c[1] * 100.0, ms);
    hipFree(d); hipFree(clk);
}
int main() {
    i32x8* fr; hipMalloc(&fr, 2048 * 32);
    static int h[2048 * 8];
    // random bit patterns with the top exponent bits of every byte cleared (no NaN / Inf codes in any of the formats)
    for (int i = 0; i < 2048 * 8; ++i) h[i] = (rand() ^ (rand() << 11)) & 0x3f3f3f3f;
    hipMemcpy(fr, h, sizeof(h), hipMemcpyHostToDevice);
    run("f16 32x32x16 (today's products)", k_f16, 16, fr);
    run("MX fp8 e4m3 32x32x64", k_mx<0>, 64, fr);
    run("MX fp6 e2m3 32x32x64", k_mx<2>, 64, fr);
    run("MX fp4 e2m1 32x32x64", k_mx<4>, 64, fr);
    return 0;
}
