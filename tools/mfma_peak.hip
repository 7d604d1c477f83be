// MFMA issue-rate probe: how fast does v_mfma_f32_32x32x16_f16 really go (independent vs dependent accumulators)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_f16(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// random operands that change from MFMA to MFMA (data toggling costs power -> the clock the chip really sustains);
// clk[0..1] += (s_memtime, s_memrealtime) deltas of wave 0 of every block
template <int NACC>
__global__ __launch_bounds__(256) void k_f16_rand(float* out, int iters, const f16x8* frags, unsigned long long* clk) {
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = frags[(i * 256 + threadIdx.x) % 2048]; b[i] = frags[((i + 4) * 256 + threadIdx.x) % 2048]; }
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(j + u) & 3], b[(j + 2 * u + 1) & 3], acc[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { atomicAdd(clk, t1 - t0); atomicAdd(clk + 1, r1 - r0); }
}
template <int NACC>
void run_rand(const char* name, int blocks, int iters) {
    float* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    f16x8* fr; hipMalloc(&fr, 2048 * 16);
    _Float16 h[2048 * 8]; for (int i = 0; i < 2048 * 8; ++i) h[i] = (_Float16)(((rand() & 4095) - 2048) / 2048.0f);
    hipMemcpy(fr, h, sizeof(h), hipMemcpyHostToDevice);
    unsigned long long* clk; hipMalloc(&clk, 16); hipMemset(clk, 0, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_f16_rand<NACC>, dim3(blocks), dim3(256), 0, 0, d, 16, fr, clk);
    hipDeviceSynchronize(); hipMemset(clk, 0, 16);
    // warm clocks first: a GPU coming out of idle runs the first milliseconds far below its sustained clock
    for (int w = 0; w < 200; ++w) hipLaunchKernelGGL(k_f16_rand<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, fr, clk);
    hipDeviceSynchronize(); hipMemset(clk, 0, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_f16_rand<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, fr, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    double nm = (double)blocks * 4 * iters * NACC;
    printf("%-28s blocks=%d: %.3f ms  %.1f TFLOP/s   shader clock %.0f MHz\n", name, blocks, ms, nm * 2.0 * 32 * 32 * 16 / ms / 1e9,
           (double)c[0] / (double)c[1] * 100.0);
    hipFree(d);
}
template <int NACC>
__global__ __launch_bounds__(256) void k_f32(float* out, int iters) {
    float a = threadIdx.x * 0.001f, b = 0.5f;
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K kern, int blocks, int threads, int iters, int nacc, double flop_per_mfma) {
    float* d; hipMalloc(&d, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 200; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters);   // warm clocks
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double nm = (double)blocks * (threads / 64) * iters * nacc;
    printf("%-28s blocks=%d waves/blk=%d: %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz if 1 wave/SIMD)\n", name, blocks,
           threads / 64, ms, nm * flop_per_mfma / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * nacc * (threads / 256.0)));
    hipFree(d);
}
int main(int argc, char** argv) {
    const double f16 = 2.0 * 32 * 32 * 16, f32 = 2.0 * 32 * 32 * 2;
    if (argc > 1) {      // `mfma_peak sustained`: only the rate a pure fp16 MFMA loop sustains on fresh random operands, warm clocks, two
                         // waves per SIMD on every CU (what bench.py quotes roofline.frac_of_sustained against)
        run_rand<4>("f16 RANDOM data 4acc 2w/SIMD", 512, 20000);
        return 0;
    }
    run("f16 32x32x16 4acc 1w/SIMD", k_f16<4>, 256, 256, 20000, 4, f16);
    run("f16 32x32x16 4acc 2w/SIMD", k_f16<4>, 512, 256, 20000, 4, f16);
    run("f16 32x32x16 1acc 1w/SIMD", k_f16<1>, 256, 256, 40000, 1, f16);
    run("f16 32x32x16 2acc 1w/SIMD", k_f16<2>, 256, 256, 40000, 2, f16);
    run("f16 32x32x16 8acc 1w/SIMD", k_f16<8>, 256, 256, 10000, 8, f16);
    run_rand<4>("f16 RANDOM data 4acc 1w/SIMD", 256, 20000);
    run_rand<4>("f16 RANDOM data 4acc 2w/SIMD", 512, 20000);
    run_rand<4>("f16 RANDOM data 4acc, 64 CUs", 64, 20000);
    run("f32 32x32x2  4acc 1w/SIMD", k_f32<4>, 256, 256, 20000, 4, f32);
    return 0;
}
