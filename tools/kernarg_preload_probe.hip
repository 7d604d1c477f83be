// Does a one-scene launch pay for FETCHING its arguments?  A kernel's first instruction that needs an argument waits for an s_load
// from the kernarg segment - a memory round trip behind a kernel boundary that invalidated the scalar cache.  gfx950 can preload the
// first 16 dwords of the kernarg segment into SGPRs at wave launch (hipcc -mllvm -amdgpu-kernarg-preload-count=16), but only for
// arguments passed one by one: a by-value struct (GemmHArgs ...) is a byref argument and is never preloaded.
// Period of a warm dependent chain, 240 workgroups x 256 threads, the body = one load through an argument pointer + a store that
// never happens:  (a) 304-byte struct argument, (b) the same fields as six flat arguments.  Build twice:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kernarg_preload_probe.hip -o build/kp_off
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 tools/kernarg_preload_probe.hip -o build/kp_on
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
struct Big { unsigned w[70]; const int* src; int* dst; int n; int key; };
__global__ void k_struct(Big b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = b.src[i % b.n];
    if (v == b.key) b.dst[i] = v + b.w[3];
}
__global__ void k_flat(const int* src, int* dst, int n, int key, unsigned w3, unsigned w4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = src[i % n];
    if (v == key) dst[i] = v + w3 + w4;
}
__global__ void k_noarg_use(const int* src, int* dst, int n, int key) {     // floor: no argument is read on the executed path
    if (blockIdx.x == 0x7fffffff) dst[0] = src[n] + key;
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
int main() {
    hipStream_t st; hipStreamCreate(&st);
    int* buf; hipMalloc(&buf, 64 << 20); hipMemset(buf, 0, 64 << 20);
    const int reps = 4000;
    Big b{}; b.src = buf; b.dst = buf + (1 << 20); b.n = 4096; b.key = 12345;
    for (int rep = 0; rep < 3; ++rep) {
        const float f0 = chain_us([&](hipStream_t s) { hipLaunchKernelGGL(k_noarg_use, dim3(240), dim3(256), 0, s, b.src, b.dst, b.n, b.key); }, reps, st);
        const float f1 = chain_us([&](hipStream_t s) { hipLaunchKernelGGL(k_struct, dim3(240), dim3(256), 0, s, b); }, reps, st);
        const float f2 = chain_us([&](hipStream_t s) { hipLaunchKernelGGL(k_flat, dim3(240), dim3(256), 0, s, b.src, b.dst, b.n, b.key, 1u, 2u); }, reps, st);
        printf("us per launch: no argument read %.3f | struct argument %.3f | flat arguments %.3f\n", f0, f1, f2);
    }
    return 0;
}
