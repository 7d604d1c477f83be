// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs (no flush-to-zero)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float av, float bv, float* out) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float tests[][2] = {{ldexpf(1, -20), 1.f}, {ldexpf(1, -24), 1024.f}, {ldexpf(1, -15), 1.f}, {ldexpf(3, -24), ldexpf(1, -10)},
                              {ldexpf(1, -14), 1.f}, {ldexpf(1, -24), ldexpf(1, -24)}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, t[0], t[1], d);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g : mfma=%g expected=%g %s\n", t[0], t[1], h, 16.0 * t[0] * t[1], fabs(h - 16.0 * t[0] * t[1]) <= 1e-6 * fabs(16.0 * t[0] * t[1]) ? "OK" : "MISMATCH");
    }
    return 0;
}
