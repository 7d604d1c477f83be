#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    if (threadIdx.x == 0) {
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hwid;
    }
}
int main() {
    for (int threads : {256, 512}) {
        int n = 64; int* d; hipMalloc(&d, n * 8);
        hipLaunchKernelGGL(k, dim3(n), dim3(threads), 0, 0, d);
        int h[128]; hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost);
        printf("threads=%d block->xcc: ", threads);
        for (int i = 0; i < 32; ++i) printf("%d ", h[2 * i]);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
