// How many 256-thread workgroups with X KB of dynamic LDS does a CU hold at once?  (attn_k64.hpp asks for 80 KB per workgroup and needs two per CU.)
// Every workgroup spins ~20 us and records its start / end time and the (XCC, SE, CU) it ran on; the host counts, per CU, the largest number of
// workgroups whose intervals overlap.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/lds_occupancy_probe.hip -o build/lds_occupancy_probe && build/lds_occupancy_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned long long* out, int spin) {
    extern __shared__ char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) lds[0] = 1;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 3 + 0] = t0;
        out[blockIdx.x * 3 + 1] = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x * 3 + 2] = ((unsigned long long)(xcc & 0xf) << 32) | (hw & 0xffffff00u);      // drop wave / SIMD ids (bits 0-5), keep CU / SH / SE
    }
}
int main() {
    const int blocks = 1024, spin = 2000;      // 100 MHz ticks: 20 us
    unsigned long long* d; hipMalloc(&d, blocks * 24);
    for (int kb : {32, 64, 72, 76, 79, 80}) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), kb * 1024, 0, d, spin);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%d KB: launch failed\n", kb); continue; }
        std::vector<unsigned long long> h(blocks * 3);
        hipMemcpy(h.data(), d, blocks * 24, hipMemcpyDeviceToHost);
        std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int b = 0; b < blocks; ++b) {
            ev[h[b * 3 + 2] & ~0x3full].push_back({h[b * 3], +1});
            ev[h[b * 3 + 2] & ~0x3full].push_back({h[b * 3 + 1], -1});
            tmin = std::min(tmin, h[b * 3]); tmax = std::max(tmax, h[b * 3 + 1]);
        }
        int worst = 0, best = 1 << 30;
        for (auto& [cu, v] : ev) {
            std::sort(v.begin(), v.end());
            int cur = 0, mx = 0;
            for (auto& e : v) { cur += e.second; mx = std::max(mx, cur); }
            worst = std::max(worst, mx); best = std::min(best, mx);
        }
        printf("%2d KB of LDS per workgroup: %zu distinct CUs seen, workgroups resident at once per CU: %d ... %d; launch %.1f us (ideal with 2 per CU: %d us)\n",
               kb, ev.size(), best, worst, (tmax - tmin) / 100.0, blocks / 512 * spin / 100);
    }
    return 0;
}
