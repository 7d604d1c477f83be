// Does v_cvt_pk_f16_f32 (what __builtin_convertvector(float2 -> half2) lowers to on gfx950) give the bits of the scalar
// v_cvt_f16_f32 for every input?  Sweeps fp32 bit patterns incl. the fp16 denormal range and ties.
//   hipcc --offload-arch=gfx950 -O3 tools/cvt_pk_probe.hip -o build/cvt_pk_probe && build/cvt_pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned long long* counts, unsigned* first) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
    unsigned long long bad = 0, bad_den = 0, total = 0;
    for (unsigned long long b = tid; b < (1ull << 32); b += (unsigned long long)n * 97) {     // 1/97 of all patterns
        float v = __builtin_bit_cast(float, (unsigned)b);
        asm volatile("" : "+v"(v));
        const _Float16 s = (_Float16)v;
        const f16x2 p = __builtin_convertvector(f32x2{v, -v}, f16x2);
        const unsigned short sb = __builtin_bit_cast(unsigned short, s), pb = __builtin_bit_cast(unsigned short, p[0]);
        const bool nan = (sb & 0x7fff) > 0x7c00;
        ++total;
        if (!nan && sb != pb) {
            ++bad;
            if ((sb & 0x7c00) == 0 || (pb & 0x7c00) == 0) ++bad_den;
            if (atomicCAS(first, 0u, (unsigned)b) == 0u) { first[1] = sb; first[2] = pb; }
        }
    }
    atomicAdd(&counts[0], total); atomicAdd(&counts[1], bad); atomicAdd(&counts[2], bad_den);
}
int main() {
    unsigned long long* c; unsigned* f;
    hipMalloc(&c, 24); hipMalloc(&f, 12); hipMemset(c, 0, 24); hipMemset(f, 0, 12);
    probe<<<1024, 256>>>(c, f);
    unsigned long long h[3]; unsigned hf[3];
    hipMemcpy(h, c, 24, hipMemcpyDeviceToHost); hipMemcpy(hf, f, 12, hipMemcpyDeviceToHost);
    printf("patterns %llu  packed != scalar: %llu  (of which denormal results: %llu)  first: f32 0x%08x scalar 0x%04x packed 0x%04x\n", h[0], h[1], h[2], hf[0], hf[1], hf[2]);
    return 0;
}
