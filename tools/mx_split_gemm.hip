// Numerics of a split product whose two correction terms run on the block-scaled MX fp8 matrix path (docs/NOTEBOOK.md section 7):
//   D = A_hi . W_hi  [fp16 MFMA]  +  fp8(A_hi) . mx8(W_lo)  +  mx8(A_lo) . fp8(W_hi)  [v_mfma_scale_f32_32x32x64_f8f6f4]
// on a 32 x 32 tile with K = 512 (the d_model contraction), against fp64, next to the pure-fp16 variants of today.
// Operands are packed per lane on the host (layouts: gemm_f16x3.hpp header for f16, tools/mfma_mx_layout.hip for MX).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mx_split_gemm.hip -o build/mx_split_gemm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int K = 512, NB16 = K / 16, NB64 = K / 64;

// mode 0: hi.hi   1: + hi.lo (F16X2)   2: + lo.hi too (F16X3)   3: hi.hi + both corrections in MX fp8
__global__ void k_tile(const f16x8* Ah, const f16x8* Al, const f16x8* Wh, const f16x8* Wl, const i32x8* A8, const i32x8* Al8,
                       const i32x8* W8, const i32x8* Wl8, const int* sA8, const int* sAl8, const int* sW8, const int* sWl8,
                       float* D, int mode) {
    const int l = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kb = 0; kb < NB16; ++kb) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[kb * 64 + l], Wh[kb * 64 + l], acc, 0, 0, 0);
        if (mode == 1 || mode == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[kb * 64 + l], Wl[kb * 64 + l], acc, 0, 0, 0);
        if (mode == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[kb * 64 + l], Wh[kb * 64 + l], acc, 0, 0, 0);
    }
    if (mode == 3)
        for (int kb = 0; kb < NB64; ++kb) {
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A8[kb * 64 + l], Wl8[kb * 64 + l], acc, 0, 0, 0, sA8[kb * 64 + l], 0, sWl8[kb * 64 + l]);
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Al8[kb * 64 + l], W8[kb * 64 + l], acc, 0, 0, 0, sAl8[kb * 64 + l], 0, sW8[kb * 64 + l]);
        }
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
static float e4m3_value(int code) {
    const int s = code >> 7, e = (code >> 3) & 15, m = code & 7;
    if (e == 15 && m == 7) return NAN;
    const float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}
static int e4m3_encode(float x) {
    int best = 0; float bd = INFINITY;
    for (int c = 0; c < 256; ++c) { const float v = e4m3_value(c); if (std::isnan(v)) continue; const float d = fabsf(v - x); if (d < bd) { bd = d; best = c; } }
    return best;
}
// X: [32 rows][K] (row = output row for A, output column for W).  Packs MX fp8 operands + scales for every k64 block.
static void pack_mx(const std::vector<float>& X, std::vector<int>& regs, std::vector<int>& scales) {
    regs.assign(NB64 * 64 * 8, 0); scales.assign(NB64 * 64, 0);
    for (int kb = 0; kb < NB64; ++kb)
        for (int row = 0; row < 32; ++row)
            for (int b = 0; b < 2; ++b) {
                const float* x = &X[row * K + kb * 64 + 32 * b];
                float mx = 0.f; for (int t = 0; t < 32; ++t) mx = fmaxf(mx, fabsf(x[t]));
                int e = mx > 0.f ? (int)ceilf(log2f(mx / 448.0f)) : -127; if (e < -127) e = -127;
                scales[kb * 64 + row + 32 * b] = e + 127;
                for (int t = 0; t < 32; ++t) {          // k = 32 b + t = 32 (p/16) + 16 h + p%16  ->  h = t / 16, p = 16 b + t % 16
                    const int h = t / 16, p = 16 * b + t % 16, lane = row + 32 * h;
                    regs[(kb * 64 + lane) * 8 + p / 4] |= e4m3_encode(ldexpf(x[t], -e)) << (8 * (p % 4));
                }
            }
}
static void pack_f16(const std::vector<float>& X, std::vector<_Float16>& hi, std::vector<_Float16>& lo, std::vector<float>& hif, std::vector<float>& lof) {
    hi.resize(NB16 * 64 * 8); lo.resize(NB16 * 64 * 8); hif.resize(32 * K); lof.resize(32 * K);
    for (int row = 0; row < 32; ++row)
        for (int k = 0; k < K; ++k) {
            const _Float16 h = (_Float16)X[row * K + k], l = (_Float16)(X[row * K + k] - (float)h);
            hif[row * K + k] = (float)h; lof[row * K + k] = (float)l;
            const int kb = k / 16, kk = k % 16, lane = row + 32 * (kk / 8);
            hi[(kb * 64 + lane) * 8 + kk % 8] = h; lo[(kb * 64 + lane) * 8 + kk % 8] = l;
        }
}
template <typename T> static T* up(const std::vector<T>& v) { T* p; hipMalloc(&p, v.size() * sizeof(T)); hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); return p; }
int main() {
    std::vector<float> A(32 * K), W(32 * K);
    for (auto& v : A) v = ((rand() & 65535) - 32768) / 16384.0f;                    // activations, O(1)
    for (auto& v : W) v = ((rand() & 65535) - 32768) / 32768.0f / sqrtf((float)K);   // weights, ~1/sqrt(K)
    std::vector<_Float16> Ah, Al, Wh, Wl; std::vector<float> Ahf, Alf, Whf, Wlf;
    pack_f16(A, Ah, Al, Ahf, Alf); pack_f16(W, Wh, Wl, Whf, Wlf);
    std::vector<int> A8, Al8, W8, Wl8, sA8, sAl8, sW8, sWl8;
    pack_mx(Ahf, A8, sA8); pack_mx(Alf, Al8, sAl8); pack_mx(Whf, W8, sW8); pack_mx(Wlf, Wl8, sWl8);
    float* dD; hipMalloc(&dD, 32 * 32 * 4);
    auto dAh = up(Ah), dAl = up(Al), dWh = up(Wh), dWl = up(Wl);
    auto dA8 = up(A8), dAl8 = up(Al8), dW8 = up(W8), dWl8 = up(Wl8), dsA8 = up(sA8), dsAl8 = up(sAl8), dsW8 = up(sW8), dsWl8 = up(sWl8);
    const char* names[4] = {"A_hi.W_hi (one fp16 MFMA)", "+ A_hi.W_lo in fp16 (F16X2)", "+ A_lo.W_hi in fp16 too (F16X3)", "A_hi.W_hi + both corrections in MX fp8"};
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k_tile, dim3(1), dim3(64), 0, 0, (const f16x8*)dAh, (const f16x8*)dAl, (const f16x8*)dWh, (const f16x8*)dWl,
                           (const i32x8*)dA8, (const i32x8*)dAl8, (const i32x8*)dW8, (const i32x8*)dWl8, dsA8, dsAl8, dsW8, dsWl8, dD, mode);
        std::vector<float> D(32 * 32); hipMemcpy(D.data(), dD, 32 * 32 * 4, hipMemcpyDeviceToHost);
        double err = 0, ref = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double r = 0; for (int k = 0; k < K; ++k) r += (double)A[i * K + k] * (double)W[j * K + k];
                err += (D[i * 32 + j] - r) * (D[i * 32 + j] - r); ref += r * r;
            }
        printf("%-42s rms error / rms value = %.3e  (2^%.1f)\n", names[mode], sqrt(err / ref), log2(sqrt(err / ref)));
    }
    return 0;
}
