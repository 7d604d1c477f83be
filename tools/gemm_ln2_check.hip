// gemm_ln2_mx_kernel (F16MX: out_proj / linear2 + residual + LayerNorm in one launch) - its tile shapes against each other on the same random
// operands: every word of the result planes, and the time per launch in alternation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops -w \
//         -DJMID_DIAGNOSTICS -I safe-interactive-crowdnav_amd/csrc -I include tools/gemm_ln2_check.hip -o build/gemm_ln2_check
//   build/gemm_ln2_check [M = 61200] [K = 512] [reps = 20] [variant list, e.g. 128,64,1]      (ln_rows values; 1 = the persistent two-tile form)
#include "gemm_ln2_mx.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
using namespace jmid;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 61200, K = argc > 2 ? atoi(argv[2]) : 512, reps = argc > 3 ? atoi(argv[3]) : 20;
    std::vector<int> variants;
    { const char* v = argc > 4 ? argv[4] : "128,64"; for (const char* p = v; *p;) { variants.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p) ++p; } }
    const int d = 512;
    const size_t Mpad = ((size_t)M + 127) / 128 * 128 + 128;
    std::mt19937 rng(11);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<half_t> A(Mpad * K), W16((size_t)K * d), Xh(Mpad * d);
    std::vector<unsigned char> W8((size_t)K * d), Xl(Mpad * d);
    for (auto& v : A) v = (half_t)nd(rng);
    for (auto& v : W16) v = (half_t)(nd(rng) * 256.f / sqrtf((float)K));      // the 2^8-scaled weight
    for (auto& v : W8) v = (unsigned char)((rng() & 0x7f) < 0x38 ? (rng() & 0xbf) % 0x38 : 0x20);      // small bf8 values
    for (auto& v : Xh) v = (half_t)(nd(rng) * 2.f);
    for (auto& v : Xl) v = (unsigned char)(0x10 + (rng() % 8));
    std::vector<float> bias(d), gamma(d), beta(d);
    for (int i = 0; i < d; ++i) { bias[i] = nd(rng); gamma[i] = 1.f + 0.1f * nd(rng); beta[i] = 0.1f * nd(rng); }
    half_t *dA, *dW, *dXh0; unsigned char *dW8, *dXl0; float *db, *dg, *dt; int* flag;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dW, W16.size() * 2)); CK(hipMalloc(&dW8, W8.size()));
    CK(hipMalloc(&dXh0, Xh.size() * 2)); CK(hipMalloc(&dXl0, Xl.size()));
    CK(hipMalloc(&db, d * 4)); CK(hipMalloc(&dg, d * 4)); CK(hipMalloc(&dt, d * 4)); CK(hipMalloc(&flag, 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W16.data(), W16.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW8, W8.data(), W8.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dXh0, Xh.data(), Xh.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dXl0, Xl.data(), Xl.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(db, bias.data(), d * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, gamma.data(), d * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, beta.data(), d * 4, hipMemcpyHostToDevice)); CK(hipMemset(flag, 0, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t nv = variants.size();
    std::vector<half_t*> dXh(nv); std::vector<unsigned char*> dXl(nv);
    std::vector<std::vector<half_t>> rh(nv); std::vector<std::vector<unsigned char>> rl(nv);
    for (size_t v = 0; v < nv; ++v) { CK(hipMalloc(&dXh[v], Xh.size() * 2)); CK(hipMalloc(&dXl[v], Xl.size())); }
    for (int round = 0; round < 3; ++round)
        for (size_t v = 0; v < nv; ++v) {
            Tuning tn; tn.ln_rows = variants[v];
            TuneScope ts(&tn);
            GemmLn2Args g{dA, dW, dW8, db, dg, dt, dXh[v], dXl[v], M, K, 1e-5f, flag, 0};
            // the result of ONE application to the pristine residual planes (kept for the comparison) ...
            CK(hipMemcpyAsync(dXh[v], dXh0, Xh.size() * 2, hipMemcpyDeviceToDevice, st)); CK(hipMemcpyAsync(dXl[v], dXl0, Xl.size(), hipMemcpyDeviceToDevice, st));
            CK(launch_gemm_ln2_mx(g, st));
            CK(hipStreamSynchronize(st));
            if (round == 0) {
                rh[v].resize(Xh.size()); rl[v].resize(Xl.size());
                CK(hipMemcpy(rh[v].data(), dXh[v], Xh.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(rl[v].data(), dXl[v], Xl.size(), hipMemcpyDeviceToHost));
            }
            // ... then the timing (the planes are rewritten in place: LayerNorm output stays O(1), the time does not depend on the values)
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) CK(launch_gemm_ln2_mx(g, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double fl = 2.0 * M * (double)d * K;
            printf("variant %4d: %.2f us per launch  (%.0f TFLOP/s algorithmic)\n", variants[v], ms / reps * 1e3, fl / (ms / reps * 1e-3) * 1e-12);
        }
    for (size_t v = 1; v < nv; ++v) {
        size_t nh = 0, nl = 0;
        for (int m = 0; m < M; ++m)
            for (int c = 0; c < d; ++c) {
                const size_t ih = blk_index(m, c, d), il = blk8_index(m, c, d);
                nh += __builtin_bit_cast(unsigned short, rh[v][ih]) != __builtin_bit_cast(unsigned short, rh[0][ih]);
                nl += rl[v][il] != rl[0][il];
            }
        printf("ln_rows = %d against %d: %zu hi words, %zu lo bytes differ of %zu\n", variants[v], variants[0], nh, nl, (size_t)M * d);
    }
    int f = 0; CK(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost));
    printf("range flag %d\n", f);
    return 0;
}
