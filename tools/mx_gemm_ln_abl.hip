// Where does the time of gemm_ln128_mx_kernel go?  ABL bit 0 = no fp16 MFMAs, 1 = no fp8 MFMAs, 3 = no epilogue.  TIMING ONLY.
// (generated from gemm_ln_f16x3.hpp by the python snippet in the commit message; operands are random bytes)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -I safe-interactive-crowdnav_amd/csrc tools/mx_gemm_ln_abl.hip -o build/mx_gemm_ln_abl
#include "gemm_ln_f16x3.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace jmid;
template <int ABL>
__global__ __launch_bounds__(512, 2) void probe_kernel(GemmLnArgs g, int ntm) {
    constexpr int WM = 4, WN = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wc = wid;
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int tm = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const int m0 = tm * GLN2_BM;
    const int nk = g.K / 32, nsteps = 2 * nk, nkb = g.K / 64;

    auto dma16 = [](const void* s, void* d) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s, (__attribute__((address_space(3))) void*)d, 16, 0, 0);
    };
    const half_t* a_hi = g.Ahi + (size_t)tm * nk * 4096 + tid * 8;
    auto issueA = [&](int ka) {     // one wave-instruction; past the end: the last tile again into its own stage
        const int kk = ka < nk ? ka : nk - 1;
        dma16(a_hi + (size_t)kk * 4096, lds + GLNX_A_OFF + (kk % 3) * GLNX_A_STAGE + wid * 512);
    };
    auto issueW = [&](int s, int stage) {
        half_t* st = lds + stage * GLNX_W_STAGE + wc * 64 * 16;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            dma16(g.W16hi + ((size_t)s * GLN_BN + wc * 64 + q * 32) * 16 + lane * 8, st + q * 512);
    };
    unsigned char* w8buf = lds_raw + GLNX_W8_OFF + wc * 4096;      // [2 column blocks][2 pieces][64 lanes][16 B]
    const unsigned char* w8src = g.W8 + (size_t)wc * 2 * 2048 + lane * 16;
    auto issueW8 = [&](int kb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16(w8src + (size_t)kb * (GLN_BN / 32) * 2048 + q * 1024, w8buf + q * 1024);
    };
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int offA[WM][2], offW[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int row = i * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offA[i][ks] = row * 32 + (((ks * 2 + hi) ^ ((row >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) offW[j] = (wc * 64 + j * 32 + l31) * 16 + hi * 8;
    i32x8 a8[WM];

    issueW8(0);
    issueA(0);
    issueW(0, 0);
    issueA(1);
    issueW(1, 1);
    int wst = 0, ast = 0;
    auto step = [&](const int s, auto q_c) {
        constexpr int Q = decltype(q_c)::value, ks = Q & 1;
        if (s + 1 >= nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (Q <= 1 && s >= 4) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        if (ks == 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < nsteps) issueW(s + 2, wst == 0 ? 2 : wst - 1);
        if (ks == 0) issueA((s >> 1) + 2);
        const half_t* stA = lds + GLNX_A_OFF + ast * GLNX_A_STAGE;
        const half_t* stW = lds + wst * GLNX_W_STAGE;
        f16x8 ah[WM], wh[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) ah[i] = *reinterpret_cast<const f16x8*>(stA + offA[i][ks]);
#pragma unroll
        for (int j = 0; j < WN; ++j) wh[j] = *reinterpret_cast<const f16x8*>(stW + offW[j]);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) { if (!(ABL & 1)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc[i][j], 0, 0, 0); else asm volatile("" ::"v"(ah[i]), "v"(wh[j])); }
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const i32x4 dw = __builtin_bit_cast(i32x4, ah[i]);
            a8[i][Q * 2 + 0] = bf8_of_f16x4(dw[0], dw[1]);
            a8[i][Q * 2 + 1] = bf8_of_f16x4(dw[2], dw[3]);
        }
        if (Q == 3) {
            i32x8 w8[WN];
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const unsigned char* p = w8buf + j * 2048 + lane * 16;
                const i32x4 lo = *reinterpret_cast<const i32x4*>(p);
                const i32x4 up = *reinterpret_cast<const i32x4*>(p + 1024);
                w8[j] = i32x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    { if (!(ABL & 2)) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], w8[j], acc[i][j], 1, 1, 0, 0, 0, 0); else asm volatile("" ::"v"(a8[i]), "v"(w8[j])); }
            __builtin_amdgcn_sched_barrier(0);
            if ((s >> 2) + 1 < nkb) issueW8((s >> 2) + 1);
        }
        wst = wst == 2 ? 0 : wst + 1;
        if (ks == 1) ast = ast == 2 ? 0 : ast + 1;
    };
    for (int s = 0; s < nsteps; s += 4) {
        step(s, std::integral_constant<int, 0>{});
        step(s + 1, std::integral_constant<int, 1>{});
        step(s + 2, std::integral_constant<int, 2>{});
        step(s + 3, std::integral_constant<int, 3>{});
    }
    if (ABL & 8) { float s = 0.f; for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r]; if (s == 1234.5f) g.Xh[tid] = (half_t)s; return; }
    gln128_epilogue(g, acc, lds_raw, m0, wid, wc, lane, l31, hi);
}


template <int ABL>
float run(GemmLnArgs g, int reps) {
    const int ntm = (g.M + 127) / 128;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GLNX_LDS_BYTES);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe_kernel<ABL>, dim3(ntm), dim3(512), GLNX_LDS_BYTES, 0, g, ntm);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe_kernel<ABL>, dim3(ntm), dim3(512), GLNX_LDS_BYTES, 0, g, ntm);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}
int main() {
    const int M = 61200;
    for (int K : {512, 1024}) {
        auto dev_rand = [&](size_t bytes, int mask) {
            std::vector<unsigned short> h(bytes / 2);
            for (auto& v : h) v = (unsigned short)(rand() & mask);
            void* p; hipMalloc(&p, bytes); hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice); return p;
        };
        GemmLnArgs g{};
        g.Ahi = (half_t*)dev_rand(blk_plane_elems(M, K) * 2, 0x3f3f); g.Alo = g.Ahi;
        g.W16hi = (half_t*)dev_rand((size_t)512 * K * 2, 0x3f3f); g.W16lo = g.W16hi;
        g.W8 = (unsigned char*)dev_rand((size_t)512 * K, 0x3f3f);
        std::vector<float> hb(512, 0.1f); float* v; hipMalloc(&v, 512 * 4); hipMemcpy(v, hb.data(), 512 * 4, hipMemcpyHostToDevice);
        g.bias = v; g.gamma = v; g.beta = v;
        g.Xh = (half_t*)dev_rand(blk_plane_elems(M, 512) * 2, 0x3f3f); g.Xl = (half_t*)dev_rand(blk_plane_elems(M, 512) * 2, 0x0f3f);
        g.M = M; g.K = K; g.eps = 1e-5f; g.x2 = 1;
        hipMalloc(&g.range_flag, 4); hipMemset(g.range_flag, 0, 4);
        const double fl = 2.0 * M * 512 * K;
        for (int rep = 0; rep < 2; ++rep)
            printf("K %d: full %.1f us (%.0f TF) | no epilogue %.1f | no fp16 MFMA %.1f | no fp8 MFMA %.1f | no MFMA at all %.1f | no MFMA, no epilogue %.1f\n",
                   K, run<0>(g, 40), fl / run<0>(g, 40) / 1e6, run<8>(g, 40), run<1>(g, 40), run<2>(g, 40), run<3>(g, 40), run<11>(g, 40));
    }
    return 0;
}
