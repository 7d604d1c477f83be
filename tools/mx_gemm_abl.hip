// Where does the time of gemm_mx_dma256x256_kernel go?  The kernel's K loop (3-stage ring, five DMA instructions per k32 tile)
// with pieces switched off: ABL bit 0 = no fp16 MFMAs, 1 = no fp8 MFMAs, 2 = no DMA after the prologue, 3 = no epilogue,
// 4 = no bf8 packing of the A fragments, 5 = no barriers.  TIMING ONLY (operands are random bytes).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -Xclang -target-feature -Xclang -packed-fp32-ops -I safe-interactive-crowdnav_amd/csrc tools/mx_gemm_abl.hip -o build/mx_gemm_abl
#include "gemm_f16x3.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace jmid;
// the probe keeps the layout of the first version of the kernel: three 32 KB stages, then two 16 KB fp8-image buffers (halves per tile)
constexpr int MX_STAGE = 4 * DMA_PLANE;
constexpr size_t MX_W8_OFF = size_t(3) * MX_STAGE * sizeof(half_t);
constexpr size_t MX_LDS_BYTES = MX_W8_OFF + 2 * 16384;

template <int ABL>
__global__ __launch_bounds__(512, 2) void probe_kernel(GemmHArgs g, int ntm, int ntn) {
    constexpr int WM = 2, WN = 4, BM = 256, BN = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wid >> 1, wc = wid & 1;
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const int tm = swz / ntn, tn = swz - tm * ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = g.K / GEMMH_BK;
    const int nrb = (g.M + 127) / 128;
    const int rb0 = 2 * tm, rb1 = (2 * tm + 1 < nrb) ? 2 * tm + 1 : nrb - 1;
    const half_t* src[4];
    src[0] = g.Ahi + (size_t)rb0 * nk * 4096 + tid * 8;
    src[1] = g.Ahi + (size_t)rb1 * nk * 4096 + tid * 8;
    src[2] = g.Whi + (size_t)(2 * tn) * nk * 4096 + tid * 8;
    src[3] = g.Whi + (size_t)(2 * tn + 1) * nk * 4096 + tid * 8;
    const unsigned char* src8 = g.W8 + (size_t)tn * 8 * 2048 + tid * 16;
    const size_t w8_kstride = (size_t)(g.N / 32) * 2048;
    unsigned char* lds8 = lds_raw + MX_W8_OFF;
    auto dma16 = [](const void* s, void* d) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s, (__attribute__((address_space(3))) void*)d, 16, 0, 0);
    };
    auto issue = [&](int kt, int stg) {
        if ((ABL & 4) && kt > 1) return;
        half_t* st = lds + stg * MX_STAGE + wid * 512;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(src[i] + (size_t)kt * 4096, st + i * 4096);
        dma16(src8 + (size_t)(kt >> 1) * w8_kstride + (kt & 1) * 8192, lds8 + ((kt >> 1) & 1) * 16384 + (kt & 1) * 8192 + wid * 1024);
    };
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int offA[WM][2], offW[WN][2];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int row = wr * 64 + i * 32 + l31, r = row & 127;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offA[i][ks] = (row >> 7) * DMA_PLANE + r * 32 + (((ks * 2 + hi) ^ ((r >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int row = wc * 128 + j * 32 + l31, r = row & 127;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offW[j][ks] = (2 + (row >> 7)) * DMA_PLANE + r * 32 + (((ks * 2 + hi) ^ ((r >> 2) & 3)) * 8);
    }
    i32x8 a8[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) a8[i][e] = 0x3c3c3c3c;
    auto tile = [&](const half_t* st, auto half_c) {
        constexpr int HALF = decltype(half_c)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah[WM], wh[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) ah[i] = *reinterpret_cast<const f16x8*>(st + offA[i][ks]);
#pragma unroll
            for (int j = 0; j < WN; ++j) wh[j] = *reinterpret_cast<const f16x8*>(st + offW[j][ks]);
            if (!(ABL & 1)) {
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < WM; ++i) asm volatile("" ::"v"(ah[i]));
#pragma unroll
                for (int j = 0; j < WN; ++j) asm volatile("" ::"v"(wh[j]));
            }
            if (!(ABL & 16)) {
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    const i32x4 d = __builtin_bit_cast(i32x4, ah[i]);
                    a8[i][HALF * 4 + ks * 2 + 0] = bf8_of_f16x4(d[0], d[1]);
                    a8[i][HALF * 4 + ks * 2 + 1] = bf8_of_f16x4(d[2], d[3]);
                }
            }
        }
    };
    auto top = [&](int kt, int stg_next) {
        if (kt + 1 < nk && !(ABL & 4)) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(ABL & 32)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) issue(kt + 2, stg_next);
    };
    issue(0, 0);
    issue(1, 1);
    int stg = 0;
    for (int kt = 0; kt < nk; kt += 2) {
        const int s1 = stg == 2 ? 0 : stg + 1, s2 = s1 == 2 ? 0 : s1 + 1;
        top(kt, s2);
        tile(lds + stg * MX_STAGE, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        top(kt + 1, stg);
        tile(lds + s1 * MX_STAGE, std::integral_constant<int, 1>{});
        {
            const unsigned char* wb = lds8 + ((kt >> 1) & 1) * 16384;
            i32x8 w8[WN];
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const unsigned char* p = wb + (size_t)((wc * 4 + j) * 2) * 1024 + lane * 16;
                const i32x4 lo = *reinterpret_cast<const i32x4*>(p);
                const i32x4 up = *reinterpret_cast<const i32x4*>(p + 1024);
                w8[j] = i32x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
            }
            if (!(ABL & 2)) {
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], w8[j], acc[i][j], 1, 1, 0, 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < WN; ++j) asm volatile("" ::"v"(w8[j]));
#pragma unroll
                for (int i = 0; i < WM; ++i) asm volatile("" ::"v"(a8[i]));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        stg = s2;
    }
    if (ABL & 8) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 1234.5f) g.C[tid] = s;
        return;
    }
    gemm_h_epilogue<WM, WN, EPI_BIAS_RELU, OUT_SPLIT, true>(g, acc, m0, n0, wr, wc, l31, hi, BM, BN);
}

template <int ABL>
float run(GemmHArgs g, int reps) {
    const int ntm = (g.M + 255) / 256, ntn = g.N / 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)MX_LDS_BYTES);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe_kernel<ABL>, dim3(ntm * ntn), dim3(512), MX_LDS_BYTES, 0, g, ntm, ntn);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe_kernel<ABL>, dim3(ntm * ntn), dim3(512), MX_LDS_BYTES, 0, g, ntm, ntn);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main() {
    const int M = 61200, K = 512;
    for (int N : {1024, 1536}) {
        auto dev_rand = [&](size_t bytes, int mask) {
            std::vector<unsigned short> h(bytes / 2);
            for (auto& v : h) v = (unsigned short)(rand() & mask);
            void* p; hipMalloc(&p, bytes); hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice); return p;
        };
        GemmHArgs g{};
        g.Ahi = (half_t*)dev_rand(blk_plane_elems(M, K) * 2, 0x3f3f); g.Alo = g.Ahi;
        g.Whi = (half_t*)dev_rand(blk_plane_elems(N, K) * 2, 0x3f3f); g.Wlo = g.Whi;
        g.W8 = (unsigned char*)dev_rand((size_t)N * K, 0x3f3f);
        std::vector<float> hb(N, 0.1f); float* bias; hipMalloc(&bias, N * 4); hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice);
        g.bias = bias; g.M = M; g.N = N; g.K = K; g.x2 = 1;
        half_t *chi, *clo; hipMalloc(&chi, blk_plane_elems(M, N) * 2); hipMalloc(&clo, blk_plane_elems(M, N) * 2);
        g.Chi = chi; g.Clo = clo;
        hipMalloc(&g.C, 4096);
        hipMalloc(&g.range_flag, 4); hipMemset(g.range_flag, 0, 4);
        const double fl = 2.0 * M * N * K;
        for (int rep = 0; rep < 2; ++rep) {
            printf("N %d: full %.1f us (%.0f TF) | no epilogue %.1f | no fp16 MFMA %.1f | no fp8 MFMA %.1f | no MFMA at all %.1f | no MFMA, no epilogue %.1f | no DMA %.1f | no DMA, no epilogue %.1f | "
                   "no bf8 packing %.1f | no barriers %.1f | MFMA only (no DMA, barriers, epilogue) %.1f\n",
                   N, run<0>(g, 40), fl / run<0>(g, 40) / 1e6, run<8>(g, 40), run<1>(g, 40), run<2>(g, 40), run<3>(g, 40), run<11>(g, 40), run<4>(g, 40), run<12>(g, 40),
                   run<16>(g, 40), run<32>(g, 40), run<4 + 8 + 32>(g, 40));
        }
    }
    return 0;
}
