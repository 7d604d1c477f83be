// Planning probe for docs/NOTEBOOK.md section 7 item 1: what would the 256x256 split-fp16 GEMM gain if the W_lo correction term
// ran on the block-scaled FP8 matrix path?  TIMING ONLY (operands are random bytes, results are meaningless): the K loop of
// gemm_f16x3_dma256x256_kernel<.., X2 = true> with the same DMA bytes per k64 (A_hi, W_hi as today; an fp8 image of A_hi
// and an fp8 image of W_lo are one 8 KB panel per k64 each - issued every other k32 tile in place of the W_lo / A_lo
// copies), the same barriers and the same epilogue, in two modes:
//   0  today:  per k16 step and output tile  A_hi x W_hi  +  A_hi x W_lo          (2 x v_mfma_f32_32x32x16_f16)
//   1  MX:     per k16 step A_hi x W_hi, per k64 ONE v_mfma_scale_f32_32x32x64_f8f6f4 (A_hi8 x W_lo8) per output tile
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -I safe-interactive-crowdnav_amd/csrc tools/mx_gemm_probe.hip -o build/mx_gemm_probe
#include "gemm_f16x3.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace jmid;
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe_kernel(GemmHArgs g, int ntm, int ntn) {
    constexpr int WM = 2, WN = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wid >> 1, wc = wid & 1;
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const int tm = swz / ntn, tn = swz - tm * ntn;
    const int m0 = tm * 256, n0 = tn * 256;
    const int nk = g.K / 32;
    const half_t* src[8];
    src[0] = g.Ahi + (size_t)(2 * tm) * nk * 4096 + tid * 8;
    src[1] = g.Ahi + (size_t)(2 * tm + 1) * nk * 4096 + tid * 8;
    src[2] = g.Alo + (size_t)(2 * tm) * nk * 4096 + tid * 8;
    src[3] = g.Alo + (size_t)(2 * tm + 1) * nk * 4096 + tid * 8;
    src[4] = g.Whi + (size_t)(2 * tn) * nk * 4096 + tid * 8;
    src[5] = g.Whi + (size_t)(2 * tn + 1) * nk * 4096 + tid * 8;
    src[6] = g.Wlo + (size_t)(2 * tn) * nk * 4096 + tid * 8;
    src[7] = g.Wlo + (size_t)(2 * tn + 1) * nk * 4096 + tid * 8;
    auto issue_one = [&](int kt, int i) {
        half_t* st = lds + (kt & 1) * DMA3_STAGE + wid * 512;
        if (MODE == 0 && (i == 2 || i == 3)) return;                    // today's F16X2: no A_lo image
        if (MODE == 1 && (i == 2 || i == 3 || i == 6 || i == 7) && (kt & 1)) return;   // fp8 images: one panel per k64
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)kt * 4096),
                                         (__attribute__((address_space(3))) void*)(st + i * 4096), 16, 0, 0);
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
        for (int ks = 0; ks < 2; ++ks) offW[j][ks] = (row >> 7) * DMA_PLANE + r * 32 + (((ks * 2 + hi) ^ ((r >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_one(0, i);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
#pragma unroll
            for (int i = 0; i < 8; ++i) issue_one(kt + 1, i);
        }
        const half_t* st = lds + (kt & 1) * DMA3_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah[WM], wh[WN], wl[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) ah[i] = *reinterpret_cast<const f16x8*>(st + offA[i][ks]);
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                wh[j] = *reinterpret_cast<const f16x8*>(st + 4 * DMA_PLANE + offW[j][ks]);
                if (MODE == 0) wl[j] = *reinterpret_cast<const f16x8*>(st + 6 * DMA_PLANE + offW[j][ks]);
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc[i][j], 0, 0, 0);
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl[j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1 && (kt & 1)) {
            // one block-scaled fp8 MFMA per output tile and k64: 32 bytes per lane of each operand = two 16-byte LDS reads
            i32x8 a8[WM], w8[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const i32x4 lo = *reinterpret_cast<const i32x4*>(st + 2 * DMA_PLANE + offA[i][0]);
                const i32x4 up = *reinterpret_cast<const i32x4*>(st + 2 * DMA_PLANE + offA[i][1]);
                a8[i] = i32x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const i32x4 lo = *reinterpret_cast<const i32x4*>(st + 6 * DMA_PLANE + offW[j][0]);
                const i32x4 up = *reinterpret_cast<const i32x4*>(st + 6 * DMA_PLANE + offW[j][1]);
                w8[j] = i32x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], w8[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0,
                                                                                0x73737373);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    gemm_h_epilogue<WM, WN, EPI_BIAS_RELU, OUT_SPLIT, true>(g, acc, m0, n0, wr, wc, l31, hi, 256, 256);
}

template <int MODE>
float run(GemmHArgs g, int reps) {
    const int ntm = (g.M + 255) / 256, ntn = g.N / 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DMA3_LDS_BYTES);
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(probe_kernel<MODE>, dim3(ntm * ntn), dim3(512), DMA3_LDS_BYTES, 0, g, ntm, ntn);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe_kernel<MODE>, dim3(ntm * ntn), dim3(512), DMA3_LDS_BYTES, 0, g, ntm, ntn);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const int M = 61440, K = 512;
    for (int N : {1024, 1536}) {
        auto dev_rand = [&](size_t halfs, int mask) {
            std::vector<unsigned short> h(halfs);
            for (size_t i = 0; i < halfs; ++i) h[i] = (unsigned short)(rand() & mask);
            half_t* p; hipMalloc(&p, halfs * 2); hipMemcpy(p, h.data(), halfs * 2, hipMemcpyHostToDevice); return p;
        };
        GemmHArgs g{};
        // fp16 patterns with |x| < 2 (exponent top bit clear); the same bytes read as fp8 pairs have their top exponent bits clear too
        g.Ahi = dev_rand(blk_plane_elems(M, K), 0x3f3f); g.Alo = dev_rand(blk_plane_elems(M, K), 0x3f3f);
        g.Whi = dev_rand(blk_plane_elems(N, K), 0x3f3f); g.Wlo = dev_rand(blk_plane_elems(N, K), 0x3f3f);
        std::vector<float> hb(N, 0.1f); float* bias; hipMalloc(&bias, N * 4); hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice);
        g.bias = bias; g.M = M; g.N = N; g.K = K; g.x2 = 1;
        half_t *chi, *clo; hipMalloc(&chi, blk_plane_elems(M, N) * 2); hipMalloc(&clo, blk_plane_elems(M, N) * 2);
        g.Chi = chi; g.Clo = clo;
        hipMalloc(&g.range_flag, 4); hipMemset(g.range_flag, 0, 4);
        for (int rep = 0; rep < 2; ++rep) {
            const float t0 = run<0>(g, 60), t1 = run<1>(g, 60);
            const double fl = 2.0 * M * N * K;
            printf("M %d N %d K %d : today (2 fp16 passes) %.1f us = %.0f TFLOP/s | hi x hi fp16 + MX-fp8 correction %.1f us = %.0f TFLOP/s | %.1f %% less time\n",
                   M, N, K, t0 * 1e3, fl / t0 / 1e9, t1 * 1e3, fl / t1 / 1e9, 100.0 * (1.0 - t1 / t0));
        }
    }
    return 0;
}
