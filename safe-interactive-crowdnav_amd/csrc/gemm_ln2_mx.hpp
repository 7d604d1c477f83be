// JMID_PREC_F16MX, d_model 512: the row-complete GEMM + residual + LayerNorm of gemm_ln_f16x3.hpp, second generation.
//     X <- LayerNorm(X + A . W^T + b) * gamma + beta          (attention out-projection + norm1, linear2 + norm2;
//                                                              nn.TransformerEncoderLayer as built at MID/models/diffusion.py:161-166)
// What the first generation (gemm_ln_mx_kernel) lost: 55 us of a 101 / 145 us launch were its epilogue - every CU reading and
// writing its residual rows (4 bytes per element each way) in the same phase while HBM idled through the K loops, one
// workgroup per CU (the fp32 [64, 520] epilogue tile alone is 133 KB of LDS).  Here:
//   * the product is computed TRANSPOSED (W fragments as the first MFMA operand, their rows permuted so that a lane ends up with
//     runs of 8 consecutive columns of ONE token row): residual add, row statistics and the normalisation happen in the
//     accumulator registers - no epilogue tile; the statistics need one cross-lane and one cross-wave (1 KB of LDS) step;
//   * the residual stream's lo plane is a BYTE plane in this mode (bf8 of fp16(x - hi): hi + lo carries x to ~14 bits, and the
//     only readers of X_lo in F16X2 / F16MX are these LayerNorms - the GEMMs take X_hi; ADE against exact fp32 1.163e-5 vs
//     1.158e-5 m): 3 instead of 4 bytes per element each way, as 16-byte (hi) and 8-byte (lo) accesses per lane;
//   * bf8(W_lo) goes L2 -> registers directly (it is wave-private: staging it in LDS bought nothing), so the rings are 60 KB and
//     TWO workgroups of 4 waves (wave tile 64 rows x 128 columns = 128 accumulators) share a CU: one's epilogue runs under the
//     other's K loop.
// The row statistics are summed in a fixed order that add_ln2_kernel (the unfused path for launches too small to fill the chip)
// reproduces, so fused and unfused rows stay bit-identical and a chunk plan cannot change a result:
//     partial(w, h) = sum over (j, p, e) in that order of v[128 w + 32 j + 16 p + 8 h + e]        w = 0..3, h = 0..1
//     total = (((P0 + P1) + P2) + P3),  Pw = partial(w, 0) + partial(w, 1)
#pragma once
#include "gemm_ln_f16x3.hpp"

namespace jmid {

// byte plane [rows, K] in 128-row x 32-column tiles like the fp16 panels (common.hpp::blk_index), rows of a tile 32 bytes apart
__host__ __device__ __forceinline__ size_t blk8_index(int row, int k, int K) {
    const int rb = row >> 7, r = row & 127, kb = k >> 5, kk = k & 31;
    return (((size_t)rb * (K >> 5) + kb) * 128 + r) * 32 + kk;
}
typedef i32x2_e i32x2;

// 8 bf8 bytes (two dwords) -> 8 floats: a bf8 value is the top byte of an fp16
__device__ __forceinline__ void f32_of_bf8x8(i32x2 b, float (&o)[8]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const unsigned d = (unsigned)b[u];
        const f16x2 p01 = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(0u, d, 0x010c000cu));
        const f16x2 p23 = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(0u, d, 0x030c020cu));
        o[4 * u + 0] = (float)p01[0];
        o[4 * u + 1] = (float)p01[1];
        o[4 * u + 2] = (float)p23[0];
        o[4 * u + 3] = (float)p23[1];
    }
}
// 8 fp16 lo values -> their bf8 images (two dwords)
__device__ __forceinline__ i32x2 bf8x8_of_f16(const f16x8& l) {
    const i32x4 d = __builtin_bit_cast(i32x4, l);
    i32x2 r;
    r[0] = bf8_of_f16x4(d[0], d[1]);
    r[1] = bf8_of_f16x4(d[2], d[3]);
    return r;
}

struct GemmLn2Args {
    const half_t* Ahi;            // [M, K] fp16 plane, blocked panel layout
    const half_t* W16hi;          // [K/16][512][16] hi plane of the 2^8-scaled weight
    const unsigned char* W8;      // bf8(W_lo) in MFMA-fragment order (gemm_f16x3.hpp::w8_image_kernel)
    const float *bias, *gamma, *beta;
    half_t* Xh;                   // residual stream [M, 512]: fp16 hi plane (blocked) ...
    unsigned char* Xl8;           // ... and the bf8 image of its lo plane (blk8_index): read, then overwritten with the result
    int M, K;
    float eps;
    int* range_flag;
    int no_lo_out;                // the last LayerNorm of the net: nobody reads its lo plane
};

constexpr int GL2_BM = 64;
constexpr int GL2_NSW = 3, GL2_NSA = 3;
constexpr int GL2_W_STAGE = GLN_BN * 16;                 // halfs: the hi plane of a k16 slice, 16 KB
constexpr int GL2_A_STAGE = GL2_BM * 32;                 // halfs: a k32 tile of 64 rows, 4 KB
constexpr int GL2_A_OFF = GL2_NSW * GL2_W_STAGE;         // halfs
constexpr size_t GL2_RING_BYTES = size_t(GL2_A_OFF + GL2_NSA * GL2_A_STAGE) * sizeof(half_t);    // 48 + 12 = 60 KB
constexpr size_t GL2_LDS_BYTES = GL2_RING_BYTES + 3 * GLN_BN * sizeof(float);                    // + bias, gamma, beta: 66 KB

// the weight row (column of the output) MFMA row `r` of a 32-row fragment carries: bits 2 and 3 of r swapped, so that the
// accumulator registers 8p .. 8p+7 of a lane are the 8 consecutive columns 16 p + 8 hi + 0..7 of its token row
__device__ __forceinline__ int gl2_frag_row(int r) { return (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1); }

__global__ __launch_bounds__(256, 2) void gemm_ln2_mx_kernel(GemmLn2Args g) {
    constexpr int WM = 2, WN = 4, d = GLN_BN;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half_t* lds = reinterpret_cast<half_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = b % 8;
    const int tm = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const int m0 = tm * GL2_BM;
    const int nk = g.K / 32, nsteps = g.K / 16;
    // bias, gamma, beta of all 512 columns into LDS (6 KB): the epilogue reads them with LDS latency.  These are ordinary VMEM
    // loads: the ds_writes retire them before the DMA ring starts, so that vmcnt counts only the ring (+ the bf8(W_lo) loads).
    float* par = reinterpret_cast<float*>(lds_raw + GL2_RING_BYTES);
    {
        const f32x4 pb = *reinterpret_cast<const f32x4*>((tid < 128 ? g.bias : g.gamma) + (tid & 127) * 4);
        const f32x4 pc = tid < 128 ? *reinterpret_cast<const f32x4*>(g.beta + tid * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(par + (tid < 128 ? 0 : GLN_BN) + (tid & 127) * 4) = pb;
        if (tid < 128) *reinterpret_cast<f32x4*>(par + 2 * GLN_BN + tid * 4) = pc;
    }

    auto dma16 = [](const void* s, void* dd) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s, (__attribute__((address_space(3))) void*)dd, 16, 0, 0);
    };
    // A: the 4 KB half of a panel image that holds this tile's 64 rows; one wave-instruction per wave and k32 tile.  Past the
    // end the last tile / slice is copied again into its own stage (identical bytes: harmless), which keeps the number of
    // DMA instructions in flight at every wait a compile-time constant.
    const half_t* a_src = g.Ahi + (size_t)(tm >> 1) * nk * 4096 + (tm & 1) * 2048 + tid * 8;
    auto issueA = [&](int ka) {
        const int kk = ka < nk ? ka : nk - 1;
        dma16(a_src + (size_t)kk * 4096, lds + GL2_A_OFF + (kk % GL2_NSA) * GL2_A_STAGE + wc * 512);
    };
    // W_hi: the wave's own 128 rows of a k16 slice (4 KB, four wave-instructions): wave-private, no workgroup barrier
    const half_t* w_src = g.W16hi + (size_t)(wc * 128) * 16 + lane * 8;
    auto issueW = [&](int s) {
        const int ss = s < nsteps ? s : nsteps - 1;
        half_t* st = lds + (ss % GL2_NSW) * GL2_W_STAGE + wc * 2048;
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16(w_src + ((size_t)ss * GLN_BN + q * 32) * 16, st + q * 512);
    };
    // bf8(W_lo) of a k64 block: this wave's four 32-column blocks, two 16-byte pieces each, straight into registers
    const int frow = gl2_frag_row(l31);
    const unsigned char* w8src = g.W8 + (size_t)(wc * 4) * 2048 + (size_t)(frow + 32 * hi) * 16;
    const size_t w8_kstride = (size_t)(GLN_BN / 32) * 2048;
    i32x8 w8[WN];
    auto loadW8 = [&](int kb) {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const unsigned char* p = w8src + (size_t)kb * w8_kstride + j * 2048;
            const i32x4 lo = *reinterpret_cast<const i32x4*>(p);
            const i32x4 up = *reinterpret_cast<const i32x4*>(p + 1024);
            w8[j] = i32x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
        }
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
    for (int j = 0; j < WN; ++j) offW[j] = (wc * 128 + j * 32 + frow) * 16 + hi * 8;
    i32x8 a8[WM];

    // VMEM issue order: A0 W0 A1 W1 | step s: W(s+2) [A(s/2+2) on even s] [the 8 loads of bf8(W_lo) at s % 4 == 0].
    // Younger than W(s) at the top of step s:  s%4 == 0: A, W = 5;  1: W, A, the 8 loads = 13;  3: W, A = 5;  2: A, W = 5 plus the 8
    // loads IF they went out after W(s) - the compiler may order them either way inside step s - 2, so that wait takes them along.
    issueA(0);
    issueW(0);
    issueA(1);
    issueW(1);
    auto step = [&](const int s, auto q_c) {
        constexpr int Q = decltype(q_c)::value, ks = Q & 1;
        if (Q == 1) wait_vmcnt<13>();
        else wait_vmcnt<5>();
        if (ks == 0) __builtin_amdgcn_s_barrier();      // A tile s/2 landed for everybody; A stage (s/2 - 1) % 3 is free again
        __builtin_amdgcn_sched_barrier(0);
        issueW(s + 2);
        if (ks == 0) issueA((s >> 1) + 2);
        if (Q == 0) loadW8(s >> 2);
        const half_t* stA = lds + GL2_A_OFF + ((s >> 1) % GL2_NSA) * GL2_A_STAGE;
        const half_t* stW = lds + (s % GL2_NSW) * GL2_W_STAGE;
        f16x8 ah[WM], wh[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) ah[i] = *reinterpret_cast<const f16x8*>(stA + offA[i][ks]);
#pragma unroll
        for (int j = 0; j < WN; ++j) wh[j] = *reinterpret_cast<const f16x8*>(stW + offW[j]);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const i32x4 dw = __builtin_bit_cast(i32x4, ah[i]);
            a8[i][Q * 2 + 0] = bf8_of_f16x4(dw[0], dw[1]);
            a8[i][Q * 2 + 1] = bf8_of_f16x4(dw[2], dw[3]);
        }
        if (Q == 3) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)       // both operands bf8, literal zero scales: the UNSCALED instruction (gemm_f16x3.hpp)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8[j], a8[i], acc[i][j], 1, 1, 0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < nsteps; s += 4) {
        step(s, std::integral_constant<int, 0>{});
        step(s + 1, std::integral_constant<int, 1>{});
        step(s + 2, std::integral_constant<int, 2>{});
        step(s + 3, std::integral_constant<int, 3>{});
    }
    wait_vmcnt<0>();                     // the copies past the end
    __builtin_amdgcn_s_barrier();        // everybody is done with the rings: their first 2 KB become the reduction scratch
    float* red = reinterpret_cast<float*>(lds_raw);      // [2 passes][4 waves][64 rows]

    // ---- epilogue, in the accumulators: register 8p + e of acc[i][j] is column 128 wc + 32 j + 16 p + 8 hi + e of row 32 i + l31.
    // The residual chunks of a row come in two batches of eight (16 + 8 bytes each) requested together: two memory round trips per
    // row instead of sixteen; the second workgroup of the CU computes meanwhile.
    float mean[WM], rstd[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int row = m0 + i * 32 + l31;      // rows past M exist in the padded planes: loads need no guard
        float s = 0.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f16x8 xh[8];
            i32x2 xb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c0 = wc * 128 + (half * 2 + (u >> 1)) * 32 + (u & 1) * 16 + hi * 8;
                xh[u] = *reinterpret_cast<const f16x8*>(g.Xh + blk_index(row, c0, d));
                xb[u] = *reinterpret_cast<const i32x2*>(g.Xl8 + blk8_index(row, c0, d));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = half * 2 + (u >> 1), p = u & 1;
                const int c0 = wc * 128 + j * 32 + p * 16 + hi * 8;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(par + c0), b1 = *reinterpret_cast<const f32x4*>(par + c0 + 4);
                float xl[8];
                f32_of_bf8x8(xb[u], xl);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = fmaf(acc[i][j][8 * p + e], kWInv, e < 4 ? b0[e] : b1[e - 4]);
                    const float a = (float)xh[u][e] + xl[e];
                    const float v = a + y;
                    acc[i][j][8 * p + e] = v;
                    s += v;
                }
            }
        }
        s += __shfl_xor(s, 32, 64);
        if (hi == 0) red[wc * 64 + i * 32 + l31] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int r = i * 32 + l31;
        mean[i] = (((red[r] + red[64 + r]) + red[128 + r]) + red[192 + r]) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float t = acc[i][j][e] - mean[i];
                q += t * t;
            }
        q += __shfl_xor(q, 32, 64);
        if (hi == 0) red[256 + wc * 64 + r] = q;
    }
    __syncthreads();
    bool overflow = false;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int r = i * 32 + l31, row = m0 + r;
        rstd[i] = rsqrtf((((red[256 + r] + red[320 + r]) + red[384 + r]) + red[448 + r]) / (float)d + g.eps);
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int c0 = wc * 128 + j * 32 + p * 16 + hi * 8;
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(par + GLN_BN + c0), g1 = *reinterpret_cast<const f32x4*>(par + GLN_BN + c0 + 4);
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(par + 2 * GLN_BN + c0), t1 = *reinterpret_cast<const f32x4*>(par + 2 * GLN_BN + c0 + 4);
                f16x8 vh, vl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float o = (acc[i][j][8 * p + e] - mean[i]) * rstd[i] * (e < 4 ? g0[e] : g1[e - 4]) + (e < 4 ? t0[e] : t1[e - 4]);
                    half_t hh, ll;
                    split_f32(o, hh, ll);
                    overflow |= !(fabsf(o) <= kHalfMax);
                    vh[e] = hh;
                    vl[e] = ll;
                }
                if (row < g.M) {
                    *reinterpret_cast<f16x8*>(g.Xh + blk_index(row, c0, d)) = vh;
                    if (!g.no_lo_out) *reinterpret_cast<i32x2*>(g.Xl8 + blk8_index(row, c0, d)) = bf8x8_of_f16(vl);
                }
            }
    }
    if (overflow) atomicOr(g.range_flag, 1);
}

inline hipError_t launch_gemm_ln2_mx(const GemmLn2Args& g, hipStream_t st) {
    static DevSeen seen;
    if (auto once_ = first_use_on_device(seen))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ln2_mx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)GL2_LDS_BYTES);
    const int ntm = (g.M + GL2_BM - 1) / GL2_BM;
    hipLaunchKernelGGL(gemm_ln2_mx_kernel, dim3(ntm), dim3(256), GL2_LDS_BYTES, st, g);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// The unfused partner (launches too small to fill the chip with 64-row tiles): X <- LN(X + Y) for the same planes, with the
// row statistics summed in gemm_ln2_mx_kernel's order - lane (row, w, h) of a wave of 8 rows owns the 64 columns of partial(w, h).
__global__ __launch_bounds__(256) void add_ln2_kernel(const float* Y, const float* gamma, const float* beta, int M, float eps,
                                                      half_t* Xh, unsigned char* Xl8, int no_lo_out, int* range_flag) {
    constexpr int d = GLN_BN;
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (lane >> 3);
    const int w = (lane >> 1) & 3, hi = lane & 1;
    const int rowc = row < M ? row : M - 1;
    float v[64];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int c0 = w * 128 + j * 32 + p * 16 + hi * 8;
            const f16x8 xh = *reinterpret_cast<const f16x8*>(Xh + blk_index(rowc, c0, d));
            const i32x2 xb = *reinterpret_cast<const i32x2*>(Xl8 + blk8_index(rowc, c0, d));
            const f32x4 y0 = *reinterpret_cast<const f32x4*>(Y + (size_t)rowc * d + c0), y1 = *reinterpret_cast<const f32x4*>(Y + (size_t)rowc * d + c0 + 4);
            float xl[8];
            f32_of_bf8x8(xb, xl);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = (float)xh[e] + xl[e];
                const float t = a + (e < 4 ? y0[e] : y1[e - 4]);
                v[(j * 2 + p) * 8 + e] = t;
                s += t;
            }
        }
    const int base = lane & ~7;
    s += __shfl_xor(s, 1, 64);
    const float mean = (((__shfl(s, base, 64) + __shfl(s, base + 2, 64)) + __shfl(s, base + 4, 64)) + __shfl(s, base + 6, 64)) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 64; ++e) {
        const float t = v[e] - mean;
        q += t * t;
    }
    q += __shfl_xor(q, 1, 64);
    const float rstd = rsqrtf((((__shfl(q, base, 64) + __shfl(q, base + 2, 64)) + __shfl(q, base + 4, 64)) + __shfl(q, base + 6, 64)) / (float)d + eps);
    bool overflow = false;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int c0 = w * 128 + j * 32 + p * 16 + hi * 8;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c0), g1 = *reinterpret_cast<const f32x4*>(gamma + c0 + 4);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(beta + c0), t1 = *reinterpret_cast<const f32x4*>(beta + c0 + 4);
            f16x8 vh, vl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float o = (v[(j * 2 + p) * 8 + e] - mean) * rstd * (e < 4 ? g0[e] : g1[e - 4]) + (e < 4 ? t0[e] : t1[e - 4]);
                half_t hh, ll;
                split_f32(o, hh, ll);
                overflow |= !(fabsf(o) <= kHalfMax);
                vh[e] = hh;
                vl[e] = ll;
            }
            if (row < M) {
                *reinterpret_cast<f16x8*>(Xh + blk_index(row, c0, d)) = vh;
                if (!no_lo_out) *reinterpret_cast<i32x2*>(Xl8 + blk8_index(row, c0, d)) = bf8x8_of_f16(vl);
            }
        }
    if (overflow && row < M) atomicOr(range_flag, 1);
}

}  // namespace jmid
