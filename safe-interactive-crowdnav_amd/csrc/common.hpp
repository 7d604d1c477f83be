// Common device/host helpers for libjmid_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace jmid {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;

// Layout of the per-(episode,agent) ConcatSquash "hyper" vector and of the per-step time table:
// [gate1 | bias1 | gate3 | bias3 | gate4 | bias4 | gateO | biasO]   (MID/models/common.py:58-72)
struct HyperLayout {
    int g1, b1, g3, b3, g4, b4, go, bo, total;
};
__host__ __device__ inline HyperLayout make_hyper_layout(int d_model, int d_mid, int d_low) {
    HyperLayout L;
    L.g1 = 0;
    L.b1 = L.g1 + d_model;
    L.g3 = L.b1 + d_model;
    L.b3 = L.g3 + d_mid;
    L.g4 = L.b3 + d_mid;
    L.b4 = L.g4 + d_low;
    L.go = L.b4 + d_low;
    L.bo = L.go + 2;
    L.total = L.bo + 2;
    return L;
}

// token m -> (episode, agent) row of the hyper buffer.  rows: r = (e*K + s)*A + a ; m = r*T + t
struct RowMap {
    int T, A, KA;  // KA = K*A
    __device__ __forceinline__ int ea(int m) const {
        int r = m / T;
        int e = r / KA;
        int a = r % A;
        return e * A + a;
    }
};

// Blocked ("panel") layout of an fp16 operand plane [rows, K] of the split-fp16 GEMMs: 128-row x 32-half tiles of
// 8 KB stored contiguously, tile (rb, kb) at ((rb * K/32 + kb) * 4096) halfs.  Inside a tile, row r is a 64-byte line
// whose four 16-byte chunks are XOR-swizzled (chunk c at c ^ ((r>>2)&3)): the memory image IS the bank-conflict-free
// LDS image, so one global_load_lds wave-instruction moves 1 KB of fully contiguous, fully used cache lines.
// Rows are padded to a multiple of 128 (padding rows only ever feed discarded output rows).
__host__ __device__ __forceinline__ size_t blk_index(int row, int k, int K) {
    const int rb = row >> 7, r = row & 127, kb = k >> 5, kk = k & 31;
    return ((size_t)rb * (K >> 5) + kb) * 4096 + (size_t)(r * 32 + ((((kk >> 3) ^ ((r >> 2) & 3)) << 3) | (kk & 7)));
}
__host__ __device__ __forceinline__ size_t blk_plane_elems(size_t rows, int K) {
    return ((rows + 127) / 128) * 128 * (size_t)K;
}

// V^T planes [sequence][head][head-dim row][Spad] are key-contiguous, Spad = S rounded up to 16 keys.  Inside every
// aligned group of 16 keys the four 4-key granules are stored in the order 0, 2, 1, 3: each half of the group then is
// exactly the 8 keys a lane-half feeds to one PV MFMA ({4hi..4hi+3, 8+4hi..8+4hi+3}: the S^T accumulator layout), i.e.
// ONE 16-byte LDS read per fragment.  The map is its own inverse.
__host__ __device__ __forceinline__ int vt_key_pos(int key) {
    const int g = (key >> 2) & 3;
    return (key & ~15) | ((((g & 1) << 1) | (g >> 1)) << 2) | (key & 3);
}
__host__ __device__ __forceinline__ int vt_spad(int S) { return (S + 15) / 16 * 16; }

// Dynamic LDS the small row-wise kernels request although they use none (tuning knob "bystander_lds").  With a
// request above 96 KB such a workgroup cannot share a CU with an attention or GEMM workgroup of another chunk lane.
static int g_bystander_lds = 0;
template <typename F>
static inline int bystander_lds(F* fn) {
    if (g_bystander_lds > 0) (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, g_bystander_lds);
    return g_bystander_lds;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// C/D fragment row of a 32x32 MFMA accumulator register (col = lane & 31)
__device__ __forceinline__ int frag_row(int reg, int hi) { return (reg & 3) + 8 * (reg >> 2) + 4 * hi; }

}  // namespace jmid
