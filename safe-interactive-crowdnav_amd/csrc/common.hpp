// Common device/host helpers for libjmid_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

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

// x / d for 0 <= x, x * d < 2^32, with m = 2^32 / d + 1 made on the host (fast_div_magic; 0: plain division)
__device__ __forceinline__ int fast_div(int x, int d, unsigned m) {
    return m ? (int)__umulhi((unsigned)x, m) : (d == 1 ? x : x / d);
}
inline unsigned fast_div_magic(int d, unsigned long long x_max) {
    return d > 1 && x_max * (unsigned long long)d < (1ull << 32) ? (unsigned)((1ull << 32) / (unsigned)d + 1) : 0u;
}

// token m -> (episode, agent) row of the hyper buffer.  rows: r = (e*K + s)*A + a ; m = r*T + t
struct RowMap {
    int T, A, KA;  // KA = K*A
    unsigned mT = 0, mKA = 0, mA = 0;     // reciprocals (make_rowmap); 0: divide
    __device__ __forceinline__ int ea(int m) const {
        const int r = fast_div(m, T, mT);
        const int e = fast_div(r, KA, mKA);
        const int a = r - fast_div(r, A, mA) * A;
        return e * A + a;
    }
    __device__ __forceinline__ int t_of(int m) const { return m - fast_div(m, T, mT) * T; }
};
inline RowMap make_rowmap(int T, int A, int KA, unsigned long long M) {
    RowMap r{T, A, KA};
    r.mT = fast_div_magic(T, M);
    r.mKA = fast_div_magic(KA, M);
    r.mA = fast_div_magic(A, M);
    return r;
}

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

// A kernel's argument block, requested in ONE batch at entry.  Left to itself hipcc sinks the scalar loads of fields that are used
// late or conditionally next to their uses: the attention kernel waited for its arguments three times in a row, each a cold miss
// (the scalar cache is invalidated at every kernel boundary and the block was just written by the host) of ~0.7 us - in a
// one-scene launch that lives 10 us.  args_now(a) makes every dword of the block an input of an empty asm statement at the top.
template <typename T>
__device__ __forceinline__ void args_now(const T& a) {
    static_assert(sizeof(T) % 4 == 0, "argument block of whole dwords");
    struct Words { unsigned w[sizeof(T) / 4]; };
    const Words w = __builtin_bit_cast(Words, a);
    // inputs only: the values the kernel goes on to use are the original ones (a pointer that went THROUGH the asm would lose its
    // address space: flat loads with vmcnt + lgkmcnt waits instead of global loads)
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) asm volatile("" ::"s"(w.w[i]));
}
template <typename... T>
__device__ __forceinline__ void args_now_each(const T&... t) { (args_now(t), ...); }

// A wave-uniform 64-bit value (an address) pinned in a scalar register pair.  LDS-DMA copies and epilogue accesses address memory as
// scalar base + 32-bit lane offset; left alone, hipcc re-associates base + offset into a per-lane 64-bit vector address (a 64-bit vector
// add per access, two registers per pointer kept).  Through an integer: a pointer that passes an asm operand comes back generic (flat).
// The value must be PROVABLY uniform (kernel arguments, blockIdx, readfirstlane results): anything else fails to compile ("illegal
// VGPR to SGPR copy"), which is the check one wants.  pin_uniform_rfl: the same behind explicit v_readfirstlane, for values the
// compiler cannot prove uniform (they cost vector registers: the F16X3 256 x 256 GEMM spilled 17 with it).
__device__ __forceinline__ unsigned long long pin_uniform(unsigned long long v) {
    asm volatile("" : "+s"(v));
    return v;
}
__device__ __forceinline__ unsigned long long pin_uniform_rfl(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    unsigned long long u = ((unsigned long long)hi << 32) | lo;
    asm volatile("" : "+s"(u));
    return u;
}

// Tuning knobs (jmid_set_tuning).  They belong to a handle: every entry point of the C ABI installs its handle's
// set for the duration of the call (TuneScope, thread-local), the launch helpers read it through tune().
struct Tuning {
    int fuse_embed = 1;      // the output kernel of step i embeds x for step i + 1
    int ln_fuse = 0;         // 0 auto, 1 always, 2 never: fused GEMM + residual + LayerNorm
    int ln_rows = 0;         // row tile of the fused GEMM + LayerNorm: 0 auto, 64, 128
    int bystander_lds = 0;   // dynamic LDS the small row-wise kernels request although they use none
    int gemm_h_variant = 0;  // 0 auto, 1..6 force a tile variant of the split-fp16 GEMM (diagnostics: 7 = 128 x 256 two per CU, 8 = 256 x 256 with a three-stage ring)
    int gemm_ng = 0;         // N-tiles per L2 group of the 256x128 GEMM (0 = auto)
    int no_vt_direct = 0;    // 1: V row-major + v_transpose_kernel even when the fused V^T epilogue applies
    int attn_h_variant = 0;  // 0 auto (DMA when head_dim == 128), 1 = register-staged, 2 = DMA
    int attn_pack = 1;       // 0 = one short sequence per wave even when S <= 16
    int vt_stage = 0;        // 256x256 QKV kernel: V^T through LDS in full rows (0 / 1 on, 2 = direct 8-byte stores)
    int graph = 0;           // captured denoise loop (hipGraph) of one-chunk calls: 1 on, 0 / 2 off (default: not faster, see jmid_planner.hip)
    int attn_nsplit = 0;     // split-KV factor of the head_dim-128 attention launches: 0 auto, 1..16 forced
    int csl_swap = 0;        // F16MX: 0 = transposed product + row-wise epilogue for the ConcatSquash GEMMs, 3 = for linear1 too (slower), 2 = neither
    int attn_sm = 0;         // head_dim-128 LDS-DMA attention: 0 = the round-6 softmax (reference maximum through the accumulator, no tile maximum on the common path, row sum from packed P), 2 = the round 2-5 form (diagnostics A/B; other bits, same softmax)
    int attn_prio = 0;       // head_dim-128 LDS-DMA attention: static s_setprio 1 for half of the workgroups (AttnHArgs::prio): 0 off, 1 / 2 on
    int attn_one_wg = 0;     // (probe) 1 = the head_dim-128 LDS-DMA attention kernels request the CU's whole 160 KB of LDS: ONE workgroup per CU, one wave per SIMD
    int h1_stage = 0;        // F16MX linear1 in the 256 x 256 / 128 x 256 shapes: 0 / 1 = tile out through LDS in whole lines (h1_staged_store), 2 = the element-wise epilogue
    int out_traj = 0;        // output layer + DDIM update + next embedding: 0 = one wave per trajectory from 4096 trajectories, 1 = always, 2 = one wave per token
    int attn_mx = 0;         // head_dim 128: F16MX 0 = bf8 logit corrections + one fp16 plane of P, 1 = P_hi + P_lo (F16X2 too), 2 = F16X2's attention, 3 = as 0 with Q_lo as an fp16 plane (A/B; same bits)
    int mx_ln = 0;           // F16MX at d_model 512: 0 = second-generation GEMM + LayerNorm (gemm_ln2_mx.hpp: byte lo plane of the residual stream, two workgroups per CU), 2 = the first generation
    int attn_pf = 0;         // F16MX / F16X2 attention with one plane of P: 2 = fragment reads one step ahead instead of three (A/B; same bits)
    int gemm_small = 0;      // launches of at most one workgroup per CU (gemm_small.hpp): 0 = the deep-ring k64 kernel, 1 = the round-3 tile shapes, 2 = the deep-ring kernel only up to one workgroup per CU
    int small_cmb = 0;       // the split-KV merge of a one-scene attention launch inside the out-projection's OUT_LNX launch (gemm_small.hpp, lnx_combine): 0 on, 2 off (attn_combine_kernel)
    int small_lnx = 0;       // out_proj / linear2 + residual + LayerNorm of a small F16MX launch in ONE kernel, the row statistics (block sums + squared deviations from the block means: gemm_ln2_mx.hpp's canonical order) exchanged ONCE between the workgroups of a row tile (gemm_small.hpp, OUT_LNX): 0 on, 2 off (GEMM + add_ln2; the same bits)
    int cus = 256;           // compute units of the handle's device (jmid_create): OUT_LNX launches only while every workgroup of the launch is resident (diagnostics: a test may lower it)
    int lnx_polls = 0;       // diagnostics: poll budget of OUT_LNX's waits (0 = SM_LNX_POLLS)
    int lnx_withhold = 0;    // diagnostics: 1 = one workgroup of an OUT_LNX launch never publishes its statistics (the give-up path under test)
    int small_lnx2 = 0;      // ... for launches of 33 ... 64 row tiles (two scenes' worth of tokens; the reference's shipped K = 100) with TWO workgroups per CU: 0 on, 2 off (GEMM + add_ln2 [+ attn_combine])
    int gemm_pn = 0;         // F16MX large-tile GEMMs: column groups of the XCD tile order (0 / 1 = N fastest over all N-tiles)
    int one_chunk = 1;       // set per call by run_network: the call is ONE chunk (OUT_LNX of gemm_small.hpp only then, whatever the lanes: a call's bits do not depend on its chunk plan)
    int small_now = 1;       // set per call by run_network: the small-launch kernels only while ONE chunk is in flight (with two lanes their
                             // one-workgroup-per-CU launches collide: 4 episodes as 2 x 2 measured 4 % slower with them)
    int small_lanes = 0;     // experiment: the small-launch kernels with several chunks in flight too: 1 = all of them, 2 = only the two-workgroups-per-CU shape
    int small_qk = 0;        // Q / K tiles of a small in_proj launch out through LDS in full rows: 0 / 1 on, 2 = the generic element-wise epilogue
    int small_pn = 0;        // its column groups per launch (two-dimensional XCD tile order): 0 = fewest Infinity-Cache bytes, 1 / 2 / 4 / 8 forced
    int attn_abl = 0;        // timing ablations (results are WRONG): only in builds with -DJMID_ABLATIONS
    int gemm_abl = 0;
};
inline const Tuning*& tuning_slot() {
    static thread_local const Tuning* p = nullptr;
    return p;
}
inline const Tuning& tune() {
    static const Tuning dflt;
    const Tuning* p = tuning_slot();
    return p ? *p : dflt;
}
struct TuneScope {
    const Tuning* prev;
    explicit TuneScope(const Tuning* t) : prev(tuning_slot()) { tuning_slot() = t; }
    ~TuneScope() { tuning_slot() = prev; }
};
#ifdef JMID_ABLATIONS
inline int attn_abl_bits() { return tune().attn_abl; }
inline int gemm_abl_bits() { return tune().gemm_abl; }
#else
inline int attn_abl_bits() { return 0; }
inline int gemm_abl_bits() { return 0; }
#endif

// One-time per-device setup (function attributes are per device).  `if (auto once = first_use_on_device(seen)) { set attributes }`:
// the body runs once per device; concurrent first users of the same kernel (two handles on two threads) wait on the lock, and
// the device is marked only AFTER the body has run, so nobody launches before the attribute is set.
struct DevSeen {
    std::atomic<bool> v[64];
    DevSeen() { for (auto& b : v) b.store(false, std::memory_order_relaxed); }
};
struct DeviceOnce {
    std::unique_lock<std::mutex> lk;
    std::atomic<bool>* slot = nullptr;
    explicit operator bool() const { return slot != nullptr; }
    DeviceOnce() = default;
    DeviceOnce(DeviceOnce&& o) noexcept : lk(std::move(o.lk)), slot(o.slot) { o.slot = nullptr; }
    ~DeviceOnce() { if (slot) slot->store(true, std::memory_order_release); }
};
inline std::mutex& device_once_mutex() {
    static std::mutex m;
    return m;
}
inline DeviceOnce first_use_on_device(DevSeen& seen) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    DeviceOnce o;
    if (seen.v[dev].load(std::memory_order_acquire)) return o;
    o.lk = std::unique_lock<std::mutex>(device_once_mutex());
    if (seen.v[dev].load(std::memory_order_acquire)) {
        o.lk.unlock();
        return o;
    }
    o.slot = &seen.v[dev];
    return o;
}

// Dynamic LDS the small row-wise kernels request although they use none (tuning knob "bystander_lds").  With a
// request above 96 KB such a workgroup cannot share a CU with an attention or GEMM workgroup of another chunk lane.
template <typename F>
static inline int bystander_lds(F* fn) {
    const int n = tune().bystander_lds;
    if (n > 0) (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, n);
    return n;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The cross-half exchange of a row statistic (lane l <-> lane l ^ 32) as ONE vector instruction: v_permlane32_swap leaves {x[l % 32]} and
// {x[l % 32 + 32]} in both halves of its two operands - no LDS round trip (ds_bpermute behind __shfl_xor) in the middle of the softmax.
// max(a, b) and a + b of the pair are what max(x, shfl_xor(x, 32)) and x + shfl_xor(x, 32) give, bit for bit (both commutative).
// (Inline assembly: hipcc folds the two results of __builtin_amdgcn_permlane32_swap into one register - max(a, b) became a and a + b
//  became a + a.  The instruction needs two wait states after a vector write of its operands.)
__device__ __forceinline__ void half_swap(float x, float& lo_half, float& hi_half) {
#ifdef JMID_ATT_SHFL      // (A/B: the ds_bpermute form)
    lo_half = x;
    hi_half = __shfl_xor(x, 32, 64);
#else
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    lo_half = a;
    hi_half = b;
#endif
}

// C/D fragment row of a 32x32 MFMA accumulator register (col = lane & 31)
__device__ __forceinline__ int frag_row(int reg, int hi) { return (reg & 3) + 8 * (reg >> 2) + 4 * hi; }

}  // namespace jmid
