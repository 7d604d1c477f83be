// How much L2 -> LDS copy traffic can a CU carry NEXT TO its matrix work?  The question behind two designs that were priced and not
// built: the fused MLP (W1 + W2 streamed once per 64-row workgroup) and the A-stationary QKV GEMM, against what the
// shipped 256 x 256 GEMM tiles draw.
//
// One 8-wave workgroup per CU runs the skeleton of a GEMM K loop: per "k64 block" every wave issues PIECES LDS-DMA copies of 1 KB
// (global_load_lds_dwordx4) from a 4 MB region that ALL workgroups share (the weights: L2-resident, as in the real kernels), waits for
// the previous block's copies, meets the barrier, reads 36 (or 72) fragments from LDS (ds_read_b128) and issues 48 v_mfma_f32_32x32x16_f16
// (1 536 matrix cycles per wave: the F16MX block of 32 fp16 + 8 bf8 instructions).  Two ring stages.  Reported: the matrix rate reached and
// the copy rate per CU, for PIECES = 0 (no copies: the ceiling of this skeleton), 10 (= 80 KB per block: the shipped tile), 19 (the fused
// MLP) and 25 (A-stationary QKV); copies as a burst, spread between the matrix instructions, or issued by dedicated loader waves.
// TIMING ONLY.  (Source offsets are 32-bit and masked, bases 16-byte aligned: the first version of this probe took a 64-bit modulo per
// copy and reported its own vector instructions as a 35 GB/s ceiling.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/dma_mfma_probe.hip -o build/dma_mfma_probe && build/dma_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PIECES, bool SPREAD, int LOADERS = 0, int READS = 3>
__global__ __launch_bounds__(512, 1) void probe(const char* src, size_t region, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // LDS: two stages of 8 waves x PIECES KB (PIECES <= 9 per stage half when > 9: the ring wraps inside the 160 KB - copies may land on
    // top of each other, which is irrelevant for timing)
    constexpr int STG_BYTES = 80 * 1024;
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * ((lane * 7 + e) % 13)); b[e] = (_Float16)(0.002f * ((lane * 5 + e) % 11)); }
    const unsigned rmask = (unsigned)region - 1u;
    unsigned off = ((unsigned)blockIdx.x * 8192u * 5u) & rmask;      // workgroups walk the shared region from different (16-byte aligned) starts
    // LOADERS > 0: the last LOADERS waves of the workgroup only copy (each 8 * PIECES / LOADERS pieces per block, a few per "step" of the
    // others), the first 8 - LOADERS only read fragments and issue matrix instructions: no wave has both copies in flight and LDS reads
    constexpr int NCOMP = 8 - LOADERS;
    const bool loader = LOADERS > 0 && wid >= NCOMP;
    constexpr int LP = LOADERS > 0 ? 8 * PIECES / LOADERS : 0;      // pieces per loader wave and block
    auto issue_piece = [&](int stage, int p) {
        const unsigned o = ((unsigned)off + (unsigned)(wid * PIECES + p) * 1024u) & rmask;      // (region: a power of two)
        char* d = lds + stage * STG_BYTES + ((wid * PIECES + p) * 1024) % STG_BYTES;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + o + lane * 16),
                                         (__attribute__((address_space(3))) void*)d, 16, 0, 0);
    };
    auto issue = [&](int stage) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) issue_piece(stage, p);
        off = (off + 8u * PIECES * 1024u) & rmask;
    };
    if (LOADERS > 0) {
        if (loader) {
            auto lpiece = [&](int stage, int p) {
                const int g = (wid - NCOMP) * LP + p;
                const unsigned o = (off + (unsigned)g * 1024u) & rmask;
                char* d = lds + stage * STG_BYTES + (g * 1024) % STG_BYTES;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + o + lane * 16),
                                                 (__attribute__((address_space(3))) void*)d, 16, 0, 0);
            };
#pragma unroll
            for (int p = 0; p < LP; ++p) lpiece(0, p);
            off = (off + 8u * PIECES * 1024u) & rmask;
            for (int it = 0; it < iters; ++it) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int p = 0; p < LP; ++p) {
                    lpiece(1 - (it & 1), p);
                    if ((p & 3) == 3) __builtin_amdgcn_s_sleep(1);      // leave issue slots to the SIMD's other wave
                }
                off = (off + 8u * PIECES * 1024u) & rmask;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        for (int it = 0; it < iters; ++it) {
            __builtin_amdgcn_s_barrier();
            const char* base = lds + (it & 1) * STG_BYTES + (lane * 16);
#pragma unroll
            for (int s = 0; s < 12; ++s) {
                const f16x8 f0 = *reinterpret_cast<const f16x8*>(base + ((s * 3 + 0) * 1024 + wid * 128) % (STG_BYTES - 1024));
                const f16x8 f1 = *reinterpret_cast<const f16x8*>(base + ((s * 3 + 1) * 1024 + wid * 128) % (STG_BYTES - 1024));
                const f16x8 f2 = *reinterpret_cast<const f16x8*>(base + ((s * 3 + 2) * 1024 + wid * 128) % (STG_BYTES - 1024));
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, f1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2, b, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, f0, acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float t = 0;
        for (int j = 0; j < 4; ++j) t += acc[j][lane & 15];
        if (t == 12345.f) sink[blockIdx.x] = t;
        return;
    }
    issue(0);
    for (int it = 0; it < iters; ++it) {
        const int stage = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!SPREAD) issue(1 - stage);      // a burst behind the barrier - or (SPREAD, what the shipped kernels do) a few per step below
        const char* base = lds + stage * STG_BYTES + (lane * 16);
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            // READS fragment reads and four matrix instructions per step: 36 reads (READS = 3: a 128 x 64 wave tile of a 256 x 256 workgroup
            // tile) or 72 (READS = 6: the 32 x 64 wave tile of a 64-row workgroup) and 48 instructions per block
            const f16x8 f0 = *reinterpret_cast<const f16x8*>(base + ((s * READS + 0) * 1024 + wid * 128) % (STG_BYTES - 1024));
            const f16x8 f1 = *reinterpret_cast<const f16x8*>(base + ((s * READS + 1) * 1024 + wid * 128) % (STG_BYTES - 1024));
            const f16x8 f2 = *reinterpret_cast<const f16x8*>(base + ((s * READS + 2) * 1024 + wid * 128) % (STG_BYTES - 1024));
            f16x8 f3 = a, f4 = b, f5 = a;
            if (READS == 6) {
                f3 = *reinterpret_cast<const f16x8*>(base + ((s * READS + 3) * 1024 + wid * 128) % (STG_BYTES - 1024));
                f4 = *reinterpret_cast<const f16x8*>(base + ((s * READS + 4) * 1024 + wid * 128) % (STG_BYTES - 1024));
                f5 = *reinterpret_cast<const f16x8*>(base + ((s * READS + 5) * 1024 + wid * 128) % (STG_BYTES - 1024));
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0, f4, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f3, f1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2, f4, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f5, f0, acc[3], 0, 0, 0);
            if (SPREAD) {
#pragma unroll
                for (int p = (s * PIECES) / 12; p < ((s + 1) * PIECES) / 12; ++p) issue_piece(1 - stage, p);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (SPREAD) off = (off + 8u * PIECES * 1024u) & rmask;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0;
    for (int j = 0; j < 4; ++j) t += acc[j][lane & 15];
    if (t == 12345.f) sink[blockIdx.x] = t;
}

template <int PIECES, bool SPREAD = false, int LOADERS = 0, int READS = 3>
void run(const char* what, const char* buf, size_t region, int blocks, int iters) {
    float* sink; hipMalloc(&sink, blocks * 4);
    const size_t ldsb = 160 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<PIECES, SPREAD, LOADERS, READS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 12; ++w) hipLaunchKernelGGL((probe<PIECES, SPREAD, LOADERS, READS>), dim3(blocks), dim3(512), ldsb, 0, buf, region, iters, sink);    // warm clocks
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<PIECES, SPREAD, LOADERS, READS>), dim3(blocks), dim3(512), ldsb, 0, buf, region, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double flops = (double)blocks * iters * (8 - LOADERS) * 48 * 32768.0;       // computing waves x 48 instructions x 32 x 32 x 16 x 2
    const double bytes = (double)blocks * iters * 8 * PIECES * 1024.0;
    const double cyc_per_block = ms * 1e-3 / iters * 2.1e9;                            // at a nominal 2.1 GHz, for orientation only
    printf("%-34s %2d KB / block / wave  %7.3f ms  matrix %6.0f TFLOP/s  copies %6.1f GB/s per CU (%5.2f TB/s)  ~%5.0f cycles per block\n",
           what, PIECES, ms, flops / ms / 1e9, bytes / ms / 1e6 / 256, bytes / ms / 1e9, cyc_per_block);
    hipFree(sink);
}

int main() {
    char* buf; const size_t region = (size_t)4 << 20;            // 4 MB (a power of two; W1 + W2 of one encoder layer as fp16 hi + bf8 are 3 MB)
    hipMalloc(&buf, region + 65536); hipMemset(buf, 1, region + 65536);
    const int blocks = 256, iters = 600;
    run<0>("no copies (skeleton's ceiling)", buf, region, blocks, iters);
    run<10>("shipped 256 x 256 tile", buf, region, blocks, iters);
    run<15>("in between", buf, region, blocks, iters);
    run<19>("fused MLP, 64-row workgroups", buf, region, blocks, iters);
    run<25>("A-stationary QKV, 64 rows", buf, region, blocks, iters);
    run<0>("no copies again (clock check)", buf, region, blocks, iters);
    printf("-- the same with the copies issued a few per step between the matrix instructions (as the shipped K loops do)\n");
    run<10, true>("shipped 256 x 256 tile", buf, region, blocks, iters);
    run<15, true>("in between", buf, region, blocks, iters);
    run<19, true>("fused MLP, 64-row workgroups", buf, region, blocks, iters);
    run<25, true>("A-stationary QKV, 64 rows", buf, region, blocks, iters);
    run<0>("no copies again (clock check)", buf, region, blocks, iters);
    printf("-- copies spread, and SIX fragment reads per four matrix instructions (the 32 x 64 wave tiles of a 64-row workgroup)\n");
    run<0, true, 0, 6>("no copies", buf, region, blocks, iters);
    run<10, true, 0, 6>("shipped tile's bytes", buf, region, blocks, iters);
    run<19, true, 0, 6>("fused MLP, 64-row workgroups", buf, region, blocks, iters);
    run<25, true, 0, 6>("A-stationary QKV, 64 rows", buf, region, blocks, iters);
    printf("-- two of the eight waves only copy, six only compute (the matrix rate counts the six)\n");
    run<0, false, 2>("no copies, six computing waves", buf, region, blocks, iters);
    run<10, false, 2>("shipped tile's bytes", buf, region, blocks, iters);
    run<15, false, 2>("in between", buf, region, blocks, iters);
    run<19, false, 2>("fused MLP's bytes", buf, region, blocks, iters);
    run<25, false, 2>("A-stationary QKV's bytes", buf, region, blocks, iters);
    printf("-- one wave copies, seven compute\n");
    run<0, false, 1>("no copies, seven computing waves", buf, region, blocks, iters);
    run<10, false, 1>("shipped tile's bytes", buf, region, blocks, iters);
    run<19, false, 1>("fused MLP's bytes", buf, region, blocks, iters);
    run<25, false, 1>("A-stationary QKV's bytes", buf, region, blocks, iters);
    return 0;
}
