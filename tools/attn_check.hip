// The head-dim-128 LDS-DMA attention kernel (attn_f16x3_dma_kernel) on random planes: every softmax variant ("attn_sm" = 0: round 6,
// 2: rounds 2-5) against a float64 softmax(Q K^T) V of sampled query rows, and the time per launch of each in alternation (the chip's
// clock follows its power budget: a kernel timed first runs faster).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops -w \
//         -DJMID_DIAGNOSTICS -I safe-interactive-crowdnav_amd/csrc -I include tools/attn_check.hip -o build/attn_check
//   build/attn_check [nseq = 51] [S = 1200] [reps = 20] [mode: 0 = F16MX operands, 1 = F16X2, 2 = F16X3] [logit scale = 0.35] [nsplit = 1]
//   TRACE=1 (mode 0, no key split): cycle stamps per phase of the key-tile loop, both softmax forms.
//   ONE_WG=1 in the environment: ONE workgroup per CU (a wave alone on its SIMD).  DRIFT=x: the keys' scale grows by x per 32 keys
//   (the reference maximum has to move late in the sequence: the slow path of "attn_sm" = 0).
#include "attn_f16x3.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
using namespace jmid;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int nseq = argc > 1 ? atoi(argv[1]) : 51, S = argc > 2 ? atoi(argv[2]) : 1200, reps = argc > 3 ? atoi(argv[3]) : 20;
    const int mode = argc > 4 ? atoi(argv[4]) : 0; const float qscale = argc > 5 ? (float)atof(argv[5]) : 0.35f;
    const int nsplit = argc > 6 ? atoi(argv[6]) : 1;
    const float drift = getenv("DRIFT") ? (float)atof(getenv("DRIFT")) : 0.f;
    const int d = 512, nhead = 4, hd = 128, Spad = vt_spad(S);
    const size_t M = (size_t)nseq * S, Mpad = (M + 127) / 128 * 128 + 128;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> qf(Mpad * d, 0.f), kf(Mpad * d, 0.f), vf((size_t)nseq * d * Spad, 0.f);      // the fp32 values the planes carry
    std::vector<half_t> qh(Mpad * d), kh(Mpad * d), vt((size_t)nseq * d * Spad), ql(Mpad * d), kl(Mpad * d), vtl((size_t)nseq * d * Spad);
    std::vector<unsigned char> q8l(Mpad * d), k8h(Mpad * d), k8l(Mpad * d);
    auto top = [](half_t v) { return (unsigned char)((__builtin_bit_cast(unsigned short, v) + 0x80u) >> 8); };
    for (size_t i = 0; i < qh.size(); ++i) {
        const size_t tok = i / d;
        const float ks = 1.f + drift * (float)((tok % S) / 32);
        const float q = nd(rng) * qscale, k = nd(rng) * ks;
        qf[i] = q; kf[i] = k;
        qh[i] = (half_t)q; kh[i] = (half_t)k;
        ql[i] = (half_t)(q - (float)qh[i]);
        kl[i] = (half_t)(k - (float)kh[i]);
        q8l[i] = top(ql[i]);
        k8h[i] = top(kh[i]);
        k8l[i] = top(kl[i]);
    }
    for (size_t i = 0; i < vt.size(); ++i) { const float v = nd(rng); vf[i] = v; vt[i] = (half_t)v; vtl[i] = (half_t)(v - (float)vt[i]); }
    half_t *dQ, *dK, *dV, *dO[2], *dQl, *dKl, *dVl, *dOl[2];
    unsigned char *dQ8, *dK8h, *dK8l;
    float *dOpart, *dML;
    int* flag;
    const size_t oelems = blk_plane_elems(M, d) + 128 * d;
    CK(hipMalloc(&dQ, qh.size() * 2)); CK(hipMalloc(&dK, kh.size() * 2)); CK(hipMalloc(&dV, vt.size() * 2));
    CK(hipMalloc(&dQl, ql.size() * 2)); CK(hipMalloc(&dKl, kl.size() * 2)); CK(hipMalloc(&dVl, vtl.size() * 2));
    CK(hipMalloc(&dQ8, q8l.size())); CK(hipMalloc(&dK8h, k8h.size())); CK(hipMalloc(&dK8l, k8l.size()));
    CK(hipMalloc(&dOpart, (size_t)std::max(nsplit, 1) * M * d * 4 + 4096)); CK(hipMalloc(&dML, (size_t)std::max(nsplit, 1) * M * nhead * 8 + 4096));
    for (int v = 0; v < 2; ++v) { CK(hipMalloc(&dO[v], oelems * 2)); CK(hipMalloc(&dOl[v], oelems * 2)); CK(hipMemset(dO[v], 0, oelems * 2)); CK(hipMemset(dOl[v], 0, oelems * 2)); }
    CK(hipMalloc(&flag, 4));
    CK(hipMemcpy(dQl, ql.data(), ql.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dKl, kl.data(), kl.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dVl, vtl.data(), vtl.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dQ, qh.data(), qh.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dK, kh.data(), kh.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dV, vt.data(), vt.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dQ8, q8l.data(), q8l.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dK8h, k8h.data(), k8h.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dK8l, k8l.data(), k8l.size(), hipMemcpyHostToDevice));
    CK(hipMemset(flag, 0, 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    Tuning tn[2];
    tn[0].attn_sm = 2;      // rounds 2-5
    tn[1].attn_sm = 0;      // round 6
    if (getenv("ONE_WG")) tn[0].attn_one_wg = tn[1].attn_one_wg = 1;
    if (getenv("PRIO")) tn[0].attn_prio = tn[1].attn_prio = atoi(getenv("PRIO"));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[2] = {"attn_sm = 2 (rounds 2-5)", "attn_sm = 0 (round 6)   "};
    for (int round = 0; round < 3; ++round)
    for (int v = 0; v < 2; ++v) {
        AttnHArgs a{dQ, reinterpret_cast<half_t*>(dQ8), dK, nullptr, dV, nullptr, dO[v], nullptr, S, Spad, d, nhead, 1.f, flag, nsplit, dOpart, dML, 1,
                    dK8h, dK8l, dQ8};
        if (mode >= 1) { a.Qlo = dQl; a.Klo = dKl; a.K8h = a.K8l = a.Q8l = nullptr; }
        if (mode == 2) { a.Vtlo = dVl; a.Olo = dOl[v]; a.x2 = 0; }
        TuneScope ts(&tn[v]);
        CK(launch_attn_f16x3(a, nseq, hd, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CK(launch_attn_f16x3(a, nseq, hd, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double fl = 4.0 * nseq * (double)S * S * d;
        printf("%s: %.4f ms per launch  (%.0f TFLOP/s algorithmic)\n", names[v], ms / reps, fl / (ms / reps * 1e-3) * 1e-12);
    }
    if (getenv("TRACE") && mode == 0 && nsplit == 1) {
        // cycle stamps per phase of the key-tile loop, per wave (the TRACE instantiation of the kernel; stamps perturb the schedule a little)
        const int nqt = (S + 127) / 128, nblk = nqt * nhead * nseq;
        unsigned long long* tr;
        CK(hipMalloc(&tr, (size_t)nblk * 4 * 12 * 8));
        const char* pn[7] = {"prologue", "wait vmcnt", "barrier", "issue DMA", "QK^T", "softmax + pack", "P.V"};
        for (int v = 0; v < 2; ++v) {
            CK(hipMemset(tr, 0, (size_t)nblk * 4 * 12 * 8));
            AttnHArgs a{dQ, reinterpret_cast<half_t*>(dQ8), dK, nullptr, dV, nullptr, dO[v], nullptr, S, Spad, d, nhead, 1.f, flag, 1, dOpart, dML, 1, dK8h, dK8l, dQ8};
            a.mq = fast_div_magic(nqt, nblk); a.ms = fast_div_magic(1, nblk); a.mh = fast_div_magic(nhead, nblk); a.nseq = nseq;
            const size_t ldsb = getenv("ONE_WG") ? 160 * 1024 : ATT_DMA_LDS;
            if (v == 0) {
                auto k = &attn_f16x3_dma_kernel<true, true, true, false, true, true, 0>;
                CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(nblk), dim3(256), ldsb, st, a, nqt, 0, tr);
            } else {
                auto k = &attn_f16x3_dma_kernel<true, true, true, false, true, true, 1>;
                CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(nblk), dim3(256), ldsb, st, a, nqt, 0, tr);
            }
            CK(hipStreamSynchronize(st));
            std::vector<unsigned long long> t((size_t)nblk * 4 * 12);
            CK(hipMemcpy(t.data(), tr, t.size() * 8, hipMemcpyDeviceToHost));
            double sum[7] = {0}; size_t nw = 0;
            for (size_t w = 0; w < (size_t)nblk * 4; ++w) {
                if (!t[w * 12 + 4]) continue;      // idle waves (queries past S)
                ++nw;
                for (int i = 0; i < 7; ++i) sum[i] += (double)t[w * 12 + i];
            }
            const double tiles = (S + 31) / 32;
            double tot = 0;
            printf("%s, traced: %zu active waves; cycles per wave and key tile:", names[v], nw);
            for (int i = 1; i < 7; ++i) { printf("  %s %.0f", pn[i], sum[i] / nw / tiles); tot += sum[i] / nw / tiles; }
            printf("  total %.0f\n", tot);
        }
    }
    // float64 reference on sampled rows
    std::vector<half_t> oh[2], ol[2];
    for (int v = 0; v < 2; ++v) {
        oh[v].resize(oelems); ol[v].resize(oelems);
        CK(hipMemcpy(oh[v].data(), dO[v], oelems * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ol[v].data(), dOl[v], oelems * 2, hipMemcpyDeviceToHost));
    }
    double maxe[2] = {0, 0}, sume[2] = {0, 0}, maxd = 0;
    size_t cnt = 0, ndiff = 0;
    const int nsample = 96;
    std::vector<double> lg(S), o(hd);
    for (int smp = 0; smp < nsample; ++smp) {
        const int seq = (int)(rng() % (unsigned)nseq), h = (int)(rng() % (unsigned)nhead);
        const int q = smp < 8 ? (smp < 4 ? smp * 37 % S : S - 1 - smp) : (int)(rng() % (unsigned)S);
        const size_t qt = (size_t)seq * S + q;
        double mx = -1e300;
        for (int k = 0; k < S; ++k) {
            const size_t ktok = (size_t)seq * S + k;
            double s = 0;
            for (int c = 0; c < hd; ++c) s += (double)qf[qt * d + h * hd + c] * (double)kf[ktok * d + h * hd + c];
            lg[k] = s * 0.6931471805599453;      // Q is pre-scaled into log2 units: p = 2^s
            mx = std::max(mx, lg[k]);
        }
        double l = 0;
        for (int c = 0; c < hd; ++c) o[c] = 0;
        for (int k = 0; k < S; ++k) {
            const double p = exp(lg[k] - mx);
            l += p;
            const int kp = vt_key_pos(k);
            for (int c = 0; c < hd; ++c) o[c] += p * (double)vf[(((size_t)seq * nhead + h) * hd + c) * Spad + kp];
        }
        for (int c = 0; c < hd; ++c) {
            const size_t ob = blk_index((int)qt, h * hd + c, d);
            const double ref = o[c] / l;
            for (int v = 0; v < 2; ++v) {
                const double got = (double)(float)oh[v][ob] + (mode == 2 ? (double)(float)ol[v][ob] : 0.0);
                const double e = fabs(got - ref);
                maxe[v] = std::max(maxe[v], e);
                sume[v] += e;
            }
            ++cnt;
        }
    }
    for (size_t i = 0; i < oelems; ++i) {
        const double a0 = (double)(float)oh[0][i], a1 = (double)(float)oh[1][i];
        if (a0 != a1) { ++ndiff; maxd = std::max(maxd, fabs(a0 - a1)); }
    }
    for (int v = 0; v < 2; ++v) printf("%s: against float64 on %d sampled rows: max |err| %.3e, mean |err| %.3e\n", names[v], nsample, maxe[v], sume[v] / cnt);
    printf("O_hi words that differ between the two: %zu of %zu, max |d| %.3e\n", ndiff, oelems, maxd);
    int f = 0;
    CK(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost));
    printf("range flag %d\n", f);
    return 0;
}
