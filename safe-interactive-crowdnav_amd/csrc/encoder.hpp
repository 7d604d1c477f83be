// Context encoder (Trajectron++ front end, PREDICT mode): three single-layer LSTMs over the hist_len
// history frames + additive attention over the two edge encodings -> ctx [*, 2H].
// MID/models/encoders/mgcvae.py:683-708 (history), :710-824 (edges), :826-880 + components/additive_attention.py.
//
// One workgroup per agent, one thread per LSTM gate row (4H threads).  The whole recurrent state and the
// agent's history live in LDS; weights are stored transposed ([in, 4H]) so that the 4H threads read
// consecutive addresses for each k.  The work is tiny (once per predictor call).
#pragma once
#include "common.hpp"

namespace jmid {

struct LstmW {
    const float* WihT;  // [in, 4H]
    const float* WhhT;  // [H, 4H]
    const float* b;     // [4H]  (bias_ih + bias_hh)
};

struct EncArgs {
    const float* x_st;       // [n, Th, 6]
    const float* nbr_sum;    // [n, 2, Th, 6]
    const float* edge_mask;  // [n, 2]
    LstmW hist, edge[2];
    const float* W1T;        // [H, H] transposed (in, out)
    const float* W2T;        // [H, H]
    const float* v;          // [H]
    float* ctx;              // [n, 2H]
    int n, Th, H;
};

constexpr int ENC_MAX_TH = 16;
constexpr int ENC_MAX_H = 256;

__device__ inline void lstm_run(const LstmW& w, const float* xin /*LDS [Th, in]*/, int in, int Th, int H, float* gates,
                                float* hbuf, float* cbuf) {
    const int row = threadIdx.x;  // gate row, 0..4H-1
    const int H4 = 4 * H;
    if (row < H) {
        hbuf[row] = 0.f;
        cbuf[row] = 0.f;
    }
    __syncthreads();
    for (int t = 0; t < Th; ++t) {
        float acc = w.b[row];
        for (int k = 0; k < in; ++k) acc += w.WihT[k * H4 + row] * xin[t * in + k];
        for (int k = 0; k < H; ++k) acc += w.WhhT[k * H4 + row] * hbuf[k];
        gates[row] = acc;
        __syncthreads();
        if (row < H) {
            const float i = sigmoidf_(gates[row]);
            const float f = sigmoidf_(gates[H + row]);
            const float g = tanhf(gates[2 * H + row]);
            const float o = sigmoidf_(gates[3 * H + row]);
            const float c = f * cbuf[row] + i * g;
            cbuf[row] = c;
            hbuf[row] = o * tanhf(c);
        }
        __syncthreads();
    }
}

// The same recurrence with the thread's weight row in REGISTERS: W_hh is 4H x H floats = 256 KB at H = 128, and lstm_run re-reads it
// from L2 at every one of the 18 recurrent steps of a call (2.6 us per step through the CU's load path: most of the 92 us the
// encoder took at the shipped operating point, where it was 10 % of the device time).  Here a thread loads its row once per LSTM
// (H + 12 registers) and the steps run from registers and LDS.  Same products, same order: bit-identical to lstm_run.
template <int H>
__device__ inline void lstm_run_reg(const LstmW& w, const float* xin /*LDS [Th, in]*/, int in, int Th, float* gates, float* hbuf,
                                    float* cbuf) {
    constexpr int H4 = 4 * H;
    const int row = threadIdx.x;  // gate row, 0..4H-1
    float wh[H], wi[12];
#pragma unroll
    for (int k = 0; k < H; ++k) wh[k] = w.WhhT[k * H4 + row];
#pragma unroll
    for (int k = 0; k < 12; ++k) wi[k] = k < in ? w.WihT[k * H4 + row] : 0.f;
    const float bias = w.b[row];
    if (row < H) {
        hbuf[row] = 0.f;
        cbuf[row] = 0.f;
    }
    __syncthreads();
    for (int t = 0; t < Th; ++t) {
        float acc = bias;
#pragma unroll
        for (int k = 0; k < 12; ++k)
            if (k < in) acc += wi[k] * xin[t * in + k];
#pragma unroll
        for (int k4 = 0; k4 < H; k4 += 4) {
            const f32x4 h4 = *reinterpret_cast<const f32x4*>(hbuf + k4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc += wh[k4 + e] * h4[e];
        }
        gates[row] = acc;
        __syncthreads();
        if (row < H) {
            const float i = sigmoidf_(gates[row]);
            const float f = sigmoidf_(gates[H + row]);
            const float g = tanhf(gates[2 * H + row]);
            const float o = sigmoidf_(gates[3 * H + row]);
            const float c = f * cbuf[row] + i * g;
            cbuf[row] = c;
            hbuf[row] = o * tanhf(c);
        }
        __syncthreads();
    }
}

template <int HT>      // HT = the hidden size the register variant is compiled for (0: the generic loop over global memory)
__global__ __launch_bounds__(HT ? 4 * HT : 1024) void encoder_kernel(EncArgs a) {
    __shared__ float xin[ENC_MAX_TH * 12];
    __shared__ float gates[4 * ENC_MAX_H];
    __shared__ __attribute__((aligned(16))) float hbuf[ENC_MAX_H];
    __shared__ float cbuf[ENC_MAX_H];
    __shared__ float h_hist[ENC_MAX_H], u[2][ENC_MAX_H], red[ENC_MAX_H];
    __shared__ float score[2];
    const int ag = blockIdx.x, tid = threadIdx.x;
    const int Th = a.Th, H = a.H;

    // ---- history LSTM (input 6)
    for (int i = tid; i < Th * 6; i += blockDim.x) xin[i] = a.x_st[(size_t)ag * Th * 6 + i];
    __syncthreads();
    if constexpr (HT > 0) lstm_run_reg<HT>(a.hist, xin, 6, Th, gates, hbuf, cbuf);
    else lstm_run(a.hist, xin, 6, Th, H, gates, hbuf, cbuf);
    if (tid < H) h_hist[tid] = hbuf[tid];
    __syncthreads();

    // ---- edge LSTMs (input = [summed neighbour state(6), own state(6)])
    for (int e = 0; e < 2; ++e) {
        for (int i = tid; i < Th * 12; i += blockDim.x) {
            const int t = i / 12, k = i % 12;
            xin[i] = k < 6 ? a.nbr_sum[(((size_t)ag * 2 + e) * Th + t) * 6 + k]
                           : a.x_st[((size_t)ag * Th + t) * 6 + (k - 6)];
        }
        __syncthreads();
        if constexpr (HT > 0) lstm_run_reg<HT>(a.edge[e], xin, 12, Th, gates, hbuf, cbuf);
        else lstm_run(a.edge[e], xin, 12, Th, H, gates, hbuf, cbuf);
        if (tid < H) u[e][tid] = hbuf[tid] * a.edge_mask[(size_t)ag * 2 + e];
        __syncthreads();
    }

    // ---- additive attention over the two edge encodings, query = history encoding
    for (int e = 0; e < 2; ++e) {
        if (tid < H) {
            float acc = 0.f;
            for (int k = 0; k < H; ++k) acc += a.W1T[k * H + tid] * u[e][k] + a.W2T[k * H + tid] * h_hist[k];
            red[tid] = tanhf(acc) * a.v[tid];
        }
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
            for (int k = 0; k < H; ++k) s += red[k];
            score[e] = s;
        }
        __syncthreads();
    }
    if (tid < H) {
        const float mx = fmaxf(score[0], score[1]);
        const float e0 = expf(score[0] - mx), e1 = expf(score[1] - mx);
        const float inv = 1.0f / (e0 + e1);
        a.ctx[(size_t)ag * 2 * H + tid] = (e0 * inv) * u[0][tid] + (e1 * inv) * u[1][tid];
        a.ctx[(size_t)ag * 2 * H + H + tid] = h_hist[tid];
    }
}

inline hipError_t launch_encoder(const EncArgs& a, hipStream_t st) {
    if (a.H == 128) hipLaunchKernelGGL(encoder_kernel<128>, dim3(a.n), dim3(512), 0, st, a);
    else if (a.H == 16) hipLaunchKernelGGL(encoder_kernel<16>, dim3(a.n), dim3(64), 0, st, a);
    else hipLaunchKernelGGL(encoder_kernel<0>, dim3(a.n), dim3(4 * a.H), 0, st, a);
    return hipGetLastError();
}

}  // namespace jmid
