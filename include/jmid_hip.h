/*
 * jmid_hip.h  --  C ABI of libjmid_hip.so: the MI355X (gfx950) implementation of the
 * SICNav-Diffusion trajectory predictor hot path (iMID / JMID).
 *
 * The reference (sepsamavi/safe-interactive-crowdnav) has no FFI: its boundary is the
 * Python class HumanTrajectoryForecasterSim (sicnav_diffusion/JMID/mid_sim_wrapper.py:207).
 * The Python host layer in safe-interactive-crowdnav_amd/ keeps that class surface and
 * binds the entry points below with ctypes.  Every entry point names
 * the reference code it replaces (paths relative to sicnav_diffusion/JMID/).
 *
 * Conventions
 *   - plain C types only: pointers, sizes, ints, floats.  No torch / HIP types.
 *   - every function returns 0 on success, a negative JMID_E* code otherwise; the message
 *     is available from jmid_last_error().
 *   - buffers are caller-owned.  `mem` says where they live: JMID_MEM_HOST (the library
 *     copies through its stream) or JMID_MEM_DEVICE (device pointers on the handle's GPU,
 *     e.g. torch.Tensor.data_ptr(); no copies are made).
 *   - one HIP stream per handle, created hipStreamNonBlocking (no implicit ordering against the legacy null stream: other work of
 *     the process on the default stream neither waits for nor delays the predictor); a handle is not re-entrant (the reference is a
 *     single-threaded caller; mid_sim_wrapper.py:174 only locks its history buffer).
 *   - JMID_MEM_DEVICE calls are stream-ordered against the CALLER's stream (jmid_set_caller_stream, default the
 *     legacy null stream): on entry the handle's stream waits for everything the caller enqueued on that stream,
 *     on exit that stream waits for the call's last kernel.  The call may return before the outputs are complete
 *     (exact-fp32 mode; the split modes read a range flag back and therefore block): consume them on the caller's
 *     stream, or call jmid_synchronize first.  Buffers produced or consumed on any OTHER stream need the caller's
 *     own events.  JMID_MEM_HOST calls return with the outputs complete.
 *   - all floating point buffers are fp32, row-major, densely packed.
 *
 * Row / token order used throughout (matches the reference's `context.repeat(sample, 1)`,
 * MID/models/diffusion.py:496, extended with a leading episode axis):
 *     row r = (e * K + s) * A + a          e: episode, s: sample, a: agent
 *     x[r, t, c]                           t: horizon step, c in {x, y}
 */
#ifndef JMID_HIP_H
#define JMID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
typedef struct jmid_ctx* jmid_handle_t;      /* opaque */

/* The library is built with -fvisibility=hidden: the entry points below are all the C++ host code it exports (next to the
 * kernel handles the HIP runtime needs). */
#pragma GCC visibility push(default)

enum { JMID_NET_IMID = 0, JMID_NET_JMID = 1 };
enum { JMID_MEM_HOST = 0, JMID_MEM_DEVICE = 1 };
/* arithmetic used for the GEMM / attention contractions (softmax, LayerNorm, gates, DDIM are
 * always fp32):
 *   JMID_PREC_F32     exact fp32 MFMA (v_mfma_f32_32x32x2_f32)
 *   JMID_PREC_F16X3   fp32 emulated by three fp16 MFMAs per product on hi/lo-split operands
 *                     (~22 significand bits, fp32 accumulate)
 *   JMID_PREC_F16X2   same operand planes, but the linear contractions take the activation as its fp16 hi plane only:
 *                     A_hi x (W_hi + W_lo) in the GEMMs, two MFMAs per product, and P_hi x V_hi in attention (P rounded to nearest);
 *                     the softmax logits Q.K keep all three terms (their error is exponentiated) and the residual
 *                     stream, LayerNorm and DDIM state keep hi + lo.  Mean ADE vs the reference 7e-6 m on the cfg3
 *                     shape (F16X3: 1e-6 m; gate 1e-4 m), ~35 % more trajectories per second on batches; the lo planes this mode never reads are not written
 *   JMID_PREC_F16MX   F16X2 with the weight-lo correction term of every GEMM on the fp8 matrix path: A_hi x W_hi as fp16 MFMAs
 *                     plus ONE v_mfma_f32_32x32x64_f8f6f4 per 64-deep block on bf8(A_hi) x bf8(W_lo) (bf8 = e5m2 = the top byte
 *                     of the fp16 value, rounded to nearest; unscaled) - 1.5 instead of 2 MFMA passes per product.  The term is
 *                     2^-11 of the product, so its 2-bit significand costs nothing measurable: same GEMM error (2^-12.7) and
 *                     same ADE as F16X2 on every fixture.  In attention (head_dim 128) the two correction terms of the logits
 *                     take the same path (bf8 images of K from the QKV GEMM, of Q made in the kernel), and P.V is one MFMA per
 *                     product: P_hi . V_hi with P rounded to nearest - the one rounding every other activation of the mode gets.
 *                     ~20 % more trajectories per second than F16X2.
 *                     Bit-identical across chunk plans like the other modes.  What bench.py quotes, and since round 6 the default of
 *                     the Python class (together with its first-call self check against JMID_PREC_F16X3).
 *   JMID_PREC_F16     single fp16 MFMA (11 bits; does NOT meet the 1e-4 ADE gate, reported only; not built) */
enum { JMID_PREC_F32 = 0, JMID_PREC_F16X3 = 1, JMID_PREC_F16 = 2, JMID_PREC_F16X2 = 3, JMID_PREC_F16MX = 4 };

enum {
    JMID_OK = 0,
    JMID_EINVAL = -1,     /* bad argument / unsupported dimension */
    JMID_ENOWEIGHT = -2,  /* a required weight has not been loaded */
    JMID_EHIP = -3,       /* HIP runtime error */
    JMID_ENOMEM = -4,
    JMID_ERANGE = -5,     /* F16X3/F16X2/F16MX: an operand left the fp16 range; rerun with JMID_PREC_F32 */
    JMID_ETIMEOUT = -6    /* a workgroup of a one-launch GEMM + LayerNorm (small F16MX calls) gave up waiting for its partner workgroups -
                             not all of the launch was resident on the GPU (another process or stream held compute units).  Nothing to do
                             with the arithmetic: the outputs are undefined, the handle runs the unfused kernels from now on (same bits,
                             ~0.6 ms more per 50-step one-scene call); repeat the call in the SAME precision.  Counted: jmid_timeout_count */
};

/* Library / build identification (also the cheap "does it load" probe). */
const char* jmid_version(void);
/* Number of visible HIP devices (0 when there is no GPU); never fails. */
int jmid_device_count(void);

/* Construct a predictor engine on `device_id`.
 * Replaces the module construction of MID._build_model (MID/mid.py:1270-1297):
 *   net_kind  JMID_NET_JMID -> JointPredictionTransformerConcatLinear (MID/models/diffusion.py:153)
 *             JMID_NET_IMID -> TransformerConcatLinear               (MID/models/diffusion.py:112)
 *   ctx_dim   the yaml key `encoder_dim` (d_model = 2*ctx_dim, ff = 4*ctx_dim, LSTM hidden = ctx_dim/2)
 *   tf_layer  the yaml key `tf_layer`;  nhead is 4 in the reference (diffusion.py:121,162)
 *   hist_len  history frames fed to the context encoder (`past_num_frames`, 6 in env.config:11) */
int jmid_create(jmid_handle_t* out, int device_id, int net_kind, int ctx_dim, int tf_layer, int nhead,
                int hist_len);
int jmid_destroy(jmid_handle_t h);
const char* jmid_last_error(jmid_handle_t h);

/* Upload one named fp32 parameter from HOST memory.  Names are the reference's state-dict keys
 * (net: "concat1._layer.weight", "transformer_encoder.layers.0.self_attn.in_proj_weight", ...;
 * encoder: "PEDESTRIAN/node_history_encoder.weight_ih_l0", ...), i.e. what
 * model.load_state_dict(ckpt["ddpm"]) / registrar.load_models(ckpt["encoder"]) consume
 * (MID/mid.py:1231-1232, 1291).  Unknown names are rejected. */
int jmid_load_weight(jmid_handle_t h, const char* name, const float* host_data, size_t n_elems);
/* Check that every parameter is present and build the device-side derived forms
 * (fp16 hi/lo planes, positional-encoding table).  Must be called once after loading. */
int jmid_finalize_weights(jmid_handle_t h);

/* Install the DDIM step table: n_steps entries, one per reverse step, in execution order.
 * Replaces VarianceSchedule + the scalar bookkeeping of sample_sicnav_inference
 * (MID/models/diffusion.py:12-64, 507-528):
 *   beta[i]                       betas[t_i]
 *   c_e[i] = sqrt(1 - abar_t)     c_x[i] = sqrt(abar_t)
 *   n_x[i] = sqrt(abar_{t-s})     n_e[i] = sqrt(1 - abar_{t-s})
 * so that  x0 = (x - e*c_e)/c_x ;  x <- n_x*x0 + n_e*e . */
int jmid_set_ddim_table(jmid_handle_t h, int n_steps, const float* beta, const float* c_e, const float* c_x,
                        const float* n_x, const float* n_e);

/* DDPM variant of the step table (sampling="ddpm", MID/models/diffusion.py:509-522, flexibility 0):
 *   c0[i] = 1/sqrt(alpha_t)   c1[i] = (1-alpha_t)/sqrt(1-abar_t)   sigma[i] = sigmas_inflex[t]
 *   use_noise[i] = (t > 1)    so that  x <- c0*(x - c1*e) + sigma*z   (z = 0 where use_noise is 0).
 * Installing it switches the handle to DDPM until jmid_set_ddim_table is called again. */
int jmid_set_ddpm_table(jmid_handle_t h, int n_steps, const float* beta, const float* c0, const float* c1,
                        const float* sigma, const int* use_noise);

/* Context encoder: Trajectron.get_latent in PREDICT mode
 * (MID/models/trajectron.py:416-454 -> MID/models/encoders/mgcvae.py:505-880).
 *   n_agents   total rows (episodes * agents), any order
 *   x_st       [n_agents, hist_len, 6]     standardized own history
 *   nbr_sum    [n_agents, 2, hist_len, 6]  per edge type (PED->PED, PED->ROBOT) the summed neighbour
 *                                          histories (zeros when there is none; mgcvae.py:726-757)
 *   edge_mask  [n_agents, 2]               clamp(sum(edge values), max=1)  (mgcvae.py:758-768, 821-822)
 *   ctx_out    [n_agents, ctx_dim] */
int jmid_encode(jmid_handle_t h, int n_agents, const float* x_st, const float* nbr_sum, const float* edge_mask,
                float* ctx_out, int mem);

/* The batched reverse-denoising loop + integrator:
 * DiffusionTraj.sample_sicnav_inference with sampling="ddim" (MID/models/diffusion.py:478-541),
 * the net forward (diffusion.py:133-150 / 173-209) and SingleIntegrator.integrate_samples
 * (MID/models/encoders/dynamics/single_integrator.py:290-321).
 *   E, A, K, T  episodes, agents per episode, samples, horizon
 *   x_T       [E, K*A, T, 2]  initial noise (drawn by the host with torch's CPU generator so that the
 *                             reference's RNG contract is kept, diffusion.py:499)
 *   ctx       [E, A, ctx_dim]
 *   p0        [E, A, 2]       current positions (initial condition of the integrator); may be NULL
 *                             when pos_out is NULL
 *   dt        env time_step
 *   vel_out   [E, K, A, T, 2] predicted velocities            (may be NULL)
 *   pos_out   [E, K, A, T, 2] cumsum(vel)*dt + p0             (may be NULL)
 * JMID attention spans all (t, s, a) tokens of ONE episode (block-diagonal over episodes). */
int jmid_denoise(jmid_handle_t h, int E, int A, int K, int T, const float* x_T, const float* ctx,
                 const float* p0, float dt, int precision, float* vel_out, float* pos_out, int mem);

/* DDPM sampling loop: as jmid_denoise, plus the per-step normal draws the reference takes from the torch generator
 * after x_T (diffusion.py:509): z [n_steps, E, K*A, T, 2] (the host draws them so the RNG contract is kept). */
int jmid_denoise_ddpm(jmid_handle_t h, int E, int A, int K, int T, const float* x_T, const float* z, const float* ctx,
                      const float* p0, float dt, int precision, float* vel_out, float* pos_out, int mem);

/* One evaluation of the denoising net e_theta([x, ctx], beta) for step-table entry `step_idx`
 * (diffusion.py:520); used by the parity tests.  x [E, K*A, T, 2] -> e_out same shape. */
int jmid_net_eval(jmid_handle_t h, int E, int A, int K, int T, int step_idx, const float* x, const float* ctx,
                  int precision, float* e_out, int mem);

/* Per-episode displacement metrics of the sampled futures for the multi-episode evaluation sweep (ADE / FDE as
 * defined in MID/evaluation/evaluation.py:11-28; minimum taken jointly over the scene's agents):
 *   pos [E, K, A, T, 2], gt [E, A, T, 2] -> out [E, 4] = {mean ADE, min-over-samples ADE, mean FDE, min FDE}. */
int jmid_episode_metrics(jmid_handle_t h, int E, int A, int K, int T, const float* pos, const float* gt,
                         float* out, int mem);

/* Joint-KDE ranking of the K sampled futures of every episode and selection of the k most likely ones:
 * get_most_likely_samples (sicnav_diffusion/JMID/mid_sim_wrapper.py:14-169, the joint branch :20-21 the predictor always takes;
 * called from predict_ret_best when num_ret_samples < K, :487-492), which the reference runs on its GPU when it has one (:26-30).
 *   pos   [E, K, A, T, 2]  integrated sample trajectories (jmid_denoise's pos_out layout), or NULL: the positions of the most
 *                          recent jmid_denoise on this handle (same E, A, K, T, called with p0) - the samples then never leave
 *                          the GPU.  They are forgotten (JMID_EINVAL here) when that call returned JMID_ERANGE and by any later
 *                          host-mode jmid_encode / jmid_episode_metrics or denoise call on the handle, which reuse the workspace
 *   bw    [T] KDE bandwidth per horizon step, exp(linspace(ln .01, ln .1, T)) as the reference computes it (:26-30), or NULL
 *                          (computed in the library)
 *   sel   [E, A, k, T, 2]  the kept samples in ascending likelihood (the reference's argsort(...)[-k:], :117-121)
 *   logw  [E, A, k]        their renormalised log-weights, the same row for every agent (:139-151)
 * fp64 inside (A <= 32, K <= 1024, T <= 24: JMID_EINVAL beyond); exact ties are broken by sample index (torch.argsort's tie
 * order is not reproduced); non-finite totals rank lowest. */
int jmid_topk(jmid_handle_t h, int E, int A, int K, int T, int k, const float* pos, const float* bw, float* sel, float* logw,
              int mem);

/* One predictor call end to end, host buffers in, host buffers out: what HumanTrajectoryForecasterSim.predict_ret_best issues per
 * MPC step (sicnav_diffusion/JMID/mid_sim_wrapper.py:482-510 -> MID.eval_sicnav, MID/mid.py:298-349) - jmid_encode, jmid_denoise
 * and, when k < K, jmid_topk chained on the handle's stream with ONE upload (a pinned staging buffer), no host round trip between
 * the stages (the context and the K sampled futures never leave the GPU) and ONE download + synchronisation at the end.
 *   x_st [E*A, hist_len, 6], nbr_sum [E*A, 2, hist_len, 6], edge_mask [E*A, 2]   as jmid_encode
 *   x_T [E, K*A, T, 2], p0 [E, A, 2], dt, precision                              as jmid_denoise
 *   k < K:  bw [T] as jmid_topk (or NULL); sel [E, A, k, T, 2] and logw [E, A, k] receive the k most likely futures; pos_out may
 *           be NULL
 *   k == K: no ranking (the reference skips it, mid_sim_wrapper.py:487-492); pos_out [E, K, A, T, 2] receives all futures, sel /
 *           logw / bw are ignored
 * Returns JMID_ERANGE like jmid_denoise (the outputs are then undefined: repeat the stages in JMID_PREC_F32). */
int jmid_predict(jmid_handle_t h, int E, int A, int K, int T, int k, const float* x_st, const float* nbr_sum, const float* edge_mask,
                 const float* x_T, const float* p0, float dt, int precision, const float* bw, float* sel, float* logw, float* pos_out);

/* The stream (a hipStream_t passed as void*, e.g. torch.cuda.current_stream().cuda_stream; NULL = the legacy default
 * stream) that produces the inputs and consumes the outputs of this handle's JMID_MEM_DEVICE calls - see Conventions. */
int jmid_set_caller_stream(jmid_handle_t h, void* stream);

/* Number of calls on this handle whose denoise loop ran as a replayed hipGraph (see "graph" below); -1 for a null handle. */
int64_t jmid_graph_replays(jmid_handle_t h);

/* Number of calls on this handle that ended with JMID_ERANGE (an fp16 operand left the fp16 range in a split-fp16 mode and the
 * caller had to repeat the call in JMID_PREC_F32 - the Python predictor does, forecaster.py: what the reference computes in fp32
 * throughout, MID/models/diffusion.py:478-541, so the result is unchanged and only the latency differs); -1 for a null handle.
 * A deployment with trained weights reads this to see how often the slow path fires. */
int64_t jmid_erange_count(jmid_handle_t h);
/* Number of calls on this handle that ended with JMID_ETIMEOUT (at most one per handle in practice: the first one switches the handle
 * to the unfused kernels for good); -1 for a null handle. */
int64_t jmid_timeout_count(jmid_handle_t h);

/* ---- tuning / measurement ------------------------------------------------------------------ */
/* Episodes processed together per pass of the 50-step loop (0 = automatic: a whole number of rounds of the attention
 * launch, a short ragged tail spread over the full chunks).  Results are bit-identical for every chunking of the same
 * call; the split-KV factor of the attention launches is a function of (E, A, K, T) and the precision only, so calls with different
 * episode counts agree to rounding (ADE ~1e-7 m), not bit for bit, when head_dim is 128. */
int jmid_set_chunk_episodes(jmid_handle_t h, int episodes);
/* Run-time switches of a handle.  The production library knows ONE key:
 *   "lanes"           chunks of the denoise loop in flight at once on separate HIP streams, 1..4 (default 2; the results do
 *                     not depend on it)
 * The diagnostics flavour of the library (built with -DJMID_DIAGNOSTICS as csrc/libjmid_hip_diag.so; what tests/ and tools/
 * load) additionally takes the implementation knobs the experiments of docs/NOTEBOOK.md are made with - kernel-variant
 * selectors such as "gemm_h_variant", "ln_fuse", "ln_rows", "mx_ln", "attn_mx", "attn_nsplit", "vt_stage", "graph",
 * "tail_fuse", "out_traj", "csl_swap", "attn_pf" (listed with their value ranges in csrc/jmid_abi.hip::jmid_set_tuning and
 * csrc/common.hpp::Tuning; every variant of a key computes the same values, most of them bit-identically) - and, with
 * -DJMID_ABLATIONS on top, the timing ablations "gemm_abl" / "attn_abl" (WRONG results).  Every switch belongs to the handle it
 * is set on.  Unknown keys return JMID_EINVAL. */
int jmid_set_tuning(jmid_handle_t h, const char* key, int value);
/* Per-kernel-class timing with HIP events recorded on the handle's stream.
 * mask: bit i enables class i (see jmid_kernel_class_name); 0 disables.  Timers accumulate until reset. */
int jmid_profile_enable(jmid_handle_t h, uint32_t class_mask);
int jmid_profile_reset(jmid_handle_t h);
/* Synchronizes the stream and returns, for kernel class `cls`, the number of launches and the summed
 * duration in milliseconds since the last reset. */
int jmid_profile_get(jmid_handle_t h, int cls, int64_t* n_launches, double* total_ms);
int jmid_kernel_class_count(void);
const char* jmid_kernel_class_name(int cls);
/* Block until all work queued on the handle's stream has finished. */
int jmid_synchronize(jmid_handle_t h);

/* ---- diagnostics: single-kernel entry points for the unit tests (HOST buffers only) ----------- */
/* Exported by the diagnostics flavour only (-DJMID_DIAGNOSTICS, csrc/libjmid_hip_diag.so). */
#ifdef JMID_DIAGNOSTICS
/* C[M,N] = A[M,K] . Wt[N,K]^T + bias (optional ReLU): the nn.Linear contraction of every layer. */
int jmid_dbg_gemm(jmid_handle_t h, int M, int N, int K, const float* A, const float* Wt, const float* bias, int relu,
                  int precision, float* C);
/* Multi-head self-attention over `nseq` sequences of length S from a packed QKV buffer
 * [nseq*S, 3*d_model] -> OUT [nseq*S, d_model] (heads/dims of the handle). */
int jmid_dbg_attention(jmid_handle_t h, int nseq, int S, const float* QKV, int precision, float* OUT);
/* X <- LayerNorm(X + A . Wt^T + bias) * gamma + beta in JMID_PREC_F16MX at d_model 512 (A [M, K], Wt [512, K], X [M, 512]), with the
 * second-generation kernels: fused = 1 the row-complete GEMM + residual + LayerNorm, 0 the GEMM + add_ln2 pair, 3 the small-launch
 * GEMM whose workgroups exchange the row statistics and normalise their own columns (all bit-identical). */
int jmid_dbg_gemm_ln_mx(jmid_handle_t h, int M, int K, const float* A, const float* Wt, const float* bias, const float* gamma,
                        const float* beta, float* X, int fused);
/* X <- LayerNorm(X + Y) * gamma + beta, eps = 1e-5 (post-norm residual of nn.TransformerEncoderLayer). */
int jmid_dbg_add_layernorm(jmid_handle_t h, int M, int d, float* X, const float* Y, const float* gamma,
                           const float* beta);
/* The chunk plan run_network would use for a call of E episodes of `tokens_per_episode` tokens (host logic only, no device):
 * writes at most `cap` chunk sizes to `sizes`, returns the number of chunks (or a negative JMID_E* code). */
int jmid_dbg_plan_chunks(int net_kind, int nhead, int lanes, int chunk_episodes, int E, int tokens_per_episode, int* sizes, int cap);
/* ... of a call in arithmetic mode `precision` (a JMID_PREC_F16MX batch of at most 2 560 tokens stays ONE chunk: its out-projection /
 * linear2 launches then carry the LayerNorm and the split-KV merge; every other mode runs such a batch as two halves side by side). */
int jmid_dbg_plan_chunks_mode(int net_kind, int nhead, int lanes, int chunk_episodes, int E, int tokens_per_episode, int precision, int* sizes, int cap);
#endif /* JMID_DIAGNOSTICS */

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* JMID_HIP_H */
