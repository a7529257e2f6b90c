/* resel_hip.h - C ABI of the MI355X (gfx950) kernels behind the `offpolicy_rnn` layer/operator interface.
 *
 * Drop-in boundary for the full-trajectory recurrent SAC/TD3 update of FanmingL/Recurrent-Offpolicy-RL.
 * The reference has no native code of its own: its GPU path binds Triton kernels and three un-vendored
 * CUDA extensions from Python.  Each entry point below names the reference interface it replaces
 * (path:line in the reference checkout).  The Python binding is recurrent-offpolicy-rl_amd/offpolicy_rnn/hip/_lib.py
 * (ctypes); INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's caching allocator); nothing is
 *     allocated, freed or synchronised inside; all work is enqueued on `stream` (a hipStream_t);
 *   - activations are TOKEN-MAJOR: logical [B, L, C] with channel stride 1, token index tok = b*L + t and an
 *     explicit token stride `ld_*` in ELEMENTS (so column slices of a wider row-major matrix are passed
 *     without copies); per-token flags are dense [B*L] fp32; base pointers and ld must be 16-byte aligned
 *     (ld % 4 == 0) where noted;
 *   - fp32 everywhere unless a name says bf16; no float atomics: results are bitwise reproducible;
 *   - return value: 0 = enqueued, <0 = RESEL_E* (nothing enqueued).  Scratch is caller-provided; its size
 *     comes from the matching *_workspace_bytes() function.
 */
#ifndef RESEL_HIP_H
#define RESEL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RESEL_OK 0
#define RESEL_EINVAL (-1)   /* unsupported shape / null pointer / misalignment */
#define RESEL_ELAUNCH (-2)  /* hipLaunchKernel reported an error */

typedef void* resel_stream_t; /* hipStream_t */

/* Library / device identification (used by the binding to fail loudly when the wrong build is loaded). */
int resel_abi_version(void);            /* bumps when a signature changes */
const char* resel_build_info(void);     /* "gfx950 <date> ..." */

/* Diagnostics for bench.py (off by default; the only global state of the library).  While enabled, the sequence-layer
 * kernels are dispatched with a (start, stop) HIP event pair bound to each dispatch on its own stream;
 * resel_profile_collect() synchronises those events, returns their summed duration and count and clears them.
 * kernel_id: one of RESEL_PROF_* (attn_q_kernel is counted as ATTN_FWD or ATTN_DQ by its mode; the gru slots cover the
 * persistent kernels or, on the per-step path, every step launch). */
enum {
    RESEL_PROF_SSCAN_FWD = 0, RESEL_PROF_SSCAN_BWD = 1, RESEL_PROF_ATTN_FWD = 2, RESEL_PROF_ATTN_DQ = 3, RESEL_PROF_ATTN_DKV = 4,
    RESEL_PROF_LINREC_REAL_FWD = 5, RESEL_PROF_LINREC_REAL_BWD = 6, RESEL_PROF_LINREC_COMPLEX_FWD = 7, RESEL_PROF_LINREC_COMPLEX_BWD = 8,
    RESEL_PROF_GRU_FWD = 9, RESEL_PROF_GRU_BWD = 10, RESEL_PROF_CONV_FWD = 11, RESEL_PROF_CONV_BWD = 12, RESEL_PROF_GEMM = 13,
    RESEL_PROF_SSCAN_FWD_LOCAL = 14, RESEL_PROF_SSCAN_BWD_LOCAL = 15,   /* time-parallel form: the local passes (+ carry) of a call */
    RESEL_PROF_NSLOTS = 16
};
int resel_profile_enable(int on);
int resel_profile_collect(int kernel_id, double* total_us, int* launches);

/* ------------------------------------------------------------------------------------------------------
 * smamba selective scan.  Replaces `selective_scan_cuda.fwd / .bwd` (modified Mamba CUDA extension with the
 * extra `start` reset input) bound at offpolicy_rnn/models/smamba/mamba_ssm/ops/selective_scan_interface_new.py:47
 * and :72; arithmetic spec = `selective_scan_ref`, same file :96-166.
 *   delta' = softplus(delta + delta_bias)            (delta_softplus = 1; 0: delta' = delta + delta_bias;
 *                                                      2: `delta` already IS softplus(raw + bias) - the producing GEMM's epilogue applied
 *                                                      it (resel_gemm_f32x act 3) - the forward uses it as it stands and the backward
 *                                                      returns the gradient with respect to the RAW value: x (1 - exp(-delta)))
 *                                                      + 4 (ABI 7): `A` holds the parameter A_log; the kernels form A = -exp(A_log)
 *                                                      themselves (reference smamba/mamba.py:187) and the backward returns dA_log)
 *   h_t    = exp(delta'_t * A) * (1 - start_t) * h_{t-1} + delta'_t * Bm_t * u_t          h: [Di, N]
 *   out_t  = (<Cm_t, h_t> + D * u_t) * silu(z_t)     (gate skipped when z == NULL, skip term when D == NULL)
 * u, delta, z, out: [B*L, Di] (ld_u, ld_delta, ld_z, ld_out; 16-byte aligned); A: [Di, N] dense;
 * Bm, Cm: [B*L, N] (ld_b, ld_c); D, delta_bias: [Di]; start: [B*L] or NULL.
 * Supported: Di % 4 == 0, N in {4, 8, 16, 32, 64}.
 * ckpt (optional, for the backward): state checkpoints every RESEL_SSCAN_CKPT steps,
 *   resel_selective_scan_ckpt_bytes(B, L, Di, N) bytes.  last_state (optional): [B, Di, N].
 * time_segments: 0 = the library decides (one workgroup scans a whole row when B * Di / 64 rows x tiles fill the chip; small
 *   batches are cut into time segments scanned in parallel: local pass, carry of the segment states, final pass), 1 = never
 *   split, k > 1 = k segments; workspace: resel_selective_scan_fwd_workspace_bytes(...) for the same arguments (NULL if 0).
 */
#define RESEL_SSCAN_CKPT 8
size_t resel_selective_scan_ckpt_bytes(int B, int L, int Di, int N);
size_t resel_selective_scan_fwd_workspace_bytes(int B, int L, int Di, int N, int time_segments);
int resel_selective_scan_fwd(const float* u, int64_t ld_u, const float* delta, int64_t ld_delta,
                             const float* z, int64_t ld_z, const float* A,
                             const float* Bm, int64_t ld_b, const float* Cm, int64_t ld_c,
                             const float* D, const float* delta_bias, const float* start,
                             float* out, int64_t ld_out, float* ckpt, float* last_state, void* workspace,
                             int B, int L, int Di, int N, int delta_softplus, int time_segments,
                             void* amax_out, unsigned amax_epoch, resel_stream_t stream);

/* Backward of the above.  dout: [B*L, Di] (ld_dout).  Outputs: du, ddelta, dz: [B*L, Di] (dz may be NULL iff
 * z is NULL); dBm, dCm: [B*L, N]; dA: [Di, N]; dD, ddelta_bias: [Di] (NULL iff the input was NULL).
 * All outputs are overwritten (not accumulated).  workspace: resel_selective_scan_bwd_workspace_bytes(). */
size_t resel_selective_scan_bwd_workspace_bytes(int B, int L, int Di, int N, int time_segments);
int resel_selective_scan_bwd(const float* u, int64_t ld_u, const float* delta, int64_t ld_delta,
                             const float* z, int64_t ld_z, const float* A,
                             const float* Bm, int64_t ld_b, const float* Cm, int64_t ld_c,
                             const float* D, const float* delta_bias, const float* start,
                             const float* dout, int64_t ld_dout, const float* ckpt,
                             float* du, int64_t ld_du, float* ddelta, int64_t ld_ddelta, float* dz, int64_t ld_dz,
                             float* dBm, int64_t ld_db, float* dCm, int64_t ld_dc,
                             float* dA, float* dD, float* ddelta_bias, void* workspace,
                             int B, int L, int Di, int N, int delta_softplus, int time_segments,
                             void* amax_dz, void* amax_ddelta, unsigned amax_epoch, resel_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * smamba depthwise causal conv1d + bias + SiLU on the masked input.  Replaces `causal_conv1d_cuda.causal_conv1d_fwd/bwd`
 * (selective_scan_interface_new.py:199,273) and the nn.Conv1d(groups=d_inner) + SiLU branch taken for d_conv > 4
 * (offpolicy_rnn/models/smamba/mamba.py:75-83, 210-212).
 *   y[tok, d] = silu(bias[d] + sum_k w[d, k] * mask[tok'] * x[tok', d]),  tok' = tok - (K-1) + k within the row b
 * x, y: [B*L, Di] (ld_x, ld_y); w: [Di, K] dense; bias: [Di] or NULL; mask: [B*L] or NULL.  1 <= K <= 32.
 * Backward: dx [B*L, Di] (ld_dx), dw [Di, K], dbias [Di]; workspace resel_causal_conv1d_bwd_workspace_bytes().
 */
int resel_causal_conv1d_fwd(const float* x, int64_t ld_x, const float* w, const float* bias, const float* mask,
                            float* y, int64_t ld_y, int B, int L, int Di, int K, int silu, void* amax_y, unsigned amax_epoch,
                            resel_stream_t stream);
size_t resel_causal_conv1d_bwd_workspace_bytes(int B, int L, int Di, int K);
int resel_causal_conv1d_bwd(const float* x, int64_t ld_x, const float* w, const float* bias, const float* mask,
                            const float* dy, int64_t ld_dy, float* dx, int64_t ld_dx, float* dw, float* dbias,
                            void* workspace, int B, int L, int Di, int K, int silu, void* amax_dx, unsigned amax_epoch,
                            resel_stream_t stream);
/* The same with TWO gradients of the output, summed on load (dy2 may be NULL): the conv output of the Mamba mixer feeds the scan AND
 * x_proj, so its gradient is the scan's du plus the x_proj input gradient - taken as two tensors there is no accumulating GEMM epilogue
 * (150 us against 67 for the plain form at 66 752 x 512 x 80) and no add pass. */
int resel_causal_conv1d_bwd2(const float* x, int64_t ld_x, const float* w, const float* bias, const float* mask,
                             const float* dy, int64_t ld_dy, const float* dy2, int64_t ld_dy2, float* dx, int64_t ld_dx,
                             float* dw, float* dbias, void* workspace, int B, int L, int Di, int K, int silu,
                             void* amax_dx, unsigned amax_epoch, resel_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Fused residual add + LayerNorm / RMSNorm.  Replaces the Triton kernels `_layer_norm_fwd_1pass_kernel` /
 * `_layer_norm_bwd_kernel` (offpolicy_rnn/models/smamba/mamba_ssm/ops/triton/layernorm.py:65,196; CPU spec
 * layernorm_cpu.py:6-35).  rows M, width C (C % 4 == 0, C <= 2048), all dense [M, C].
 *   res_out = x (+ residual) ; y = (res_out - mean) * rstd * w + b      (rms: y = res_out * rstd * w (+ b))
 * residual / res_out / bias may be NULL.  stats: [M, 2] (mean, rstd) saved for the backward.
 * Backward: dres_in (optional, [M, C]) is the gradient arriving at res_out from downstream; dx receives the
 * gradient w.r.t. x (== w.r.t. residual).  dw, db: [C]; workspace resel_add_layernorm_bwd_workspace_bytes().
 * act (ABI 8): 1 = y = elu(.) of the above (the plain ELU that follows a gilr / lru layer, reference rnn_base.py:456-469, applied where the
 * layer's closing add + LayerNorm stores its output); the backward then takes the bias `b` and forms elu' from the recomputed
 * pre-activation (no extra tensor saved).  0 = none.
 */
int resel_add_layernorm_fwd(const float* x, const float* residual, const float* w, const float* b,
                            float* y, float* res_out, float* stats, int M, int C, float eps, int rms, int act,
                            void* amax_y, unsigned amax_epoch, resel_stream_t stream);
size_t resel_add_layernorm_bwd_workspace_bytes(int M, int C);
int resel_add_layernorm_bwd(const float* dy, const float* dres_in, const float* res, const float* w, const float* b,
                            const float* stats, float* dx, float* dw, float* db, void* workspace,
                            int M, int C, int rms, int has_bias, int act, void* amax_dx, unsigned amax_epoch, resel_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Linear-recurrence scans (gilr, lru).  Replace the Triton kernels
 *   fwd_sequential_scan / bwd_sequential_scan          offpolicy_rnn/models/gilr/scan_triton/real_rnn_tie_input_gate.py:9,67
 *   fwd_sequential_scan_complex / bwd_...               offpolicy_rnn/models/lru/scan_triton/complex_rnn.py:44,91
 * (CPU specs real_rnn_tie_input_gate_cpu.py:4-14, complex_rnn_cpu.py:4-26) with a time-parallel chunked scan.
 * real:    v' = act ? tanh(v) : v ; f' = (act ? sigmoid(f) : f) * (1 - start) ; h_t = f'_t h_{t-1} + (1 - f'_t) v'_t
 * complex: h_t = lambda (1 - start_t) h_{t-1} + gamma * (vr_t + i vi_t)       (lambda, gamma per channel)
 * h, hr, hi, dh*: dense [B, L, C]; start: [B*L] or NULL; h0*: [B, C] or NULL (zeros).
 * v, f / vr, vi and their gradients may be column blocks of a wider token-major matrix (ABI 8): `ld_u` / `ld_du` = floats between the
 * rows of consecutive tokens (>= C; C for dense tensors).  The layers hand in the [M, E C] output of their shared-input EnsembleLinear
 * in place (reference gilr.py:60-62 / lru.py:112-120 slice an [E, B, T, C] tensor) and receive the gradients in the same layout.
 * Backward outputs are gradients w.r.t. the PRE-activation v, f (real) / vr, vi, lambda, gamma (complex;
 * dlam_re, dlam_im, dgamma: [C], reduced over B and L).  No gradient flows into h0 (reference: complex_rnn.py:242).
 * amax_h / amax_du (optional magnitude handles, see resel_amax): max |h| (complex: over hr and hi) / max over dv and df.
 */
int resel_linrec_real_fwd(const float* v, const float* f, int64_t ld_u, const float* start, const float* h0, float* h,
                          int B, int L, int C, int fuse_act, void* amax_h, unsigned amax_epoch, resel_stream_t stream);
int resel_linrec_real_bwd(const float* v, const float* f, int64_t ld_u, const float* start, const float* h0, const float* h,
                          const float* dh, float* dv, float* df, int64_t ld_du, int B, int L, int C, int fuse_act,
                          void* amax_du, unsigned amax_epoch, resel_stream_t stream);
int resel_linrec_complex_fwd(const float* vr, const float* vi, int64_t ld_u, const float* lam_re, const float* lam_im,
                             const float* gamma, const float* start, const float* h0r, const float* h0i,
                             float* hr, float* hi, int B, int L, int C, void* amax_h, unsigned amax_epoch, resel_stream_t stream);
size_t resel_linrec_complex_bwd_workspace_bytes(int B, int L, int C);
int resel_linrec_complex_bwd(const float* vr, const float* vi, int64_t ld_u, const float* lam_re, const float* lam_im,
                             const float* gamma, const float* start, const float* h0r, const float* h0i,
                             const float* hr, const float* hi, const float* dhr, const float* dhi,
                             float* dvr, float* dvi, int64_t ld_du, float* dlam_re, float* dlam_im, float* dgamma,
                             void* workspace, int B, int L, int C, resel_stream_t stream);

/* lru's per-channel parameters in one launch each way (reference offpolicy_rnn/models/lru/lru.py:104-110: exp / cos / sin / mul on [C]
 * tensors).  params_log [3, C] = (nu_log | theta_log | gamma_log) -> out [3, C] = (lam_re | lam_im | gamma) with
 * lambda = exp(-exp(nu_log)) (cos, sin)(exp(theta_log)), gamma = exp(gamma_log); bwd: dout [3, C] -> dparams_log [3, C].  (ABI 8) */
int resel_lru_params_fwd(const float* params_log, float* out, int C, resel_stream_t stream);
int resel_lru_params_bwd(const float* params_log, const float* dout, float* dparams_log, int C, resel_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * GRU recurrence on a hoisted input projection.  Replaces cuDNN/MIOpen GRU reached through
 * torch.nn.GRU(batch_first=True) (offpolicy_rnn/models/rnn_base.py:59,247; called at :453-454).
 * gi = x W_ih^T + b_ih: [B, L, 3H] dense (r, z, n blocks); w_hh: [3H, H]; b_hh: [3H]; h0: [B, H] or NULL.
 *   r = sig(gi_r + W_hr h + b_hr) ; z = sig(gi_z + W_hz h + b_hz) ; n = tanh(gi_n + r * (W_hn h + b_hn))
 *   h' = (1 - z) n + z h ;  h_all: [B, L, H].  gates (optional, training): [B, L, 4H] saves r, z, n and
 *   (W_hn h + b_hn) for the backward.  Supported: H % 16 == 0 and H <= 256, or H in {320, 384, 448, 512}.
 * The recurrence is T-sequential; each step is one launch (grid = H/16 unit slices x ceil(B/16) row groups)
 * enqueued back-to-back from C.  workspace: resel_gru_workspace_bytes() (re-laid-out W_hh slices, carry).
 * Backward returns the pre-activation gradients of BOTH projections: dgi [B, L, 3H] (w.r.t. gi) and
 * dgh [B, L, 3H] (w.r.t. W_hh h + b_hh); the caller finishes dW_hh = dgh^T h_prev and db_hh = sum dgh with one
 * library GEMM (they are plain GEMM/reduction shapes over B*L rows).  No gradient is produced for h0.
 */
size_t resel_gru_workspace_bytes(int B, int L, int H);
int resel_gru_seq_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* h0,
                      float* h_all, float* gates, void* workspace, int B, int L, int H, resel_stream_t stream);
int resel_gru_seq_bwd(const float* w_hh, const float* h0, const float* h_all, const float* gates,
                      const float* dh_all, float* dgi, float* dgh, void* workspace,
                      int B, int L, int H, resel_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * cgpt attention core: packed var-len causal attention with ALiBi, bf16 MFMA (v_mfma_f32_32x32x16_bf16), fp32 softmax.
 * Replaces the flash-attn varlen kernels reached through `flash_attn.modules.mha.MHA(causal=True, use_alibi=True)`
 * (offpolicy_rnn/models/flash_attention/TransformerFlashAttention.py:67-70, called under bf16 autocast at :80-81; packing
 * :107-112).  *** flash_attn is an un-vendored dependency: semantics restated (softmax(q k^T * scale - slope_h (i - j)), j <= i),
 * parity unpinned ***
 * qkv: [T, 3, H, hd] bf16 packed tokens (q | k | v); cu_seqlens: int32 [S + 1] device; slopes: [H] fp32 or NULL;
 * out: [T, H, hd] bf16; lse: [H, T] fp32 (base-2 log-sum-exp, saved for the backward).  hd in {32, 64}; max_seqlen bounds
 * the grid.  Backward: dqkv [T, 3, H, hd] bf16 (fully overwritten); workspace resel_attn_varlen_bwd_workspace_bytes().
 * Forward workspace (resel_attn_varlen_fwd_workspace_bytes(), may be NULL): holds the device-built work list - the real
 * (sequence, 128-token block) items of the ragged batch, longest first - that the kernels' workgroups take their work from;
 * without it they run in (sequence, block) order, which is slower on ragged batches.  Results do not depend on it.
 * Attention-probability dropout (MHA(dropout=p), TransformerFlashAttention.py:67-70, active in .train() passes): p_drop in
 * [0, 1); the keep mask is a counter function of (seed, offset, head, packed query token, key position) - no state, the
 * backward regenerates it from the same (seed, offset).  8-bit keep threshold floor((1 - p) * 255) + 1 and the 1 / (1 - p)
 * rescale as in flash-attn; the counter function itself is this library's (oracle/kernels.py `attn_dropout_keep`).
 * p_drop == 0 runs the dropout-free kernels.
 */
size_t resel_attn_varlen_fwd_workspace_bytes(int S, int max_seqlen);
int resel_attn_varlen_fwd(const uint16_t* qkv, const int32_t* cu_seqlens, const float* slopes, uint16_t* out, float* lse,
                          void* workspace, int T, int S, int H, int hd, int max_seqlen, float scale,
                          float p_drop, uint64_t seed, uint64_t offset, resel_stream_t stream);
size_t resel_attn_varlen_bwd_workspace_bytes(int T, int S, int H, int hd, int max_seqlen);
int resel_attn_varlen_bwd(const uint16_t* qkv, const int32_t* cu_seqlens, const float* slopes, const uint16_t* out,
                          const float* lse, const uint16_t* dout, uint16_t* dqkv, void* workspace,
                          int T, int S, int H, int hd, int max_seqlen, float scale,
                          float p_drop, uint64_t seed, uint64_t offset, resel_stream_t stream);

/* Element-wise dropout y = keep(i) ? x / (1 - p) : 0 with the keep mask a counter function of (seed, offset, element index)
 * (16-bit threshold round((1 - p) * 65536); oracle/kernels.py `dropout_keep`).  Replaces the `nn.Dropout`s of the cgpt
 * decoder block (TransformerFlashAttention.py:48,52,72,84-85) in training-mode passes; the backward is the same call on dy
 * with the same (seed, offset).  In place (y == x) allowed. */
int resel_dropout(const float* x, float* y, int64_t n, float p_drop, uint64_t seed, uint64_t offset, resel_stream_t stream);
/* Offset base of the counter-keyed masks: while `base` (a device uint64, 8-byte aligned; NULL switches it off) is set, every mask
 * kernel launched afterwards - resel_dropout, resel_gelu_dropout_fwd / _bwd, resel_attn_varlen_fwd / _bwd with p_drop > 0 - adds
 * *base to its `offset` argument WHEN IT RUNS.  A captured update (algorithm/graphed_update.py) bakes the host-drawn offsets into
 * its kernel nodes; a node of the same graph advances *base, so every replay draws fresh masks while the forward and backward
 * kernels of one replay agree.  Process-wide host state (one device per process); not a per-call argument. */
int resel_dropout_offset_base(const void* base);
/* y = dropout(gelu(x)) (erf form) in one pass and its backward dx = dy * keep / (1 - p) * gelu'(x) from the pre-activation x; same
 * counter-keyed mask as resel_dropout (p_drop = 0: plain GELU).  FFN hidden of the cgpt block (reference
 * models/flash_attention/TransformerFlashAttention.py:46-53: nn.GELU() followed by nn.Dropout). */
int resel_gelu_dropout_fwd(const float* x, float* y, int64_t n, float p_drop, uint64_t seed, uint64_t offset,
                           void* amax_y, unsigned amax_epoch, resel_stream_t stream);
int resel_gelu_dropout_bwd(const float* x, const float* dy, float* dx, int64_t n, float p_drop, uint64_t seed, uint64_t offset,
                           void* amax_dx, unsigned amax_epoch, resel_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * SAC / TD3 head, target and loss arithmetic + optimizer tail (the "fusions" of SURVEY.md section 8 row a16-a18).
 */
/* tanh-Gaussian head: contextual_sac_policy_single_head.py:109-123.  out2: [M, 2A] = (logstd | mean) as produced
 * by the actor's last Linear; noise: [M, A].  action_mean, action_sample: [M, A]; logp: [M].
 * Backward: d_out2 [M, 2A] from d_sample [M, A] and d_logp [M] (action_mean carries no gradient in the update). */
int resel_tanh_gaussian_fwd(const float* out2, const float* noise, float* action_mean, float* action_sample,
                            float* logp, int M, int A, resel_stream_t stream);
int resel_tanh_gaussian_bwd(const float* out2, const float* noise, const float* d_sample, const float* d_logp,
                            float* d_out2, int M, int A, resel_stream_t stream);

/* REDQ target: sac_full_length_rnn_redq.py:28-33 / td3_full_length_rnn_redq.py:29-35, clamp = QValueGuard
 * (utility/q_value_guard.py:22-38).  q: [E, M]; subset: m ensemble indices; next_logp: [M] or NULL (TD3);
 * done (already timeout-corrected), reward, mask: [M].  guard: device fp32[4] = {min, max, initialised, decay}:
 *   if !initialised: min/max <- extrema of v (first call, q_value_guard.py:23-26);  y = r + (1-d) gamma clamp(v)
 *   then the running update of q_value_guard.py:29-38 with y*mask - all on device, no host sync.
 * target: [M].  stats (optional) fp32[2]: max |target|, sum(mask). */
int resel_sac_target(const float* q, const int32_t* subset, int m, const float* next_logp, const float* log_alpha,
                     const float* reward, const float* done, const float* mask, float gamma, float* guard,
                     float* target, float* stats, void* workspace, int E, int M, resel_stream_t stream);
/* The same target in three phases for data-parallel runs: the Q-guard of the reference sees the extrema of the WHOLE batch
 * (utility/q_value_guard.py:22-38), so between the phases the caller all-reduces (MAX) the two-float blocks of `extrema`
 * [4] = {-min v, max v, -min(y mask), max(y mask)}: phase 0 writes [0:2] (rank-local), phase 1 initialises the guard from
 * the (now global) [0:2], clamps, forms y and writes [2:4], phase 2 updates the guard from the (now global) [2:4].
 * Same workspace for all phases of one target. */
int resel_sac_target_phase(int phase, const float* q, const int32_t* subset, int m, const float* next_logp, const float* log_alpha,
                           const float* reward, const float* done, const float* mask, float gamma, float* guard,
                           float* target, float* stats, float* extrema, void* workspace, int E, int M, resel_stream_t stream);
/* Data parallel with ONE collective per optimizer step (north_star: "a RCCL all-reduce of gradients ... and no other collectives"):
 * resel_sac_target_local forms the target from the rank's own rows with the guard AS IT STANDS (an uninitialised guard does not
 * clamp - on one process its first interval is the batch's own range, so the first clamp is the identity there too) and writes
 * the rank-local extrema [4] = {-min v, max v, -min(y mask), max(y mask)}; the caller puts them into its own row of a zero-filled
 * [world][4] tail of the critic gradient bucket, and after the SUM all-reduce resel_guard_apply_slots initialises / updates the
 * guard from the maxima over the rows - the guard of update k + 1 is then the one a single process over the global batch has
 * (the guard of the reference acts on the NEXT target only: utility/q_value_guard.py:22-38). */
int resel_sac_target_local(const float* q, const int32_t* subset, int m, const float* next_logp, const float* log_alpha,
                           const float* reward, const float* done, const float* mask, float gamma, const float* guard,
                           float* target, float* stats, float* extrema, void* workspace, int E, int M, resel_stream_t stream);
int resel_guard_apply_slots(const float* slots, int world, float* guard, resel_stream_t stream);
size_t resel_sac_target_workspace_bytes(int M);

/* Masked losses of the update, one forward and one backward pass each (reference sac_full_length_rnn_ensembleQ.py:80-81 `_mask_mean`,
 * :105-114 `_Q_loss`, sac_full_length_rnn_redq.py:37-49 / td3_full_length_rnn_redq.py:39-51 `_policy_loss`, :130-132 `_alpha_loss`);
 * UN-normalised sums (the global valid count divides the gradient inside AdamW).  q [E][M], y / mask / logp [M] (mask NULL = ones).
 *   critic  out2[0] = sum_m mask sum_e (q - y)^2;                           dq = 2 g[0] mask (q - y)
 *   actor   out2[0] = sum_m mask (use_logp exp(log_alpha) logp - red_e q),  out2[1] = sum_m mask logp;   red = mean (reduce_min 0) / min (1)
 *           dlogp = g[0] exp(log_alpha) mask;  dq = -g[0] mask / E (mean)  or  -g[0] mask at the FIRST minimal member, 0 elsewhere (min)
 * g: device scalar (the incoming gradient of the sum).  Fixed-order two-stage sums; workspace: resel_masked_loss_workspace_bytes(). */
size_t resel_masked_loss_workspace_bytes(void);
int resel_q_loss_fwd(const float* q, const float* y, const float* mask, float* out2, void* workspace, int E, int M, resel_stream_t stream);
int resel_q_loss_bwd(const float* q, const float* y, const float* mask, const float* g, float* dq, int E, int M, resel_stream_t stream);
int resel_actor_loss_fwd(const float* logp, const float* q, const float* mask, const float* log_alpha, float* out2, void* workspace,
                         int E, int M, int use_logp, int reduce_min, resel_stream_t stream);
int resel_actor_loss_bwd(const float* q, const float* mask, const float* log_alpha, const float* g, float* dlogp, float* dq,
                         int E, int M, int use_logp, int reduce_min, resel_stream_t stream);

/* Flat-buffer optimizer tail.  All parameter / gradient / moment tensors of one network live in ONE fp32 buffer.
 * soft update: rnn_base.py:490-491   target <- tau * target + (1 - tau) * online
 * adamw: torch.optim.AdamW (sac.py:61) with a per-segment learning rate table (RESeL groups,
 *        sac_full_length_rnn_redq_sep_optim.py:49-66): seg_end[i] = one-past-last element of segment i,
 *        seg_lr[i], seg_wd[i]; grad_scale multiplies the gradient first (1/valid_num or 1/world).
 * sumsq: out[0] = sum(x^2)  (l2_norm_square, rnn_base.py:531-532), deterministic two-stage reduction. */
int resel_soft_update(float* target, const float* online, float tau, int64_t n, resel_stream_t stream);
int resel_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end,
                     const float* seg_lr, const float* seg_wd, int nseg, float beta1, float beta2, float eps,
                     int step, const float* grad_scale, resel_stream_t stream);
/* The same step with the step-dependent factors read from DEVICE memory: bias_corrections[0] = 1 - beta1^t,
 * bias_corrections[1] = sqrt(1 - beta2^t).  For updates replayed from a hipGraph, where a by-value step would be frozen at capture. */
int resel_adamw_flat_dev(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end,
                         const float* seg_lr, const float* seg_wd, int nseg, float beta1, float beta2, float eps,
                         const float* bias_corrections, const float* grad_scale, resel_stream_t stream);
size_t resel_sumsq_workspace_bytes(int64_t n);
int resel_sumsq(const float* x, int64_t n, float* out, void* workspace, resel_stream_t stream);

/* ---- bias + activation tail of fc / efc-E layers, fused around the library GEMM -------------------------------
 * Replaces `activation(linear(x))` of reference models/rnn_base.py:462-474 (nn.Linear / EnsembleLinear followed by the
 * per-layer activation module) for act = ELU, and the bias add of EnsembleLinear (ensemble_linear_model.py:60-67).
 * y [rows, C] contiguous, rows grouped in rows / rows_per_seg segments with one bias row [C] each (bias [nseg, C]).
 * fwd (in place): y <- act(y + bias);  act: 0 = identity, 1 = ELU(alpha = 1).
 * bwd (from the forward OUTPUT a): gy = g * act'(.), dbias[nseg, C] = per-segment column sums of gy (dbias may be NULL).  g has row
 *      stride ldg >= C (a column block of a wider gradient is read in place: ABI 7), a has row stride lda >= C (a block of a row buffer
 *      that its producing GEMM wrote in place: ABI 8), gy is dense; gy NULL with act == 0 = column sums only.
 *      workspace: resel_bias_act_bwd_workspace_bytes. */
int resel_bias_act_fwd(float* y, const float* bias, int64_t rows, int C, int64_t rows_per_seg, int act, resel_stream_t stream);
size_t resel_bias_act_bwd_workspace_bytes(int64_t rows, int C, int64_t rows_per_seg);
int resel_bias_act_bwd(const float* g, int64_t ldg, const float* a, int64_t lda, float* gy, float* dbias, void* workspace, int64_t rows, int C,
                       int64_t rows_per_seg, int act, void* amax_gy, unsigned amax_epoch, resel_stream_t stream);

/* ---- ensemble head: last hidden layer's tail + the width-1 output layer of an efc-E MLP ----------------------------
 * The critic head of the reference is `efc-E(H) ELU -> efc-E(1)` (policy_value_models/contextual_sac_value.py via
 * models/rnn_base.py:462-474; EnsembleLinear weight [E, H, 1], bias [E, 1, 1]).  y [E*M, H] is the hidden layer's GEMM output.
 * fwd: a = elu(y + b2[e]) in place, q[e*M + m] = sum_c a[c] * w3[e, c] + b3[e]          (b2, w3: [E, H]; b3: [E] or NULL)
 * bwd: gy = gq[r] * w3[e] * elu'(a), db2[E, H] = sum_m gy, dw3[E, H] = sum_m a * gq     (db3 = sum_m gq is left to the host) */
int resel_ensemble_head_fwd(float* y, const float* b2, const float* w3, const float* b3, float* q, int64_t rows, int H,
                            int64_t rows_per_seg, resel_stream_t stream);
size_t resel_ensemble_head_bwd_workspace_bytes(int64_t rows, int H, int64_t rows_per_seg);
int resel_ensemble_head_bwd(const float* gq, const float* a, const float* w3, float* gy, float* db2, float* dw3, void* workspace,
                            int64_t rows, int H, int64_t rows_per_seg, void* amax_gy, unsigned amax_epoch, resel_stream_t stream);

/* ---- fp32 GEMM on the matrix cores, tails fused --------------------------------------------------------------------------
 * C[b][m][n] = act( sum_k A[b](m, k) B[b](n, k) + bias[b][n] ),  b < batch (ensemble member; strides in floats).
 * An operand is a [rows][K] matrix (x_kcontig = 1: activations, nn.Linear / EnsembleLinear weights [out][in]) or a [K][rows]
 * matrix (x_kcontig = 0: the transposed use of an activation matrix in a weight gradient, of a weight in an input gradient):
 *   forward  y = x W^T (+ bias, ELU)   A = x  (1), B = W (1)      torch.nn.Linear (models/rnn_base.py:101-105),
 *   dgrad    dx = dy W                 A = dy (1), B = W (0)      EnsembleLinear (models/ensemble_linear_model.py:36-49)
 *   wgrad    dW = dy^T x               A = dy (0), B = x (0)      K = number of tokens: split over blocks, partial tiles summed in
 *                                                                  a fixed order (workspace: resel_gemm_f32_workspace_bytes)
 * act: 0 none, 1 ELU, 2 accumulate (C += product + bias: the accumulating form of an input gradient; no activation), 3 softplus (resel_gemm_f32x only).
 * EVERY extent and alignment is taken (round 6; no ABI change): the matrix-core kernels need the contiguous extent of each operand (K or rows) to be a
 * multiple of 4, 16-byte aligned rows and row strides < 2^22; operands that are not - the 6-wide TD3 / discrete heads and their gradients, a rank-2
 * dt_proj, odd action counts - and the M <= 8 rows of a rollout step against a whole weight matrix (models/rnn_base.py single-step branch,
 * smamba/mamba.py:257-305) run csrc/gemm_any.hip: exact fp32 FMAs, the same epilogues, the same magnitude publication, long reductions cut in
 * K and summed in a fixed order.  No call of this entry is left to a vendor library by the host side (tests/test_no_library_gemm_gpu.py).
 * split selects how the fp32 products are formed (inputs, accumulation and outputs are fp32 in every mode):
 *   0  v_mfma_f32_32x32x2_f32 (fp32 operands, exact products);
 *   6  each operand split EXACTLY into three bf16 planes (8 + 8 + 8 significant bits), the six leading plane products - each exact -
 *      accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (the three dropped terms are each <= 2^-24 |a b|, the size of one fp32 rounding
 *      of the product);
 *   3  "bf16x3": two planes per operand (16 significant bits) and the three leading plane products; every dropped term is
 *      <= 2^-16 |a b|.  The class torch names float32 matmul precision 'high' (TF32 / bf16x3) - NOT the reference's default
 *      ('highest' = modes 0 / 2 / 6); half the matrix instructions of mode 6.  Shapes with M <= 128 run mode 6;
 *   2  see resel_gemm_f32x below.
 * (Modes 9 - all nine plane products - and 106 / 109 - modes 6 / 9 on the first-edition kernel - were A/B forms of rounds 2-4; removed
 * in ABI 7.)  Mode 6 splits each operand element once per block on its way into LDS (csrc/gemm_bf3.hip, 256 x 128 tiles); M <= 128
 * takes the first-edition kernel of csrc/gemm_f32.hip (128 x 128 tiles, every wave splits the fragments it reads). */
size_t resel_gemm_f32_workspace_bytes(int M, int N, int K, int batch);
int resel_gemm_f32(const float* A, int64_t lda, int64_t strideA, int a_kcontig,
                   const float* B, int64_t ldb, int64_t strideB, int b_kcontig,
                   const float* bias, int64_t strideBias, int act,
                   float* C, int64_t ldc, int64_t strideC, void* workspace,
                   int M, int N, int K, int batch, int split, resel_stream_t stream);

/* Mode 2 ("f16x3") of the same contract: fp16 planes of the SCALED operands, three plane products on v_mfma_f32_32x32x16_f16,
 * one fp32 accumulator - half the matrix instructions of mode 6.  Each operand is multiplied by a power of two s that brings its
 * largest magnitude into [2^14, 2^15), then A -> a1 = fp16(x s), a2 = fp16(2^11 (x s - a1)); B -> b1 = fp16(x s),
 * b2 = fp16(x s - b1), b1s = 2^-11 b1;  C = (a1 b1 + a1 b2 + a2 b1s) / (sA sB).  The residuals and the plane products are exact in
 * fp32; the dropped term is <= 2^-22 |a b|.  Every element of A keeps 22 significant bits down to 2^-29 max|A|, of B down to
 * 2^-18 max|B| (below: 40 - log2(max|B| / |x|) bits) - pass the weights (forward, input gradient) or the narrower-range operand
 * as B.  Error against fp64 <= the fp32 instruction's for A over 12 decades and B over 6 (tools/eval_f16_split.py,
 * tests/test_hip_ops.py::test_gemm_f32_f16x3_mode_error_against_fp64).  amax_a / amax_b: magnitude HANDLES (below) holding an upper bound of
 * max |A| / max |B| over the whole operand (all batch members) - resel_amax, or what the producer of the operand published; a
 * bound 2^k too large costs k bits of those ranges, one too small overflows fp16 (inf in C).  Other modes ignore the two pointers
 * (may be NULL).  M <= 128 and K < 32 fall back to modes 6 / 0.
 * Two kernels serve mode 2 (same results to rounding of the same three plane products; b1s is formed from b1 next to the matrix
 * instruction in both): the producer / consumer edition for K and K slices that are multiples of 32 and row strides < 2^22
 * (csrc/gemm_bf3.hip `gemm_ws_kernel`), the one-role edition for the rest.  Environment, read once at load: RESEL_GEMM_EDITION=2 sends
 * everything to the one-role edition; =4 sends tall products with A in its [rows][K] form to an experimental 256 x 256 block tile
 * (`gemm_w8_kernel`; measured equal or slower as a whole, profiles/r05_gemm.md 4b). */
int resel_gemm_f32x(const float* A, int64_t lda, int64_t strideA, int a_kcontig,
                    const float* B, int64_t ldb, int64_t strideB, int b_kcontig,
                    const float* bias, int64_t strideBias, int act,
                    float* C, int64_t ldc, int64_t strideC, void* workspace,
                    int M, int N, int K, int batch, int split, const float* amax_a, const float* amax_b,
                    void* amax_c, unsigned amax_epoch, resel_stream_t stream);
/* Fused epilogues of the producer / consumer edition (ABI 7).  Product mode 2 only (magnitude handles required), K a multiple of 32,
 * M > 128, row strides < 2^22; dact also needs M >= 256 and N a multiple of 128 (rows past the last whole 256-row tile take the plain
 * product + one small in-place pass inside the call) - resel_gemm_f32_fused_supported(kind, ...) says whether a shape qualifies; operand
 * layouts as resel_gemm_f32, A in its [rows][K] form.
 * Both write per-wave partial reductions into `workspace` (resel_gemm_f32_fused_workspace_bytes(M, N, K, batch, kind); kind 4 = dact,
 * 5 = head) and fold them with a second small kernel in a fixed order: deterministic, no atomics.
 *
 * resel_gemm_f32_dact - the input gradient of a layer whose input is the ELU output Y of the layer below, with that ELU's derivative and
 *   that layer's bias gradient in the epilogue (reference: rnn_base.py:461-469 applies the activation module after every fc / efc layer;
 *   autograd runs elu_backward and the bias sum as separate passes):
 *     C[b][m][n] = (sum_k A[b](m, k) B[b](n, k)) * (Y[b][m][n] > 0 ? 1 : Y[b][m][n] + 1);   dbias[b][n] = sum_m C[b][m][n]  (dbias may be NULL)
 *   Y: row stride ldy, batch stride strideY, same [m][n] indexing as C.
 * resel_gemm_f32_head - the last two layers of an efc-E critic head (reference ensemble_linear_model.py:36-49 via rnn_base.py:421-469:
 *   efc-E(H) -> ELU -> efc-E(1)) in one GEMM:
 *     C[b][m][n] = a = elu(sum_k A[b](m, k) B[b](n, k) + bias[b][n]);   q[b][m] = sum_n a[b][m][n] w3[b][n] + b3[b]  (b3 may be NULL)
 *   (C is kept: the backward needs the hidden activation). */
int resel_gemm_f32_fused_supported(int kind, int M, int N, int K, int64_t lda, int64_t ldb);
size_t resel_gemm_f32_fused_workspace_bytes(int M, int N, int K, int batch, int kind);
int resel_gemm_f32_dact(const float* A, int64_t lda, int64_t strideA, int a_kcontig,
                        const float* B, int64_t ldb, int64_t strideB, int b_kcontig,
                        const float* Y, int64_t ldy, int64_t strideY,
                        float* C, int64_t ldc, int64_t strideC, float* dbias, void* workspace,
                        int M, int N, int K, int batch, const float* amax_a, const float* amax_b,
                        void* amax_c, unsigned amax_epoch, resel_stream_t stream);
int resel_gemm_f32_head(const float* A, int64_t lda, int64_t strideA, int a_kcontig,
                        const float* B, int64_t ldb, int64_t strideB, int b_kcontig,
                        const float* bias, int64_t strideBias, const float* w3, int64_t strideW3, const float* b3,
                        float* C, int64_t ldc, int64_t strideC, float* q, void* workspace,
                        int M, int N, int K, int batch, const float* amax_a, const float* amax_b,
                        void* amax_c, unsigned amax_epoch, resel_stream_t stream);
/* out [rows, cols] (row stride ld_out) = zeros with nblk <= 8 source blocks copied in: block k = src[k] [nr[k], nc[k]] (row stride ld_src[k]) placed
 * at (r0[k], c0[k]); later blocks win where blocks overlap.  The arrays are HOST arrays of length nblk (read before the call returns).  One launch
 * for the operands of the merged input-encoder GEMM - the block-diagonal of the encoder weights, their concatenated biases, the concatenated
 * zero-padded input rows (reference contextual_sac_value.py:90-99 / contextual_sac_policy.py: one nn.Linear per input and a cat of the outputs).
 * (ABI 8) */
int resel_place_blocks(float* out, int64_t ld_out, int rows, int cols, int nblk, const float* const* src, const int64_t* ld_src,
                       const int* r0, const int* nr, const int* c0, const int* nc, resel_stream_t stream);

/* Magnitude handles.  A kernel that writes a tensor which a later GEMM reads can publish max |x| of what it stored, so that mode 2
 * needs no extra pass over the operand.  A handle is 1 KiB (8-byte aligned; NULL = off): eight 8-byte words 128 bytes apart, word j =
 * {float bits : low 32 | epoch : high 32}; a publishing wave raises the word its workgroup id selects with one 64-bit atomicMax
 * (eight lines: thousands of waves finishing together would serialise on one).  A larger epoch outranks any older content - handles
 * are never zeroed, give every tensor a fresh epoch - and kernels that fill parts of one tensor may share handle and epoch.  Readers
 * (amax_a / amax_b above) take the newest epoch among the eight words and the largest magnitude carrying it.  Publishers:
 * resel_amax (a pre-pass), resel_gemm_f32x (amax_c: the values stored to C), resel_bias_act_bwd (gy), resel_ensemble_head_bwd (gy), resel_add_layernorm_fwd
 * (y: the analytic bound sqrt(C) max|w| + max|b|, published by one wave), resel_add_layernorm_bwd (dx), resel_selective_scan_fwd (out), resel_selective_scan_bwd (dz, ddelta),
 * resel_causal_conv1d_fwd (y), resel_causal_conv1d_bwd (dx), resel_gelu_dropout_fwd / _bwd (y / dx). */
/* Magnitude pre-pass: max |x| over a [batch][rows][cols] box (row stride ld, batch stride `stride`, cols % 4 == 0) written into the
 * magnitude handle `out` with epoch `epoch` (see "magnitude handles" above); one HBM-bound pass, no host synchronisation.
 * The call keeps no state: any number of pre-passes may be in flight on any streams (a maximum is order-independent, every wave raises
 * the handle with one conditional atomicMax).  `state` is ignored (may be NULL) and resel_amax_state_bytes() returns 0 since ABI 7; both
 * stay in the signature so that ABI 6 callers keep linking. */
/* Magnitudes of every tensor of a flat parameter buffer in ONE launch: segment g = flat[begin[g], begin[g] + len[g]) (device int64
 * tables) publishes into the g-th of `nseg` consecutive 1 KiB handles at `handles` under `epoch` (the weights change once per optimizer
 * step / soft update; the GEMMs of the update then find their weight's magnitude without a pre-pass per call). */
int resel_amax_segments(const float* flat, const int64_t* begin, const int64_t* len, int nseg, void* handles, unsigned epoch,
                        resel_stream_t stream);
size_t resel_amax_state_bytes(void);
int resel_amax(const float* x, int64_t ld, int64_t stride, int rows, int cols, int batch, void* out, unsigned epoch, void* state,
               resel_stream_t stream);
/* Verify mode (ABI 7): does `handle` hold an upper bound of max |x| over the same box?  Waves whose maximum exceeds it report into the
 * device words err[0..3] (int32, zero = clean): [0] reporting waves, [1] float bits of the largest violating magnitude, [2] `tag` (!= 0)
 * of the first report, [3] float bits of the bound it saw.  No host synchronisation; the caller reads `err` when it wants the verdict.
 * The Python side runs it in front of every mode-2 product when RESEL_AMAX_VERIFY=1 (hip/ops.py `amax_verify_raise`). */
int resel_amax_check(const float* x, int64_t ld, int64_t stride, int rows, int cols, int batch, const void* handle, int* err, int tag,
                     resel_stream_t stream);

/* ---- mixed-precision GEMM for the bf16 attention projections (cgpt) ------------------------------------------------------
 * C[m][n] = sum_k bf16(A(m, k)) bf16(B(n, k)) + bf16(bias[n]): operands rounded to bf16 (round to nearest even) on their way
 * into LDS, fp32 accumulation (v_mfma_f32_32x32x16_bf16), C stored as bf16 (c_bf16 = 1), as fp32 (0), or as fp32 holding the
 * bf16-rounded value (2: the autocast's rounding point without the cast pass that follows it).  A / B are fp32 or bf16 in
 * memory (x_bf16) and [rows][K] (x_kcontig = 1) or [K][rows]; lda / ldb / ldc in ELEMENTS; bias fp32 or NULL.  This is what
 * F.linear computes under the reference's bf16 autocast (flash-attn MHA, TransformerFlashAttention.py:67-70) without the
 * separate cast passes: forward (A = activations, B = weight [N][K]), input gradient (B = weight as [K][rows]), weight
 * gradient (both operands [K = tokens][rows]; K slices summed in a fixed order, workspace: resel_gemm_bf16_workspace_bytes).
 * The matrix-core kernel needs the contiguous extent of each operand to be a multiple of 4, leading dimensions multiples of 4 and bases 16-byte
 * (fp32) or 8-byte (bf16) aligned; other fp32 operands, and M <= 8 rows against a [N][K] fp32 weight (the decode step of flash-attn's MHA; A fp32 or
 * bf16), run csrc/gemm_any.hip with the same rounding points (round 6; no ABI change). */
size_t resel_gemm_bf16_workspace_bytes(int M, int N, int K);
int resel_gemm_bf16(const void* A, int64_t lda, int a_kcontig, int a_bf16, const void* B, int64_t ldb, int b_kcontig, int b_bf16,
                    const float* bias, void* C, int64_t ldc, int c_bf16, void* workspace, int M, int N, int K,
                    resel_stream_t stream);

/* Bias gradient of such a projection: out[n] = sum_m x[m][n] in fp32 for a bf16 matrix (row stride ld elements, ld % 8 == 0, N % 8 == 0,
 * N <= 2048, 16-byte aligned base) - the `gy.sum(0)` of the reference's autocast nn.Linear backward (TransformerFlashAttention.py:67-70
 * through flash-attn's MHA).  Fixed summation order (per-256-row partials, then their sum): bitwise reproducible. */
size_t resel_colsum_bf16_workspace_bytes(int M, int N);
int resel_colsum_bf16(const uint16_t* x, int64_t ld, int M, int N, float* out, void* workspace, resel_stream_t stream);

/* ---- packed trajectory batch from a device-resident replay ring --------------------------------------------------
 * Device counterpart of NestedMemoryArray.sample_trajs' packing loop (reference buffers/transition_buffer/
 * nested_replay_memory.py:140-176) plus the trainer's flag surgery (algorithm/sac_full_length_rnn_ensembleQ.py:338-342).
 * buffer [capacity, W] fp32 ring; segments [nseg][4] int32 = (batch row, first slot, length incl. `skip` leading slots,
 * first transition index) - the host-side sampling plan; pre_pairs [npairs][2] = (dst column, src column) of the pre-step
 * slot (next_state <- state, reward <- reward_input, state <- last_state of the trajectory's first transition).
 * out [rows, Tp, W + 3]: the W field columns, then validity, the target pass's validity and start flags.
 * A plan entry that does not fit (row >= rows, slot range beyond Tp, transitions beyond `capacity`) is dropped. */
int resel_gather_trajs(const float* buffer, int W, int64_t capacity, const int* segments, int nseg, int max_len, int skip, int rows, int Tp,
                       int c_mask, int c_start, int c_done, int c_timeout, const int* pre_pairs, int npairs,
                       float* out, resel_stream_t stream);

/* ---- one-token rollout step (T = 1): the policy forward between updates -------------------------------------------
 * Reference: the outer loop calls policy.forward once per environment step (algorithm/sac.py:319-326), which reaches
 * Mamba.step (models/smamba/mamba.py:257-305; its GPU branch calls causal_conv1d_update and the Triton
 * selective_state_update, mamba_ssm/ops/triton/selective_state_update.py:21-154) and, for cgpt, flash-attn's MHA with
 * InferenceParams (models/flash_attention/TransformerFlashAttention.py:13-27,76-81; models/rnn_base.py:437-452).
 * All three read the old state and write a NEW state buffer (functional, like the reference's returned hidden), take
 * row strides in elements, and read nothing from the host - so a whole policy step can be captured in a hipGraph.
 *
 * resel_mamba_conv_step: window[b, d, :] <- (window[b, d, 1:], x[b, d]);  xc = act(sum_k taps * w[d, k] + bias[d]) over the
 *   newest K - 1 stored taps + x.  x [B, Di] (row stride ldx); state rows (strides ld_in / ld_out) hold W taps per channel at
 *   [d * stride_d + j * stride_k], oldest first: W = K, strides (K, 1) for smamba's [Di, K] window; W = K - 1, strides
 *   (1, Di) for the time-major [K - 1, Di] tail of the s6 `mamba` and `conv1d` layers (models/s6/mamba.py:166-176,
 *   models/conv1d/conv1d.py:27-37).  w [Di, K], bias [Di] or NULL; act: 1 = SiLU, 0 = none.
 * resel_selective_state_update: dt = softplus(x_db[:, :R] w_dt^T + dt_bias); A = -exp(A_log);
 *   h <- h * exp(dt A) + dt * Bm * xc;  y = sum_n h * Cm + D * xc;  y *= silu(z) if z.
 *   x_db [B, R + 2N] = (dt low-rank | Bm | Cm) (row stride ld_xdb), w_dt [Di, R], state [B, Di, N], z [B, Di] (row stride ldz).
 * resel_attn_decode: appends this token's k, v to kv_cache [Bmax, max_seqlen, 2, H, hd] (bf16) at position `pos` and
 *   returns softmax_j<=pos(scale q.k_j - slope_h (pos - j)) v_j as out [B, H, hd] bf16.  qkv [B, 3, H, hd] bf16 (row stride
 *   ld_qkv).  pos = *pos_dev when pos_dev != NULL (device step counter: graph replay), else pos_host.  hd in {32, 64}.
 *   pos >= max_seqlen: RESEL_EINVAL for a host position; with a device counter the output row is NaN (the reference's
 *   flash-attn asserts on a full cache). */
int resel_mamba_conv_step(const float* x, int64_t ldx, const float* state_in, int64_t ld_in, float* state_out, int64_t ld_out,
                          int64_t stride_d, int64_t stride_k, int W, const float* w, const float* bias, float* xc, int B, int Di,
                          int K, int act, resel_stream_t stream);
int resel_selective_state_update(const float* state_in, int64_t ld_in, float* state_out, int64_t ld_out, const float* xc,
                                 const float* x_db, int64_t ld_xdb, const float* w_dt, const float* dt_bias, const float* A_log,
                                 const float* D, const float* z, int64_t ldz, float* y, int B, int Di, int N, int R,
                                 resel_stream_t stream);
int resel_attn_decode(const uint16_t* qkv, int64_t ld_qkv, uint16_t* kv_cache, const int32_t* pos_dev, int pos_host,
                      const float* slopes, uint16_t* out, float scale, int B, int H, int head_dim, int max_seqlen,
                      resel_stream_t stream);

/* ---- C = W^T N over a very long reduction dimension (weight gradients of the narrow Mamba projections) ----------------
 * Replaces autograd's `grad.t() @ input` of the x_proj / dt_proj F.linear calls (reference
 * models/smamba/mamba_ssm/ops/selective_scan_interface_new.py:261-335; models/smamba/mamba.py:231-233).
 * wide [K, Wd] (row stride ldw, Wd % 4 == 0, 16-byte aligned rows), narrow [K, Nd <= 96] (row stride ldn);
 * out = wide^T narrow as [Wd, Nd], or its transpose [Nd, Wd] when transposed != 0.  fp32 MFMA, the reduction is split over
 * the grid and summed in a fixed order (deterministic).  workspace: resel_atb_workspace_bytes(K, Wd, Nd). */
size_t resel_atb_workspace_bytes(int64_t K, int Wd, int Nd);
int resel_atb(const float* wide, int64_t ldw, int Wd, const float* narrow, int64_t ldn, int Nd, float* out, int transposed,
              void* workspace, int64_t K, resel_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RESEL_HIP_H */
