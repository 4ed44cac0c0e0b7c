/* spa3d.h -- C-ABI of libspa3d_hip.so: the MI355X (gfx950) implementation of the 3DSPA
 * TrackAutoEncoder3D train-step hot path.
 *
 * The reference (TheProParadox/3dspa_code) has no FFI: its boundary is the Flax module
 * protocol.  Each entry point below names the reference call it replaces:
 *
 *   spa3d_create / spa3d_leaf_*     TrackAutoEncoder3D(...) + model.init(rng, batch)['params']
 *                                   track_autoencoder_3d.py:43-115, train.py:221-233
 *   spa3d_encode                    TrackAutoEncoder3D.encode            track_autoencoder_3d.py:190-204
 *   spa3d_decode                    get_decoder_context + decode         track_autoencoder_3d.py:206-307
 *   spa3d_forward                   model.apply({'params': p}, batch)    track_autoencoder_3d.py:309-357
 *   spa3d_loss                      compute_loss_3d                      train.py:96-129
 *   spa3d_loss_and_grads            jax.value_and_grad(loss_fn)(params)  train.py:134-162
 *   spa3d_adamw_step                optax.chain(clip_by_global_norm(1.0), adamw(lr, 0.01)) + apply_updates
 *                                   train.py:164-165,239-242 (intended semantics, repair R6)
 *   spa3d_uniform_noise             jax.random.uniform(PRNGKey(0), shape) track_autoencoder_3d.py:254-257
 *   spa3d_op_*                      single building blocks (attention.py, track_autoencoder.py:18-38),
 *                                   exported so tests can check each kernel against the oracle.
 *
 * Rules: plain C; every function returns an int status (0 = ok) and never throws; no
 * allocation inside (the caller supplies one workspace from its own allocator); every call
 * is asynchronous on the hipStream_t passed as `void* stream` -- except that, in the 16-bit
 * modes, two 4-byte plan counts per sample chunk (kept frame tokens; distinct query frames) are
 * read back to the host, which synchronises the stream at those points (spa3d_set_option(h, "prune", 0) and
 * spa3d_set_option(h, "ro_share", 0) remove the reads together with the savings they size; no compute path reads the environment --
 * the six options may only be PRESET from it when spa3d_create runs);
 * one handle per stream (thread-compatible, not thread-safe).  All tensors are row-major contiguous in the
 * reference's layouts.  Parameters and gradients are ONE flat float32 buffer each whose
 * leaf order / offsets the library defines (spa3d_leaf_*); names are the Flax paths of
 * SURVEY.md 0.3 and shapes are Flax shapes ([in,out] kernels, [in,H,Dh] / [H,Dh,out]).
 */
#ifndef SPA3D_H_
#define SPA3D_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPA3D_OK 0
#define SPA3D_ERR_ARG 1      /* bad argument / unsupported shape           */
#define SPA3D_ERR_WORKSPACE 2 /* workspace too small for even one sample    */
#define SPA3D_ERR_HIP 3      /* a HIP runtime call or launch failed        */

#define SPA3D_F32 0  /* exact-fp32 path: v_mfma_f32_16x16x4_f32, fp32 activations (parity runs) */
#define SPA3D_BF16 1 /* bf16 activations + bf16 MFMA, fp32 accumulate, fp32 master params/grads  */
#define SPA3D_F16 2  /* IEEE fp16 activations + fp16 MFMA (same rate), fp32 accumulate / master params/grads; the 16-bit backward
                        runs at loss x 2^k (k per call from the loss denominator), BASELINE.json configs[4] */

typedef struct spa3d_ctx* spa3d_handle;

/* Hyper-parameters: fields of TrackAutoEncoder3D (track_autoencoder_3d.py:53-67) plus the
 * transformer sizes hard-coded in setup() (:89-112) and the compute precision. */
typedef struct {
  int32_t num_output_frames;   /* 150 */
  int32_t num_latent_tokens;   /* 128 */
  int32_t latent_token_dim;    /* 96  */
  int32_t num_frequencies;     /* 32  */
  float track_scale_factor;    /* 1.0 */
  float time_scale_factor;     /* 150.0 */
  int32_t track_token_dim;     /* 384 */
  int32_t encoder_latent_dim;  /* 512 */
  int32_t decoder_num_channels;/* 1280 */
  int32_t dino_feature_dim;    /* 768; 0 = no dino_projection leaf (use_dino False / key absent) */
  int32_t depth_feature_dim;   /* channels of depth_features actually fed; 0 = no depth_projection */
  int32_t num_heads;           /* 8 */
  int32_t qkv_size;            /* 768 */
  int32_t enc_mlp, enc_layers; /* 1536, 3 */
  int32_t t2l_mlp, t2l_layers; /* 2048, 4 */
  int32_t dec_mlp, dec_layers; /* 2048, 4 */
  int32_t ro_mlp, ro_layers;   /* 1536, 4 */
  int32_t precision;           /* SPA3D_F32 | SPA3D_BF16 | SPA3D_F16 */
  int32_t model_kind;          /* 0 = TrackAutoEncoder3D (track_autoencoder_3d.py:43-357);
                                  1 = the 2-D TRAJAN twin TrackAutoEncoder (track_autoencoder.py:117-390): 2 coordinates, no readout
                                      token, visible-mean pooling, certainty head; dino/depth dims must be 0.  Tensors then carry 2
                                      coordinates ([..,2] tracks, [B,Q,3] query points) */
} spa3d_config;

/* One batch: TrackAutoEncoder3DInputs (track_autoencoder_3d.py:23-40) + the loss targets
 * read from the same dict (train.py:99-100).  Device pointers. */
typedef struct {
  int32_t B, N, Q, T;
  const float* support_tracks;          /* [B,N,T,3] f32 */
  const float* support_tracks_visible;  /* [B,N,T,1] f32, 0/1 */
  const float* query_points;            /* [B,Q,4] f32 (t,x,y,z); required (host builds the default grid) */
  const int32_t* boundary_frame;        /* [B] */
  const void* dino_features;            /* [B,N,T,dino_feature_dim] or NULL; f32 in F32 mode, bf16 in BF16 mode */
  const void* depth_features;           /* [B,N,T,depth_feature_dim] or NULL; same dtype rule */
  const float* noise;                   /* [B,L,latent_token_dim] uniform [0,1) or NULL */
  int32_t discretize;                   /* decode(discretize=...); with noise==NULL the library draws
                                           spa3d_uniform_noise (legacy threefry layout) itself */
  const float* query_tracks;            /* [B,Q,T,3] f32 targets (loss entry points only) */
  const float* query_tracks_visible;    /* [B,Q,T,1] f32 targets */
} spa3d_batch;

/* TrackAutoEncoderResults (track_autoencoder.py:72-91); certain_logits is identically 0. */
typedef struct {
  float* tracks;          /* [B,Q,T_out,3] f32 */
  float* visible_logits;  /* [B,Q,T_out,1] f32 */
  float* certain_logits;  /* [B,Q,T_out,1] f32 or NULL */
  float* latents;         /* [B,L,latent_token_dim] f32 or NULL: encode() output */
} spa3d_outputs;

const char* spa3d_version(void);
int spa3d_create(const spa3d_config* cfg, spa3d_handle* out);
int spa3d_destroy(spa3d_handle h);
const char* spa3d_last_error(spa3d_handle h);

/* parameter tree */
int64_t spa3d_param_elems(spa3d_handle h);
int32_t spa3d_num_leaves(spa3d_handle h);
/* name: >=160 bytes; shape: >=4 int64; offset in floats into the flat buffer */
int spa3d_leaf_info(spa3d_handle h, int32_t i, char* name, int32_t* ndim, int64_t* shape, int64_t* offset);

/* Bytes of workspace needed to process `chunk` samples at a time (1 <= chunk <= B) of a
 * [B,N,Q,T] batch; train!=0 sizes forward+backward, else forward only. */
int64_t spa3d_workspace_bytes(spa3d_handle h, int32_t B, int32_t N, int32_t Q, int32_t T, int32_t chunk, int32_t train);

int spa3d_encode(spa3d_handle h, const float* params, const spa3d_batch* b, float* latents,
                 void* ws, int64_t ws_bytes, void* stream);
int spa3d_decode(spa3d_handle h, const float* params, const spa3d_batch* b, const float* latents,
                 spa3d_outputs* out, void* ws, int64_t ws_bytes, void* stream);
int spa3d_forward(spa3d_handle h, const float* params, const spa3d_batch* b, spa3d_outputs* out,
                  void* ws, int64_t ws_bytes, void* stream);

/* loss3 (device, >= 12 floats, 8-byte aligned: [0..2] = total, position, visible; the rest is scratch -- [3] a sticky flag word
 * that any non-finite partial sum sets, after which all three results are NaN).
 * denom<=0: use max(sum(visible),1) of this batch.  The batch sums are order-independent (64-bit fixed-point accumulation): the same
 * inputs give the same bits on every run and on every data-parallel replica. */
int spa3d_loss(spa3d_handle h, const spa3d_batch* b, const spa3d_outputs* preds, float denom,
               float* loss3, void* stream);

/* forward + loss + backward.  grads (flat f32, same layout as params) is OVERWRITTEN unless
 * accumulate!=0.  denom: global sum(query_tracks_visible) for data-parallel runs (the loss
 * normalisers are batch-global, train.py:111-113,119-121); <=0 = this batch's own.
 * loss3 (device): {total, position, visible} with that denominator.  out may be NULL. */
int spa3d_loss_and_grads(spa3d_handle h, const float* params, const spa3d_batch* b, float denom,
                         float* grads, int32_t accumulate, float* loss3, spa3d_outputs* out,
                         void* ws, int64_t ws_bytes, void* stream);

/* clip_by_global_norm(clip) -> adamw(b1,b2,eps,wd) -> apply_updates on flat buffers, in place.
 * step = number of spa3d_adamw_step calls BEFORE this one; the bias correction uses step + 1 - scratch[3] (updates actually applied:
 * a skipped step leaves m and v untouched, so it does not count).
 * The global norm is a fixed-order two-stage reduction (no float atomics): bit-identical gradients give bit-identical updates on every replica.
 * scratch: >= 4 KiB device, zero-initialised once by the caller and then left alone between steps (floats [256, 768) are per-call partial sums;
 * a multiplier in [4] that is not a power of two in [2^-24, 1] reads as 1): scratch[0] returns the global grad norm,
 * [1] is internal, [2] = 1 when THIS step was skipped because the norm was inf/NaN (an fp16 overflow; params, m, v unchanged) else 0,
 * [3] counts skipped steps, [4] / [5] hold the dynamic loss-scale multiplier and its good-step counter (spa3d_set_loss_scale_state). */
int spa3d_adamw_step(float* params, const float* grads, float* m, float* v, int64_t n, float lr,
                     int64_t step, float clip, float b1, float b2, float eps, float wd,
                     float* scratch, void* stream);

/* Per-handle switches -- the only ones the library has (seven + one test mode); each may be preset at spa3d_create from the environment variable of the same name in
 * capitals with an SPA3D_ prefix (SPA3D_PRUNE ...).  Unknown names return SPA3D_ERR_ARG.
 *   "prune"      0/1  token pruning of the track encoder (16-bit modes)            } with both 0 every entry point is fully asynchronous
 *   "ro_share"   0/1  shared latent rows of the first readout block (16-bit modes) } (no plan count is read back)
 *   "loss_scale"      SPA3D_F16 handles: > 0 fixed, < 0 automatic with that head-gradient target
 *   "chunk"           samples processed at a time; 0 = as many as fit the workspace
 *   "gemm_impl"       0 product dispatch | 1 generic strided MFMA kernel only | 2 tiled kernels | diagnostics that put small problems on the big
 *                     kernels: 3 every eligible GEMM on the 8-phase kernels, 4 the same with the non-persistent 128x384 kernel, 5 without the
 *                     single-buffer short-K kernel, 6 tiled GEMMs without the round-4 / round-5 kernels (MLP forward as two GEMMs, multi-pass input embedding, no row-stationary K = 384 kernel,
 *                     no large-register-tile kernels), 8 the product dispatch without the round-5 large-register-tile kernels (dW: csrc/gemm_tnb.hip; NT: csrc/gemm_ntb.hip), 9 = 3 with every
 *                     eligible dW / NT GEMM on those kernels whatever its row count.  (The `impl` argument of spa3d_op_linear* takes
 *                     the same values; spa3d_op_linear: | 16 = also write the pre-activation, the MLP-in form of the step.)
 *   "attn_impl"       0 product dispatch | 1 generic composition | 2 fused kernels | 3, 4 fused with the split-pass backward on 4 / 8 waves (tests) |
 *                     6 fused kernels with the track encoder's QKV projection + attention forward as one launch (built in round 5, slower than the pair: opt-in)
 *   "det_grads"  0/1  order-independent parameter gradients: every reduction into the gradient buffer (split-M dW tiles, bias / scale column sums, broadcast
 *                     gradients) adds 64-bit fixed-point integers (2^-32 units) into a shadow of the buffer instead of float atomics, so two runs -- and two
 *                     data-parallel schedules -- give the same bits.  Costs 8 bytes of workspace per parameter and ~3.6 % of the step at BASELINE configs[2]
 *                     (1.82 -> 1.88 s: 64-bit atomics in the dW epilogues, the 1-channel depth gradient on the GEMM path); off by default.
 * and one test mode: "poison" 0/1 -- the workspace is filled with 16-bit NaN patterns before every sample chunk, so a read of a row that this
 * call has not written (the rounded-up tails of pruned GEMMs, chunk-to-chunk reuse of the bump allocator) shows up as NaN instead of as a
 * plausible stale value (tests/test_gpu_poison.py). */
int spa3d_set_option(spa3d_handle h, const char* name, double value);

/* Dynamic loss scaling of the SPA3D_F16 backward.  `state` (device, caller-owned, may be NULL to detach) is one float: a power-of-two
 * multiplier in (0,1] applied on top of the per-call scale (0 reads as 1).  Point it at scratch + 4 of spa3d_adamw_step: that call skips the
 * update when the gradient norm is not finite (parameters and moments untouched), halves the multiplier, and doubles it back (up to 1)
 * after 200 finite steps. */
int spa3d_set_loss_scale_state(spa3d_handle h, const float* state);

/* Overlapping the data-parallel gradient all-reduce with the backward (no reference counterpart: the reference is single-device, SURVEY 2;
 * the split is SURVEY 8(e)'s).  Parameter gradients accumulate over the call's sample chunks, so a leaf is final only in the LAST chunk's
 * backward, in reverse graph order.  spa3d_grad_segments: bounds4 = {0, b1, b2, n} (floats) cut the flat gradient buffer into
 *   [b2, n)  track_readout_attn, query_encoder, track_predictor            -- final first  (event ev_readout)
 *   [b1, b2) tracks_to_latents, compressor, decompressor, decompress_attn  -- final second (event ev_latents)
 *   [0, b1)  state_init leaves, token / dino / depth projections, input_track_transformer -- final when the call's work is.
 * spa3d_set_grad_events registers two hipEvent_t (or NULL to detach) that spa3d_loss_and_grads records on its stream at those two points of
 * the last chunk; a collective on a side stream that waits for an event may then run under the rest of the backward.  Nothing is recorded
 * on SPA3D_F16 handles (the loss-scaled buffer is rescaled as a whole at the end): use stream order there. */
int spa3d_grad_segments(spa3d_handle h, int64_t* bounds4);
int spa3d_set_grad_events(spa3d_handle h, void* ev_readout, void* ev_latents);
/* out4 = {records of the readout event so far, records of the latents event so far, the registered ev_readout, the registered ev_latents}
 * (host counters, bumped when a record is enqueued; the pointers as integers).  A caller about to wait on ITS events checks that they are
 * still the registered ones and that both counters advanced during its last spa3d_loss_and_grads, and otherwise falls back to stream order: waiting on an event that was NOT re-recorded (another owner re-registered or detached the events, an SPA3D_F16 handle)
 * would let the side-stream all-reduce start before the backward has produced the gradients. */
int spa3d_grad_events_recorded(spa3d_handle h, int64_t* out4);
/* Owner-aware detach: clears the registered events only if the handle still holds exactly (ev_readout, ev_latents), and the loss-scale
 * state only if it still is `loss_scale_state`; NULL arguments are skipped.  Lets a train state that is being destroyed release what IT
 * registered without tearing down what a newer state has registered on the same handle since. */
int spa3d_detach(spa3d_handle h, void* ev_readout, void* ev_latents, const float* loss_scale_state);

/* Data-dependent plan sizes of the last spa3d_loss_and_grads / forward call on this handle, summed over its sample chunks:
 * out4 = {track-encoder token rows kept, token rows before pruning, distinct (sample, query frame) slots, queries}; zeros when the
 * respective saving is off.  Host values (they are the counts the call read back to size its launches). */
int spa3d_plan_stats(spa3d_handle h, double* out4);

/* Live timing of the hot kernel classes with HIP event pairs recorded on the launch stream (bench.py's
 * "roofline" object).  cls: 0 tiled NT GEMM (incl. the fused MLP forward), 1 tiled TN GEMM (dW), 2 generic GEMM, 3 fused attention fwd,
 * 4 fused attention bwd, 5 / 6 LayerNorm fwd / bwd, 7 single-query attention, 8 the input-embedding stage as a whole (its GEMMs are also
 * counted in class 0).  out4 = {launches, total ms, algorithmic FLOPs, algorithmic bytes}. */
int spa3d_prof_enable(spa3d_handle h, int32_t on);
int spa3d_prof_read(spa3d_handle h, int32_t cls, double* out4);
/* One CSV line per profiled launch group since spa3d_prof_enable(h, 1): class, ms, algorithmic FLOPs, algorithmic bytes, M, N, K, flags
 * (tools/step_shapes.py aggregates it per GEMM shape).  Synchronises on the recorded events. */
int spa3d_prof_dump(spa3d_handle h, const char* path);

/* jax.random.uniform(PRNGKey(0),[n]) in the legacy (non-partitionable) threefry layout. */
int spa3d_uniform_noise(float* out, int64_t n, uint32_t key0, uint32_t key1, void* stream);

/* ---- single-op entry points (tests / benchmarks).  dtype: SPA3D_F32 | SPA3D_BF16 for the
 * activation tensors (void*); parameters, statistics and gradients of parameters are f32. ---- */

/* out[rows, C*2*nf] = SinusoidalEmbedding(x[rows,C]) (track_autoencoder.py:18-38) */
int spa3d_op_sin_embed(const float* x, int64_t rows, int32_t C, int32_t nf, void* out, int32_t dtype, void* stream);

/* C[M,N] = act(A[M,K] @ B[K,N] + bias) (+ residual); A,B,C,residual dense row-major `dtype`;
 * act: 0 none, 1 tanh-gelu, 2 = no activation and `residual` is not added but holds a pre-activation: C = (A @ B + bias) o gelu'(residual), the MLP backward's
 * dh = (dy . W_out^T) o gelu'(hpre).  impl: 0 auto, 1 generic kernel, 2 tiled bf16 kernel (error if unusable), ... as "gemm_impl" above; 7 = the row-stationary
 * K = 384 kernel (csrc/gemm_rs.hip: 16-bit, 128 | N <= 2304, act 0 or 2) or an error; 10 = the large-register-tile NT kernel (csrc/gemm_ntb.hip: 16-bit,
 * 384 | N or 256 | N, 32 | K >= 64, act 0, no residual) or an error -- spa3d_op_linear_bwd: its dA on that kernel. */
int spa3d_op_linear(const void* A, const void* B, const float* bias, const void* residual, void* C,
                    int64_t M, int32_t N, int32_t K, int32_t act, int32_t dtype, int32_t impl,
                    void* ws, int64_t ws_bytes, void* stream);
/* dA[M,K] = dC @ B^T ; dB[K,N] (f32, overwritten) = A^T @ dC ; dbias[N] (f32, overwritten) = colsum(dC) */
int spa3d_op_linear_bwd(const void* A, const void* B, const void* dC, void* dA, float* dB, float* dbias,
                        int64_t M, int32_t N, int32_t K, int32_t dtype, int32_t impl,
                        void* ws, int64_t ws_bytes, void* stream);

/* The MLP of ImprovedTransformerBlock (attention.py:103-108) as ONE sequence-resident kernel, 16-bit dtypes, d = 384 and mlp = 1536 only
 * (the track encoder's widths; anything else returns SPA3D_ERR_ARG):  hpre = na @ w_in + b_in, h = tanh-gelu(hpre), y = a + h @ w_out + b_out.
 * na, a, y [M,d]; h, hpre [M,mlp]; w_in [d,mlp], w_out [mlp,d] in `dtype`; biases f32.  ws >= 4 MiB. */
int spa3d_op_mlp_fused(const void* na, const void* a, const void* w_in, const float* b_in, const void* w_out, const float* b_out,
                       void* y, void* h, void* hpre, int64_t M, int32_t d, int32_t mlp, int32_t dtype,
                       void* ws, int64_t ws_bytes, void* stream);

/* y = LayerNorm(x)*scale (no bias, eps 1e-6, fast variance); stats[rows,2] = (mean, rstd) */
int spa3d_op_layernorm(const void* x, const float* scale, void* y, float* stats, int64_t rows, int32_t d,
                       int32_t dtype, void* stream);
int spa3d_op_layernorm_bwd(const void* x, const float* scale, const float* stats, const void* dy, void* dx,
                           float* dscale /* f32[d], accumulated into */, int64_t rows, int32_t d, int32_t dtype, void* stream);

/* QKV projection + attention core of ImprovedMHDPAttention as ONE kernel (attention.py:154-175; round 5): q | k | v = nq . (Wq | Wk | Wv) (16-bit, stored:
 * qkv [rows, 3*H*96]), per-head RMSNorm of q and k, softmax, PV -> o [rows, H*96], lse [nseq, H, S, 2] (may be NULL).  nq [rows, 384] with row stride ldn;
 * wq / wk / wv [384, H*96] in the activation type; d = 384 and Dh = 96 only, S <= 160.  seq_off (int32 [nseq + 1], device) = ragged sequences (token pruning) or
 * NULL for nseq dense sequences of S rows; keymask as spa3d_op_attention.  16-bit dtypes only; SPA3D_ERR_ARG when the shape is not covered. */
int spa3d_op_qkv_attention(const void* nq, int64_t ldn, const void* wq, const void* wk, const void* wv, const float* scale_q, const float* scale_k,
                           const float* keymask, const int32_t* seq_off, int64_t nseq, int32_t S, int32_t H, void* qkv, void* o, float* lse,
                           int32_t dtype, void* ws, int64_t ws_bytes, void* stream);

/* Multi-head attention core of ImprovedMHDPAttention (attention.py:166-175): per-head RMSNorm of
 * q and k (scales f32[Dh]), q/sqrt(Dh), key mask (f32 [nseq,Sk] or NULL, non-zero = keep),
 * softmax, PV.  q [nseq,Sq,H*Dh] with row stride ldq (elements); k,v [nseq,Sk,H*Dh] strides ldk/ldv;
 * o [nseq,Sq,H*Dh] dense.  impl: 0 auto, 1 generic (GEMM+softmax kernels), 2 fused kernel. */
int spa3d_op_attention(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldk, int64_t ldv,
                       const float* scale_q, const float* scale_k, const float* keymask,
                       int64_t nseq, int32_t Sq, int32_t Sk, int32_t H, int32_t Dh, void* o,
                       float* lse /* [nseq,H,Sq,2] = (row max, log row sum); written by the fused kernel only; may be NULL */,
                       int32_t dtype, int32_t impl, void* ws, int64_t ws_bytes, void* stream);
/* gradients of the above: dq,dk,dv dense-strided like q,k,v (same ld*); dscale_q/k f32[Dh] accumulated into */
int spa3d_op_attention_bwd(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldk, int64_t ldv,
                           const float* scale_q, const float* scale_k, const float* keymask,
                           int64_t nseq, int32_t Sq, int32_t Sk, int32_t H, int32_t Dh,
                           const void* o, const float* lse /* forward outputs: needed by the fused kernel, NULL -> generic */,
                           const void* d_o, void* dq, void* dk, void* dv, float* dscale_q, float* dscale_k,
                           int32_t dtype, int32_t impl, void* ws, int64_t ws_bytes, void* stream);

/* ---- feature producers that build the hot path's inputs.  Host pointers: `intrinsics` only. ----
 * Float32 arithmetic in the reference's operation order: bit-identical to the reference functions under NumPy >= 2. */

/* sample_dino_features_for_tracks (inference.py:339-395): feat [T,Hp,Wp,D] f32, tracks_2d [N,T,2] f32 in pixels of an
 * H x W video -> out [N,T,D] (out_dtype SPA3D_F32 as the reference, or SPA3D_BF16 = the same values rounded once). */
int spa3d_op_sample_dino(const float* feat, const float* tracks_2d, int32_t N, int32_t T, int32_t Hp, int32_t Wp, int32_t D,
                         int32_t H, int32_t W, void* out, int32_t out_dtype, void* stream);
/* sample_depth_features_for_tracks (inference.py:398-447): depth [T,H,W,1] f32 -> out [N,T,256] f32 */
int spa3d_op_sample_depth_features(const float* depth, const float* tracks_2d, int32_t N, int32_t T, int32_t H, int32_t W,
                                   float* out, void* stream);
/* lift_2d_to_3d (inference.py:287-336): intrinsics = host double[4] {fx,fy,cx,cy} or NULL (fx=fy=max(H,W), cx=W/2, cy=H/2) */
int spa3d_op_lift_2d_to_3d(const float* tracks_2d, const float* depth, int32_t N, int32_t T, int32_t H, int32_t W,
                           const double* intrinsics, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPA3D_H_ */
