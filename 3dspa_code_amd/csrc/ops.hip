// ops.hip -- single-op C-ABI entry points (spa3d_op_*): each building block of the hot path callable on
// its own so tests/ can compare every kernel with the oracle.  Thin wrappers over the same kernels the
// model orchestration uses.
#include <cmath>

#include "common.hpp"

namespace SPA_NS {
namespace {
struct OpCtx : spa3d_ctx {
  OpCtx(void* stream_, void* ws, int64_t ws_bytes) {  // (no environment switches: kernel choice comes from the entry point's `impl` argument)
    stream = (hipStream_t)stream_; ar.base = (char*)ws; ar.cap = ws_bytes;
  }
  template <typename U> U* alloc(int64_t n) { return (U*)ar.alloc(n * (int64_t)sizeof(U)); }
  int status() { return ar.overflow ? SPA3D_ERR_WORKSPACE : (hip_err ? SPA3D_ERR_HIP : SPA3D_OK); }
};

template <typename T>
int op_linear(OpCtx& c, const T* A, const T* B, const float* bias, const T* res, T* C, int64_t M, int N, int K, int act, int impl) {
  GemmDesc d{};
  d.A = A; d.B = B; d.C = C; d.M = M; d.N = N; d.K = K; d.sAm = K; d.sAk = 1; d.sBk = N; d.sBn = 1; d.sCm = N;
  d.bias = bias; d.epi = act == 1 ? EPI_GELU : (act == 2 ? EPI_MUL_GELU_GRAD : EPI_NONE); d.aux = res;  // act 2: C = (A.B + bias) o gelu'(residual)
  apply_gemm_impl(&c, impl & 15);
  if constexpr (sizeof(T) == 2) {
    if ((impl & 15) == 7) {  // the row-stationary K = 384 kernel (gemm_rs.hip) or an error
      if (act == 1 || (act == 0 && res) || (act == 2 && !res) || !gemm_rs_ok(K, N)) return SPA3D_ERR_ARG;
      T* pk = c.alloc<T>(gemm_rs_pack_elems(N));
      if (c.ar.overflow) return SPA3D_ERR_WORKSPACE;
      gemm_rs_pack<T>(&c, B, N, 1, N, pk);
      return gemm_rs(&c, A, K, pk, bias, C, N, M, N, act == 2 ? res : nullptr, N) ? c.status() : SPA3D_ERR_ARG;
    }
    if ((impl & 15) == 10) {  // the large-register-tile NT kernel (gemm_ntb.hip) or an error
      if (act != 0 || res || !gemm_ntb_ok(K, N)) return SPA3D_ERR_ARG;
      T* pk = c.alloc<T>(gemm_ntb_pack_elems(K, N));
      if (c.ar.overflow) return SPA3D_ERR_WORKSPACE;
      gemm_ntb_pack<T>(&c, B, N, 1, K, N, pk);
      return gemm_ntb(&c, A, K, pk, bias, C, N, M, N, K) ? c.status() : SPA3D_ERR_ARG;
    }
    if (impl != 1) {
      // impl | 16 (benchmarks): the MLP-in form of the step -- a second output stream (the pre-activation) from the same epilogue
      if (impl & 16) { d.pre_out = c.alloc<T>(M * N); if (c.ar.overflow) return SPA3D_ERR_WORKSPACE; }
      T* Bt = c.alloc<T>((int64_t)K * N);
      if (c.ar.overflow) return SPA3D_ERR_WORKSPACE;
      k_transpose<T>(&c, B, K, N, Bt);
      d.Bt = Bt; d.ldBt = K;
      if (gemm_nt_bf16(&c, d)) return c.status();
      if ((impl & 15) >= 2) return SPA3D_ERR_ARG;
    }
  } else if ((impl & 15) >= 2) return SPA3D_ERR_ARG;
  gemm_generic<T>(&c, d);
  return c.status();
}

template <typename T>
int op_linear_bwd(OpCtx& c, const T* A, const T* B, const T* dC, T* dA, float* dB, float* dbias, int64_t M, int N, int K, int impl) {
  apply_gemm_impl(&c, impl);
  if (dA) {  // dA[M,K] = dC[M,N] . B[K,N]^T
    GemmDesc d{};
    d.A = dC; d.B = B; d.C = dA; d.M = M; d.N = K; d.K = N; d.sAm = N; d.sAk = 1; d.sBk = 1; d.sBn = N; d.sCm = K;
    d.Bt = B; d.ldBt = N;
    bool done = false;
    if constexpr (sizeof(T) == 2) {
      if (impl == 10) {  // dA on the large-register-tile NT kernel (gemm_ntb.hip) or an error: element (k' = n, n' = k) of B^T is B[k * N + n]
        if (!gemm_ntb_ok(N, K)) return SPA3D_ERR_ARG;
        T* pk = c.alloc<T>(gemm_ntb_pack_elems(N, K));
        if (c.ar.overflow) return SPA3D_ERR_WORKSPACE;
        gemm_ntb_pack<T>(&c, B, 1, N, N, K, pk);
        if (!gemm_ntb(&c, dC, N, pk, nullptr, dA, K, M, K, N)) return SPA3D_ERR_ARG;
        done = true;
      } else if (impl != 1) done = gemm_nt_bf16(&c, d);
    }
    if (!done) { if (impl >= 2) return SPA3D_ERR_ARG; gemm_generic<T>(&c, d); }
  }
  if (dB) {  // dB[K,N] = A[M,K]^T . dC[M,N]
    k_zero(&c, dB, (int64_t)K * N * 4);
    GemmDesc d{};
    d.A = A; d.B = dC; d.C = dB; d.M = K; d.N = N; d.K = M; d.sAm = 1; d.sAk = K; d.sBk = N; d.sBn = 1; d.sCm = N;
    d.out_f32 = 1; d.accumulate = 1;
    void* zp = c.alloc<char>(256); k_zero(&c, zp, 256); d.zero_page = zp;
    bool done = false;
    if constexpr (sizeof(T) == 2) { if (impl != 1) done = gemm_tn_bf16(&c, d); }
    if (!done) { if (impl >= 2) return SPA3D_ERR_ARG; gemm_generic<T>(&c, d); }
  }
  if (dbias) { k_zero(&c, dbias, (int64_t)N * 4); k_colsum<T>(&c, dC, M, N, N, dbias); }
  return c.status();
}
}  // namespace

// attention front ends live in attention.hip
template <typename T>
void attention_fwd(spa3d_ctx* c, const T* q, const T* k, const T* v, int64_t ldq, int64_t ldk, int64_t ldv, const float* sq,
                   const float* sk, const float* km, int64_t nseq, int Sq, int Sk, int H, int Dh, T* o, float* lse, int impl,
                   const int32_t* seq_off = nullptr, int64_t total_rows = 0);
template <typename T>
void attention_bwd(spa3d_ctx* c, const T* q, const T* k, const T* v, int64_t ldq, int64_t ldk, int64_t ldv, const float* sq,
                   const float* sk, const float* km, int64_t nseq, int Sq, int Sk, int H, int Dh, const T* o, const float* lse, const T* d_o,
                   T* dq, T* dk, T* dv, float* dsq, float* dsk, int impl, const int32_t* seq_off = nullptr, int64_t total_rows = 0);
}  // namespace SPA_NS

// The entry points below are compiled twice: as declared in include/spa3d.h (this build's 16-bit type is bf16) and, with
// -DSPA_F16=1, under a hidden `_f16` suffix with fp16 as the 16-bit type; the public functions forward dtype == SPA3D_F16 there.
extern "C" {
#pragma GCC visibility push(hidden)
int spa3d_op_sin_embed_f16(const float* x, int64_t rows, int32_t C, int32_t nf, void* out, int32_t dtype, void* stream);
int spa3d_op_linear_f16(const void* A, const void* B, const float* bias, const void* residual, void* C, int64_t M, int32_t N, int32_t K,
                        int32_t act, int32_t dtype, int32_t impl, void* ws, int64_t ws_bytes, void* stream);
int spa3d_op_linear_bwd_f16(const void* A, const void* B, const void* dC, void* dA, float* dB, float* dbias, int64_t M, int32_t N, int32_t K,
                            int32_t dtype, int32_t impl, void* ws, int64_t ws_bytes, void* stream);
int spa3d_op_mlp_fused_f16(const void* na, const void* a, const void* w_in, const float* b_in, const void* w_out, const float* b_out, void* y,
                           void* h, void* hpre, int64_t M, int32_t d, int32_t mlp, int32_t dtype, void* ws, int64_t ws_bytes, void* stream);
int spa3d_op_qkv_attention_f16(const void* nq, int64_t ldn, const void* wq, const void* wk, const void* wv, const float* scale_q, const float* scale_k,
                               const float* keymask, const int32_t* seq_off, int64_t nseq, int32_t S, int32_t H, void* qkv, void* o, float* lse,
                               int32_t dtype, void* ws, int64_t ws_bytes, void* stream);
int spa3d_op_layernorm_f16(const void* x, const float* scale, void* y, float* stats, int64_t rows, int32_t d, int32_t dtype, void* stream);
int spa3d_op_layernorm_bwd_f16(const void* x, const float* scale, const float* stats, const void* dy, void* dx, float* dscale, int64_t rows,
                               int32_t d, int32_t dtype, void* stream);
int spa3d_op_attention_f16(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldk, int64_t ldv, const float* scale_q,
                           const float* scale_k, const float* keymask, int64_t nseq, int32_t Sq, int32_t Sk, int32_t H, int32_t Dh, void* o,
                           float* lse, int32_t dtype, int32_t impl, void* ws, int64_t ws_bytes, void* stream);
int spa3d_op_attention_bwd_f16(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldk, int64_t ldv, const float* scale_q,
                               const float* scale_k, const float* keymask, int64_t nseq, int32_t Sq, int32_t Sk, int32_t H, int32_t Dh,
                               const void* o, const float* lse, const void* d_o, void* dq, void* dk, void* dv, float* dscale_q,
                               float* dscale_k, int32_t dtype, int32_t impl, void* ws, int64_t ws_bytes, void* stream);
#pragma GCC visibility pop
}
#if SPA_F16
#define spa3d_op_sin_embed spa3d_op_sin_embed_f16
#define spa3d_op_linear spa3d_op_linear_f16
#define spa3d_op_linear_bwd spa3d_op_linear_bwd_f16
#define spa3d_op_layernorm spa3d_op_layernorm_f16
#define spa3d_op_mlp_fused spa3d_op_mlp_fused_f16
#define spa3d_op_qkv_attention spa3d_op_qkv_attention_f16
#define spa3d_op_layernorm_bwd spa3d_op_layernorm_bwd_f16
#define spa3d_op_attention spa3d_op_attention_f16
#define spa3d_op_attention_bwd spa3d_op_attention_bwd_f16
#define FWD16(call)
#else
#define FWD16(call) if (dtype == SPA3D_F16) return call;
#endif

extern "C" {

int spa3d_op_sin_embed(const float* x, int64_t rows, int32_t C, int32_t nf, void* out, int32_t dtype, void* stream) {
  FWD16(spa3d_op_sin_embed_f16(x, rows, C, nf, out, dtype, stream))
  if (!x || !out || nf <= 0 || nf > 64) return SPA3D_ERR_ARG;
  OpCtx c(stream, nullptr, 0);
  if (dtype == SPA3D_F32) k_sin_embed<float>(&c, x, rows, C, nf, 1.0f, (float*)out);
  else k_sin_embed<bf16_t>(&c, x, rows, C, nf, 1.0f, (bf16_t*)out);
  return c.status();
}

int spa3d_op_linear(const void* A, const void* B, const float* bias, const void* residual, void* C, int64_t M, int32_t N, int32_t K,
                    int32_t act, int32_t dtype, int32_t impl, void* ws, int64_t ws_bytes, void* stream) {
  FWD16(spa3d_op_linear_f16(A, B, bias, residual, C, M, N, K, act, dtype, impl, ws, ws_bytes, stream))
  if (!A || !B || !C) return SPA3D_ERR_ARG;
  OpCtx c(stream, ws, ws_bytes);
  if (dtype == SPA3D_F32) return op_linear<float>(c, (const float*)A, (const float*)B, bias, (const float*)residual, (float*)C, M, N, K, act, impl);
  return op_linear<bf16_t>(c, (const bf16_t*)A, (const bf16_t*)B, bias, (const bf16_t*)residual, (bf16_t*)C, M, N, K, act, impl);
}
int spa3d_op_linear_bwd(const void* A, const void* B, const void* dC, void* dA, float* dB, float* dbias, int64_t M, int32_t N, int32_t K,
                        int32_t dtype, int32_t impl, void* ws, int64_t ws_bytes, void* stream) {
  FWD16(spa3d_op_linear_bwd_f16(A, B, dC, dA, dB, dbias, M, N, K, dtype, impl, ws, ws_bytes, stream))
  if (!A || !B || !dC) return SPA3D_ERR_ARG;
  OpCtx c(stream, ws, ws_bytes);
  if (dtype == SPA3D_F32) return op_linear_bwd<float>(c, (const float*)A, (const float*)B, (const float*)dC, (float*)dA, dB, dbias, M, N, K, impl);
  return op_linear_bwd<bf16_t>(c, (const bf16_t*)A, (const bf16_t*)B, (const bf16_t*)dC, (bf16_t*)dA, dB, dbias, M, N, K, impl);
}

int spa3d_op_mlp_fused(const void* na, const void* a, const void* w_in, const float* b_in, const void* w_out, const float* b_out, void* y,
                       void* h, void* hpre, int64_t M, int32_t d, int32_t mlp, int32_t dtype, void* ws, int64_t ws_bytes, void* stream) {
  FWD16(spa3d_op_mlp_fused_f16(na, a, w_in, b_in, w_out, b_out, y, h, hpre, M, d, mlp, dtype, ws, ws_bytes, stream))
  if (!na || !a || !w_in || !b_in || !w_out || !b_out || !y || !h || !hpre || dtype == SPA3D_F32) return SPA3D_ERR_ARG;
  if (d != 384 || mlp != 1536 || M < 1) return SPA3D_ERR_ARG;   // shape first: a wrong shape is an argument error whatever the workspace
  for (const void* p : {na, a, (const void*)y, (const void*)h, (const void*)hpre, w_in, w_out})
    if (((uintptr_t)p) & 15) return SPA3D_ERR_ARG;               // 16-byte vector / LDS-DMA accesses on every operand
  OpCtx c(stream, ws, ws_bytes);
  bf16_t* wpk = c.alloc<bf16_t>(mlp_fused_pack_elems());
  if (c.ar.overflow) return SPA3D_ERR_WORKSPACE;
  mlp_fused_pack<bf16_t>(&c, (const bf16_t*)w_in, (const bf16_t*)w_out, wpk);
  if (!mlp_fused_fwd(&c, (const bf16_t*)na, (const bf16_t*)a, (bf16_t*)y, (bf16_t*)h, (bf16_t*)hpre, M, d, mlp, wpk, b_in, b_out)) return SPA3D_ERR_ARG;
  return c.status();
}

int spa3d_op_qkv_attention(const void* nq, int64_t ldn, const void* wq, const void* wk, const void* wv, const float* scale_q, const float* scale_k,
                           const float* keymask, const int32_t* seq_off, int64_t nseq, int32_t S, int32_t H, void* qkv, void* o, float* lse,
                           int32_t dtype, void* ws, int64_t ws_bytes, void* stream) {
  FWD16(spa3d_op_qkv_attention_f16(nq, ldn, wq, wk, wv, scale_q, scale_k, keymask, seq_off, nseq, S, H, qkv, o, lse, dtype, ws, ws_bytes, stream))
  if (!nq || !wq || !wk || !wv || !scale_q || !scale_k || !qkv || !o || dtype == SPA3D_F32 || H < 1 || H > 64) return SPA3D_ERR_ARG;
  OpCtx c(stream, ws, ws_bytes);
  bf16_t* wpk = c.alloc<bf16_t>(qkv_attn_pack_elems(H));
  if (c.ar.overflow) return SPA3D_ERR_WORKSPACE;
  qkv_attn_pack<bf16_t>(&c, (const bf16_t*)wq, (const bf16_t*)wk, (const bf16_t*)wv, H * 96, H, wpk);
  if (!qkv_attn_fwd(&c, (const bf16_t*)nq, ldn, wpk, scale_q, scale_k, keymask, nseq, S, H, 96, 384, (bf16_t*)qkv, (bf16_t*)o, lse, seq_off, 0)) return SPA3D_ERR_ARG;
  return c.status();
}

int spa3d_op_layernorm(const void* x, const float* scale, void* y, float* stats, int64_t rows, int32_t d, int32_t dtype, void* stream) {
  FWD16(spa3d_op_layernorm_f16(x, scale, y, stats, rows, d, dtype, stream))
  if (!x || !scale || !y || d <= 0 || d > 2048) return SPA3D_ERR_ARG;
  OpCtx c(stream, nullptr, 0);
  if (dtype == SPA3D_F32) k_layernorm<float>(&c, (const float*)x, scale, (float*)y, stats, rows, d);
  else k_layernorm<bf16_t>(&c, (const bf16_t*)x, scale, (bf16_t*)y, stats, rows, d);
  return c.status();
}
int spa3d_op_layernorm_bwd(const void* x, const float* scale, const float* stats, const void* dy, void* dx, float* dscale, int64_t rows,
                           int32_t d, int32_t dtype, void* stream) {
  FWD16(spa3d_op_layernorm_bwd_f16(x, scale, stats, dy, dx, dscale, rows, d, dtype, stream))
  if (!x || !scale || !stats || !dy || !dx || !dscale || d <= 0 || d > 2048) return SPA3D_ERR_ARG;
  OpCtx c(stream, nullptr, 0);
  if (dtype == SPA3D_F32) k_layernorm_bwd<float>(&c, (const float*)x, scale, stats, (const float*)dy, (float*)dx, dscale, rows, d, nullptr);
  else k_layernorm_bwd<bf16_t>(&c, (const bf16_t*)x, scale, stats, (const bf16_t*)dy, (bf16_t*)dx, dscale, rows, d, nullptr);
  return c.status();
}

int spa3d_op_attention(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldk, int64_t ldv, const float* scale_q,
                       const float* scale_k, const float* keymask, int64_t nseq, int32_t Sq, int32_t Sk, int32_t H, int32_t Dh, void* o,
                       float* lse, int32_t dtype, int32_t impl, void* ws, int64_t ws_bytes, void* stream) {
  FWD16(spa3d_op_attention_f16(q, k, v, ldq, ldk, ldv, scale_q, scale_k, keymask, nseq, Sq, Sk, H, Dh, o, lse, dtype, impl, ws, ws_bytes, stream))
  if (!q || !k || !v || !o || !scale_q || !scale_k || Dh > 128) return SPA3D_ERR_ARG;
  OpCtx c(stream, ws, ws_bytes);
  apply_attn_impl(&c, impl);
  if (dtype == SPA3D_F32)
    attention_fwd<float>(&c, (const float*)q, (const float*)k, (const float*)v, ldq, ldk, ldv, scale_q, scale_k, keymask, nseq, Sq, Sk, H, Dh,
                         (float*)o, lse, c.attn_impl);
  else
    attention_fwd<bf16_t>(&c, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, ldk, ldv, scale_q, scale_k, keymask, nseq, Sq, Sk, H,
                          Dh, (bf16_t*)o, lse, c.attn_impl);
  return c.status();
}
int spa3d_op_attention_bwd(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldk, int64_t ldv, const float* scale_q,
                           const float* scale_k, const float* keymask, int64_t nseq, int32_t Sq, int32_t Sk, int32_t H, int32_t Dh,
                           const void* o, const float* lse, const void* d_o, void* dq, void* dk, void* dv, float* dscale_q,
                           float* dscale_k, int32_t dtype, int32_t impl,
                           void* ws, int64_t ws_bytes, void* stream) {
  FWD16(spa3d_op_attention_bwd_f16(q, k, v, ldq, ldk, ldv, scale_q, scale_k, keymask, nseq, Sq, Sk, H, Dh, o, lse, d_o, dq, dk, dv, dscale_q,
                                   dscale_k, dtype, impl, ws, ws_bytes, stream))
  if (!q || !k || !v || !d_o || !dq || !dk || !dv || !dscale_q || !dscale_k || Dh > 128) return SPA3D_ERR_ARG;
  OpCtx c(stream, ws, ws_bytes);
  apply_attn_impl(&c, impl);
  if (dtype == SPA3D_F32)
    attention_bwd<float>(&c, (const float*)q, (const float*)k, (const float*)v, ldq, ldk, ldv, scale_q, scale_k, keymask, nseq, Sq, Sk, H, Dh,
                         (const float*)o, lse, (const float*)d_o, (float*)dq, (float*)dk, (float*)dv, dscale_q, dscale_k, c.attn_impl);
  else
    attention_bwd<bf16_t>(&c, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, ldk, ldv, scale_q, scale_k, keymask, nseq, Sq, Sk, H,
                          Dh, (const bf16_t*)o, lse, (const bf16_t*)d_o, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, dscale_q, dscale_k, c.attn_impl);
  return c.status();
}

}  // extern "C"
