// common.hpp -- shared declarations for libspa3d_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/spa3d.h"
#include "ablate.inc"  // every work-skipping diagnostic mask, all 0 in libspa3d_hip.so

// One translation unit is compiled for exactly ONE 16-bit activation type: bf16 (default) or IEEE fp16 (-DSPA_F16=1, BASELINE
// cfg#5).  The raw 16-bit storage type is `bf16_t` (unsigned short) in both builds; what differs -- the two conversions, the MFMA
// instruction, the packed dot product -- is defined here, and everything device-side lives in a per-type namespace so the two builds
// of every source link into one library.  (No dual "platform" paths: both are gfx950-only code.)
#ifndef SPA_F16
#define SPA_F16 0
#endif
#if SPA_F16
#define SPA_NS h_f16
#else
#define SPA_NS h_bf16
#endif
typedef unsigned short bf16_t;  // raw 16-bit pattern (bf16, or fp16 in the SPA_F16 build); arithmetic is always done in f32

namespace SPA_NS {
#if SPA_F16
__device__ __forceinline__ float bf2f(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
// round-to-nearest-even; overflows to inf beyond 65504 and flushes below 6e-8 (the fp16 mode scales the loss, model.hip)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
typedef __attribute__((ext_vector_type(8))) _Float16 mfma16x8;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define MFMA32_ASM "v_mfma_f32_32x32x16_f16"
#define DOT2C_F32_16 "v_dot2c_f32_f16"
#define ONES2_16 0x3C003C00u  // (1.0, 1.0)
#else
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// round-to-nearest-even; a plain cast emits v_cvt_pk_bf16_f32 and keeps NaN a NaN
// (MI355X_MICROARCH.md "Correctness boundaries")
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
// the 16-bit activation type's MFMA operand vector and instructions (8 elements per lane)
typedef __attribute__((ext_vector_type(8))) __bf16 mfma16x8;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define MFMA32_ASM "v_mfma_f32_32x32x16_bf16"
#define DOT2C_F32_16 "v_dot2c_f32_bf16"
#define ONES2_16 0x3F803F80u  // (1.0, 1.0)
#endif
// two f32 -> one dword of two 16-bit values (low half = a): ONE v_cvt_pk_bf16_f32 (fp16 build: two converts + a pack).  The element-wise form
// (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16) compiles to a convert per element plus a shift / mask / or: three VALU operations per pair in every GEMM epilogue.
#if SPA_F16
typedef __attribute__((ext_vector_type(2))) _Float16 spa_h16x2;
#else
typedef __attribute__((ext_vector_type(2))) __bf16 spa_h16x2;
#endif
typedef __attribute__((ext_vector_type(2))) float spa_f32x2;
__device__ __forceinline__ unsigned f2bf_pack2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(spa_f32x2{a, b}, spa_h16x2)); }
// the two 16-bit halves of a packed dword as f32
__device__ __forceinline__ float unpack_lo(unsigned u) { return bf2f((bf16_t)(u & 0xffffu)); }
__device__ __forceinline__ float unpack_hi(unsigned u) { return bf2f((bf16_t)(u >> 16)); }
template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float c = 0.7978845608028654f;  // sqrt(2/pi)
  float u = c * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + tanhf(u));
}
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
  const float c = 0.7978845608028654f;
  float u = c * (x + 0.044715f * x * x * x);
  float t = tanhf(u);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * c * (1.0f + 3.0f * 0.044715f * x * x);
}

// bf16-path variants (tiled GEMM epilogues only; the fp32 parity path keeps tanhf): gelu(x) = x*s, s = sigmoid(2u) = 1/(1+exp(-2u)),
// gelu'(x) = s*(1 + 2x(1-s)u'), one v_exp_f32 + one v_rcp_f32 (abs error ~1e-7, far below bf16 rounding).
__device__ __forceinline__ float gelu_sigmoid_2u(float x) {
  // exp(-2u) = 2^(x (A + B x^2)) with A = -2 sqrt(2/pi) log2(e), B = 0.044715 A: one fma and two multiplies feed v_exp_f32 directly
  constexpr float A = (float)(-2.0 * 0.7978845608028654 * 1.4426950408889634), B = (float)(-2.0 * 0.7978845608028654 * 1.4426950408889634 * 0.044715);
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * __builtin_fmaf(B, x * x, A)));
}
__device__ __forceinline__ float gelu_tanh_fast_f(float x) { return x * gelu_sigmoid_2u(x); }
__device__ __forceinline__ float gelu_tanh_grad_fast_f(float x) {
  const float s = gelu_sigmoid_2u(x);
  const float du = 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
  return s * (1.0f + 2.0f * x * (1.0f - s) * du);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- deterministic parameter gradients (spa3d_set_option "det_grads").  Every reduction INTO the flat gradient buffer (split-M dW tiles, bias / scale column sums,
// broadcast gradients) is a float atomic by default: fast, and its result depends on arrival order in the last bits.  In this mode the same call sites add 64-bit FIXED-POINT
// integers (2^-32 units: exact, order-independent) into a shadow of the gradient buffer, which det_flush adds to the float buffer once a range is final.  The switch is a
// per-translation-unit device variable (no relocatable device code here), uploaded by det_upload_* at the start of a call; nullptr = float atomics.
struct DetCfg { float* gbase; long long* shadow; long long n; unsigned* flag; };
static __device__ DetCfg det_cfg_dev;
// A kernel with many additions loads the switch ONCE (det_load: scalar loads into SGPRs) and passes it along; re-reading the device variable at every addition cost the
// large-tile dW kernel's 384-atomic epilogue and the single-query attention backward's inner loop 60 ms/step with the mode OFF (round 5, first version).
__device__ __forceinline__ DetCfg det_load() { return det_cfg_dev; }
__device__ __forceinline__ bool det_on() { return det_cfg_dev.shadow != nullptr; }
__device__ __forceinline__ void grad_add(const DetCfg& dc, float* p, float v) {
  if (dc.shadow) {
    const long long i = p - dc.gbase;
    if ((unsigned long long)i < (unsigned long long)dc.n) {
      const float f = v * 4294967296.f;
      if (fabsf(f) < 9.0e18f) atomicAdd((unsigned long long*)(dc.shadow + i), (unsigned long long)__float2ll_rn(f));   // NaN fails the comparison too
      else atomicOr(dc.flag, 1u);                                                                                    // sticky: det_flush then writes NaN
      return;
    }
  }
  atomicAdd(p, v);
}
__device__ __forceinline__ void grad_add(float* p, float v) { grad_add(det_load(), p, v); }   // cold sites: one addition per thread at a workgroup's end
// (a one-thread kernel, not hipMemcpyToSymbolAsync: a copy from pageable host memory may block the host until the stream has drained)
#define SPA_DET_UPLOAD_DEF(fn)                                                         \
  __global__ void fn##_kernel(DetCfg d) { det_cfg_dev = d; }                           \
  void fn(hipStream_t st_, const DetCfg* d) { fn##_kernel<<<1, 1, 0, st_>>>(*d); }
void det_upload_kernels(hipStream_t, const DetCfg*);
void det_upload_gemm_fast(hipStream_t, const DetCfg*);
void det_upload_gemm_tnb(hipStream_t, const DetCfg*);
void det_upload_gemm_generic(hipStream_t, const DetCfg*);
void det_upload_attn(hipStream_t, const DetCfg*);

}  // namespace SPA_NS

// ------------------------------------------------------------------------------------------
// host-side context
// ------------------------------------------------------------------------------------------
struct Leaf {
  std::string name;
  int ndim;
  int64_t shape[4];
  int64_t offset;
  int64_t numel() const { int64_t n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i]; return n; }
};

struct Arena {  // bump allocator over the caller's workspace; dry mode only counts
  char* base = nullptr;
  int64_t cap = 0, off = 0, peak = 0;
  bool dry = false, overflow = false;
  void* alloc(int64_t bytes) {
    int64_t a = (off + 255) & ~int64_t(255);
    off = a + bytes;
    if (off > peak) peak = off;
    if (dry) return (void*)(uintptr_t)(0x1000 + a);  // never dereferenced
    if (off > cap) { overflow = true; return base; }
    return base + a;
  }
  int64_t mark() const { return off; }
  void release(int64_t m) { off = m; }
};

// live per-kernel-class timing with HIP events on the launch stream (bench.py "roofline"; off by default)
enum { PROF_GEMM_NT = 0, PROF_GEMM_TN = 1, PROF_GEMM_GENERIC = 2, PROF_ATTN_FWD = 3, PROF_ATTN_BWD = 4, PROF_LN_FWD = 5, PROF_LN_BWD = 6,
       PROF_ATTN_Q1 = 7, PROF_EMBED = 8, PROF_NCLS = 9 };  // PROF_EMBED brackets the whole input-embedding stage (its GEMMs are also in PROF_GEMM_NT)
struct ProfRec { int cls; double flops, bytes; hipEvent_t e0, e1; int64_t tag[4] = {0, 0, 0, 0}; };  // tag: M, N, K, flags (GEMMs)
struct Prof {
  bool on = false;
  std::vector<hipEvent_t> pool; size_t used = 0;
  std::vector<ProfRec> recs;
  hipEvent_t get() {
    if (used == pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; pool.push_back(e); }
    return pool[used++];
  }
};

struct spa3d_ctx {
  spa3d_config cfg;
  std::vector<Leaf> leaves;
  int64_t nparams = 0;
  std::string err;
  hipStream_t stream = nullptr;
  Arena ar;
  bool dry = false;   // orchestration runs without launching (workspace sizing)
  int hip_err = 0;
  int gemm_impl = 0;  // 0 auto, 1 generic only, 2 tiled (see apply_gemm_impl)
  int attn_impl = 0;  // 0 auto, 1 generic, 2 fused (see apply_attn_impl)
  int chunk = 0;      // samples per chunk: 0 = as many as fit the workspace (spa3d_set_option "chunk")
  int ro_share = 1;       // readout block 1: LayerNorm / QKV once per distinct (sample, query frame) instead of per query (16-bit modes); SPA3D_RO_SHARE=0 disables
  int prune = 1;          // drop masked frame tokens from the track encoder (3DSPA model, fused 16-bit attention path); SPA3D_PRUNE=0 disables
  float loss_scale = 1.f;  // the 16-bit backward runs at loss x scale, parameter gradients are scaled back at the end: 1 = off (bf16 / fp32),
                           // > 0 a fixed scale, < 0 automatic with |loss_scale| the target head-gradient magnitude (fp16 mode: -16)
  void* grad_ev[2] = {nullptr, nullptr};  // spa3d_set_grad_events: recorded on the launch stream when a gradient segment is final (last chunk)
  int64_t grad_ev_gen[2] = {0, 0};  // how many times each event has been recorded (spa3d_grad_events_recorded)
  bool last_chunk = false;
  const float* loss_scale_state = nullptr;  // caller-owned device float (spa3d_set_loss_scale_state): dynamic multiplier of the loss scale
  double plan_stats[4] = {0, 0, 0, 0};  // last train call: encoder rows kept, encoder rows dense, readout slots, queries (spa3d_plan_stats)
  // kernel-choice knobs behind gemm_impl / attn_impl (spa3d_set_option; values 3+ are test hooks that put small problems on the big kernels)
  int attn_bwd_mode = 0;  // fused attention backward structure: 0 auto (1 up to S = 160, else 3), 1 four images + concurrent roles, 2 split-pass 4 waves, 3 split-pass 8 waves
  int nt_8p = 1;          // 8-phase NT kernels (256x256 / 128x384): 1 for M >= 16 384, 2 for any M (tests), 0 off
  int nt_8pp = 5;         // their persistent forms: 5 both tiles (default), 1 the 256x256 one only (tests: the non-persistent 128x384 kernel), 0 off
  int tn_8p = 1;          // 8-phase TN (dW) kernels: 1 by size, 2 forced (tests), 0 off
  int tn_big = 1;         // large-register-tile TN (dW) kernel (gemm_tnb.hip): 1 by size, 2 forced (tests), 0 off (gemm_impl 6 / 8)
  int nt_occ = 1;         // single-buffer 4-workgroups/CU NT kernel for K <= 512; 0 (tests) = the double-buffered kernel
  int nt_stream = 1;      // non-temporal stores for 16-bit outputs >= 512 MB
  int embed_fused = 1;    // input embedding as ONE GEMM over the concatenated K written once, in compact row order (model.hip encode_chunk); gemm_impl 6 = the multi-pass path
  int mlp_fused = 1;      // track-encoder MLP forward as ONE sequence-resident kernel (mlp_fused.hip); gemm_impl 6 = the two tiled GEMMs
  int nt_big = 1;         // large-register-tile NT kernel (gemm_ntb.hip) for the N = 384 dX GEMMs of the track encoder; 2 forced for any M (tests, gemm_impl 9), 0 off (gemm_impl 6 / 8)
  int rs_gemm = 1;        // K = 384 projections on the row-stationary kernel (gemm_rs.hip); gemm_impl 6 = the tiled kernels; 7 (ops) = required
  int qkv_attn = 0;       // track-encoder QKV projection + attention forward as ONE kernel (qkv_attn.hip): built and measured in round 5, 1.47x SLOWER than the
                          // projection GEMM + attention kernel pair (profiles/r05_qkv_attn_fused.log), so opt-in only: attn_impl 6
  int det_grads = 0;      // spa3d_set_option "det_grads": order-independent parameter gradients (fixed-point shadow accumulation, DetCfg above); costs a few %
  int det_uploaded = 0;   // the device-side switch currently holds a live shadow (must be cleared by the next call that runs without it)
  SPA_NS::DetCfg det_host = {nullptr, nullptr, 0, nullptr};  // what was uploaded last (kept alive for the asynchronous copy)
  int poison = 0;         // spa3d_set_option "poison": NaN-fill the workspace before every chunk and every op output before its launch (tests)
  bool tn_colsum_fused = false;  // set by gemm_tn_bf16: the last call also produced GemmDesc::colsum_out
  Prof prof;
};

// gemm_impl: 0 product dispatch | 1 generic kernels only | 2 tiled kernels, product tile choice (ops: error when unusable) -- and test hooks that put
// SMALL problems on the big kernels: 3 every eligible GEMM on the 8-phase kernels (persistent forms included; dW on the 8-wave kernels), 4 the same with the non-persistent
// 128x384 kernel, 5 tiled without the single-buffer short-K kernel, 6 tiled GEMMs without the round-4 kernels (MLP forward as two GEMMs, multi-pass input embedding, no row-stationary
// K = 384 kernel, no round-5 large-tile dW kernel), 7 (ops only) the row-stationary kernel or an error, 8 the product dispatch without the round-5 large-tile dW kernel, 9 = 3 with every divisible dW on the large-tile kernel
inline void apply_gemm_impl(spa3d_ctx* c, int v) {
  c->gemm_impl = v == 1 ? 1 : (v >= 2 ? 2 : 0);
  c->nt_8p = 1; c->nt_8pp = 5; c->tn_8p = 1; c->tn_big = 1; c->nt_big = 1; c->nt_occ = 1; c->mlp_fused = 1; c->embed_fused = 1; c->rs_gemm = 1;
  if (v == 3 || v == 4) { c->nt_8p = 2; c->tn_8p = 2; c->tn_big = 0; c->nt_big = 0; }
  if (v == 9) { c->nt_8p = 2; c->tn_8p = 2; c->tn_big = 2; c->nt_big = 2; }
  if (v == 4) c->nt_8pp = 1;
  if (v == 5) c->nt_occ = 0;
  if (v == 6) { c->mlp_fused = 0; c->embed_fused = 0; c->rs_gemm = 0; c->tn_big = 0; c->nt_big = 0; }
  if (v == 8) { c->tn_big = 0; c->nt_big = 0; }
}
// attn_impl: 0 product dispatch | 1 generic composition (GEMMs + softmax kernels) | 2 fused kernels (ops: error when unusable) | 3 / 4 fused with the
// split-pass backward on 4 / 8 waves also where the four-image kernel would run (S <= 160; tests) | 6 fused kernels with the track encoder's QKV projection + attention forward as ONE launch (qkv_attn.hip)
inline void apply_attn_impl(spa3d_ctx* c, int v) {
  c->attn_impl = v == 1 ? 1 : (v >= 2 ? 2 : 0);
  c->attn_bwd_mode = v == 3 ? 2 : (v == 4 ? 3 : 0);
  c->qkv_attn = v == 6;   // 6 = the fused kernels with the round-5 QKV projection + attention forward (opt-in: measured slower)
}

struct ProfScope {  // records an event pair around the launches issued in its lifetime
  spa3d_ctx* c; ProfRec r; bool on;
  ProfScope(spa3d_ctx* c_, int cls, double flops, double bytes) : c(c_), on(c_->prof.on && !c_->dry) {
    if (!on) return;
    r.cls = cls; r.flops = flops; r.bytes = bytes; r.e0 = c->prof.get(); r.e1 = c->prof.get();
    if (!r.e0 || !r.e1) { on = false; return; }
    (void)hipEventRecord(r.e0, c->stream);
  }
  void tag(int64_t a, int64_t b, int64_t c2, int64_t d) { r.tag[0] = a; r.tag[1] = b; r.tag[2] = c2; r.tag[3] = d; }
  ~ProfScope() { if (on) { (void)hipEventRecord(r.e1, c->stream); c->prof.recs.push_back(r); } }
};

#define SPA_LAUNCH_CHECK(ctx)                                                     \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess && (ctx)->hip_err == 0) {                               \
      (ctx)->hip_err = (int)e__;                                                  \
      (ctx)->err = std::string("HIP launch failed at ") + __FILE__ + ":" +        \
                   std::to_string(__LINE__) + ": " + hipGetErrorString(e__);      \
    }                                                                             \
  } while (0)

// ------------------------------------------------------------------------------------------
// GEMM descriptor (generic, batched, strided).  C[b][m][n] (op)= alpha*sum_k A[b][m][k]*B[b][k][n]
// ------------------------------------------------------------------------------------------
enum { EPI_NONE = 0, EPI_GELU = 1, EPI_MUL_GELU_GRAD = 2 };
struct GemmDesc {
  const void* A; const void* B; void* C;
  int64_t M; int32_t N; int32_t K;
  int64_t sAm, sAk, sBk, sBn, sCm;   // element strides (C is n-contiguous)
  int32_t nb1 = 1, nb2 = 1;          // two-level batch: blockIdx.z = b1*nb2 + b2
  int64_t bA1 = 0, bA2 = 0, bB1 = 0, bB2 = 0, bC1 = 0, bC2 = 0;
  float alpha = 1.f;
  const float* bias = nullptr;       // [N] f32, added before act
  int epi = EPI_NONE;
  const void* aux = nullptr;         // residual (added after act) or pre-activation (EPI_MUL_GELU_GRAD); C layout, T
  int aux_is_residual = 1;
  int out_f32 = 0;                   // C is float regardless of T
  int accumulate = 0;                // C += (non-atomic)
  int atomic = 0;                    // C += via atomicAdd (f32 C only)
  float* colsum_out = nullptr;       // TN (dW) only: also accumulate the column sums of B (bias gradient) when the kernel can;
                                     // the callee reports it in spa3d_ctx::tn_colsum_fused
  const void* Bt = nullptr;          // optional copy of B stored [N][K] (K contiguous, row stride ldBt) for the tiled kernels
  int64_t ldBt = 0;
  int32_t crow_group = 0, crow_skip = 0;  // C row m is stored at row m + (m/crow_group + 1)*crow_skip (token rows behind a readout row)
  int32_t brow_group = 0, brow_skip = 0;  // same remap on B's k index (dW over token rows that skip the readout row)
  void* pre_out = nullptr;                // with EPI_GELU: the pre-activation (T, C layout) is stored here as well
  const void* zero_page = nullptr;        // >= 16 B of zeros (tiled TN kernel: rows past the end of the reduction)
  // one-pass input embedding (tiled NT, N == 384, 16-bit): K columns [0, K1) from A, [K1, K) from A2 (row stride sA2m); input rows gathered through
  // arow_idx (both sources; also indexes r1_x), output rows scattered through crow_idx (< 0 = dropped); epilogue += r1_x[input row] * r1_w[n] in f32.
  // gemm_nt_bf16 returns false when it cannot honour them (the caller then takes the multi-pass path)
  const void* A2 = nullptr; int64_t sA2m = 0; int32_t K1 = 0; const int32_t* arow_idx = nullptr; const int32_t* crow_idx = nullptr;
  const void* r1_x = nullptr; const float* r1_w = nullptr;
  int32_t seg_n = 0; void* C_seg[2] = {nullptr, nullptr};  // tiled TN (dW) only: output columns in seg_n-wide segments, segment s >= 1 in C_seg[s-1]
                                                          // (the q / k / v kernels of a fused projection are separate leaves); gemm_tn_bf16 returns
                                                          // false when it cannot honour it
};

namespace SPA_NS {
template <typename T> void gemm_generic(spa3d_ctx* c, const GemmDesc& d);
// tiled bf16 kernels (gemm_fast.hip).  Return false if the shape/layout is not supported.
bool gemm_nt_bf16(spa3d_ctx* c, const GemmDesc& d);
bool gemm_tn_bf16(spa3d_ctx* c, const GemmDesc& d);
// large-register-tile NT GEMM (gemm_ntb.hip): C[M,N] (16-bit) = A[M,K] . W (+ bias), N a multiple of 384 (tile 256 x 384) or 256 (384 x 256), K % 32 == 0;
// W pre-packed by gemm_ntb_pack (element (k, n) of W at w[k*sk + n*sn])
bool gemm_ntb_ok(int K, int N);
int64_t gemm_ntb_pack_elems(int K, int N);
template <typename S> void gemm_ntb_pack(spa3d_ctx* c, const S* w, int64_t sk, int64_t sn, int K, int N, bf16_t* wpk);
bool gemm_ntb(spa3d_ctx* c, const bf16_t* A, int64_t lda, const bf16_t* wpk, const float* bias, bf16_t* C, int64_t ldc, int64_t M, int N, int K);
// row-stationary K = 384 GEMM (gemm_rs.hip): C[M,N] = A[M,384] . W (+ bias) with W pre-packed into the kernel's fragment stream (element (k, n) of W at w[k*sk + n*sn])
bool gemm_rs_ok(int K, int N);
int64_t gemm_rs_pack_elems(int N);
template <typename S> void gemm_rs_pack(spa3d_ctx* c, const S* w, int64_t sk, int64_t sn, int N, bf16_t* wpk);
// gelu_pre != null: C = (A . W + bias) o gelu'(gelu_pre) (gelu_pre in C's layout, row stride ldpre): the MLP backward's dh
bool gemm_rs(spa3d_ctx* c, const bf16_t* A, int64_t lda, const bf16_t* wpk, const float* bias, bf16_t* C, int64_t ldc, int64_t M, int N,
             const bf16_t* gelu_pre = nullptr, int64_t ldpre = 0);
// QKV projection + attention forward of one (sequence, head) per workgroup pass (qkv_attn.hip): d = 384, Dh = 96, S <= 160; false = shape not covered
int64_t qkv_attn_pack_elems(int H);
template <typename S> void qkv_attn_pack(spa3d_ctx* c, const S* wq, const S* wk, const S* wv /* [384][E] */, int E, int H, bf16_t* wpk);
bool qkv_attn_fwd(spa3d_ctx* c, const bf16_t* nq, int64_t ldn, const bf16_t* wpk, const float* sq, const float* sk, const float* km, int64_t nseq, int S, int H,
                  int Dh, int d, bf16_t* qkv, bf16_t* o, float* lse, const int32_t* seq_off, int64_t total_rows);
// sequence-resident MLP forward for d = 384, mlp = 1536 (mlp_fused.hip): y = a + MLP(na), h / hpre kept; false = shape not covered
template <typename S> void mlp_fused_pack(spa3d_ctx* c, const S* w_in /*[384][1536]*/, const S* w_out /*[1536][384]*/, bf16_t* wpk);
int64_t mlp_fused_pack_elems();
bool mlp_fused_fwd(spa3d_ctx* c, const bf16_t* na, const bf16_t* a, bf16_t* y, bf16_t* h, bf16_t* hpre, int64_t M, int d, int mlp,
                   const bf16_t* wpk, const float* b_in, const float* b_out);

// ------------------------------------------------------------------------------------------
// elementwise / reduction kernels (kernels.hip), all asynchronous on c->stream; no-ops when c->dry
// ------------------------------------------------------------------------------------------
template <typename T> void k_layernorm(spa3d_ctx*, const T* x, const float* scale, T* y, float* stats, int64_t rows, int d);
template <typename T> void k_layernorm_bwd(spa3d_ctx*, const T* x, const float* scale, const float* stats, const T* dy, T* dx,
                                           float* dscale, int64_t rows, int d, const T* add);
template <typename T> void k_rmsnorm_heads(spa3d_ctx*, const T* x, int64_t ldx, const float* scale, T* y, int64_t ldy, int64_t rows, int H, int Dh);
template <typename T> void k_rmsnorm_heads_bwd(spa3d_ctx*, const T* x, int64_t ldx, const float* scale, const T* dy, int64_t lddy, T* dx,
                                               int64_t lddx, float* dscale, int64_t rows, int H, int Dh);
template <typename T> void k_softmax(spa3d_ctx*, T* s, const float* keymask, int64_t nseq, int H, int Sq, int Sk);
template <typename T> void k_softmax_bwd(spa3d_ctx*, const T* p, T* dp, int64_t rows, int Sk, const float* keymask = nullptr,
                                         int64_t rows_per_seq = 1);
template <typename T> void k_sin_embed(spa3d_ctx*, const float* x, int64_t rows, int C, int nf, float prescale, T* out);
template <typename T> void k_embed_tokens(spa3d_ctx*, const float* tracks, int64_t nrows, int T_, int nf, float prescale, T* sinbuf, int NC = 3);
template <typename T> void k_colsum(spa3d_ctx*, const T* x, int64_t rows, int n, int64_t ld, float* out /*accumulated*/, int rgroup = 0,
                                    int rskip = 0);
template <typename T> void k_gelu(spa3d_ctx*, const T* x, T* y, int64_t n);
template <typename T> void k_add(spa3d_ctx*, T* dst, const T* src, int64_t n);
void k_fill(spa3d_ctx*, float* p, float v, int64_t n);
void k_zero(spa3d_ctx*, void* p, int64_t bytes);
void k_mul(spa3d_ctx*, float* a, const float* b, int64_t n);
void k_set_loss_scale(spa3d_ctx*, const float* denom_dev, float l1w, float setting, float* scale_dev);
void k_unscale(spa3d_ctx*, float* a, const float* scale_dev, int64_t n);
template <typename T> void k_cast_from_f32(spa3d_ctx*, const float* src, T* dst, int64_t n);
template <typename T> void k_cast_to_f32(spa3d_ctx*, const T* src, float* dst, int64_t n);
template <typename T> void k_pack(spa3d_ctx*, const float* src, int64_t src_ld, int rows, int cols, T* dst_native, int64_t ldn, T* dst_T, int64_t ldt);
template <typename T> void k_transpose(spa3d_ctx*, const T* src, int rows, int cols, T* dst);  // dst[c][r] = src[r][c]
template <typename T> void k_set_readout_rows(spa3d_ctx*, T* tok, const float* readout, int64_t nseq, int S, int d);
template <typename T> void k_embed_maps(spa3d_ctx*, const int32_t* row_src, int64_t rows, int S, int T_, int32_t* arow, int32_t* crow, T* tok, const float* readout, int d);
void k_sum3(spa3d_ctx*, const float* a, const float* b, const float* c3, float* out, int n);
void k_keymask(spa3d_ctx*, const float* visible, const int32_t* boundary, int64_t nseq, int N, int T_, float* km);
template <typename T> void k_gather_rows(spa3d_ctx*, const T* src, int64_t src_stride_rows, T* dst, int64_t n, int d);
// token pruning of the track encoder (kernels.hip): plan (returns the kept-row count, one stream sync) and rows-by-index movers
int64_t k_prune_plan(spa3d_ctx*, const float* km, int64_t nseq, int S, int32_t* cnt, int32_t* seq_off, int32_t* row_src);
template <typename T> void k_rows_idx(spa3d_ctx*, int mode /*0 gather, 1 scatter, 2 scatter-add*/, const T* src, const int32_t* idx, T* dst, int64_t n, int d);
template <typename T> void k_scatter_rows(spa3d_ctx*, const T* src, T* dst, int64_t dst_stride_rows, int64_t n, int d);
template <typename T> void k_compact_tokens(spa3d_ctx*, const T* tok, T* dst, int64_t nseq, int S, int d);
template <typename T> void k_broadcast_rows(spa3d_ctx*, const float* src, int rows, int d, T* dst, int64_t B);
template <typename T> void k_bcast_grad(spa3d_ctx*, const T* dsrc, int64_t per, int64_t B, int64_t bstride, float* dparam);
// Dense with input width K <= 4 as streaming kernels (false: shape not covered, use the GEMM path)
template <typename T> bool k_rank_fwd(spa3d_ctx*, const T* x, const T* w /*[K][N]*/, const float* bias, T* out /*+=*/, int64_t M, int N, int K, int64_t ldo,
                                      int rgroup, int rskip);
template <typename T> bool k_rank_bwd(spa3d_ctx*, const T* x, const T* dy, int64_t M, int N, int K, int64_t ldy, int rgroup, int rskip, float* gw /*+=*/,
                                      float* gb /*+=, may be null*/);
void k_discretize(spa3d_ctx*, const float* lat, const float* noise, int discretize, float* out, float* clipmask, int64_t n);
void k_query_embed1(spa3d_ctx*, const float* qp, int64_t nq, int nf, float track_scale, float time_scale, float* feat, int32_t* qframe,
                    int NC = 3);
template <typename T> void k_assemble_readout(spa3d_ctx*, const T* qtok, const T* lat, const int32_t* qframe, int64_t B, int Q, int L, int Cl,
                                              int D, T* seq);
// shared latent rows of the readout stack's first block (kernels.hip "Shared latent rows"; model.hip Share)
int64_t k_share_plan(spa3d_ctx*, const int32_t* qframe, int64_t B, int Q, int32_t* slot, int32_t* slot_b, int32_t* slot_f, int32_t* slot_q0, int32_t* scratch);
template <typename T> void k_share_assemble(spa3d_ctx*, const T* qtok, const T* lat, const int32_t* slot_b, const int32_t* slot_f, int64_t nslot, int64_t BQ,
                                            int L, int Cl, int D, T* xU);
template <typename T> void k_share_expand(spa3d_ctx*, const T* srcU, const int32_t* slot, const int32_t* slot_q0, int64_t nslot, int64_t nseq, int S, int d,
                                          const T* add, T* dst);
template <typename T> void k_share_reduce(spa3d_ctx*, const T* src, const int32_t* slot, const int32_t* slot_b, int64_t nslot, int64_t nseq, int Q, int S, int d,
                                          T* dstU);
template <typename T> void k_assemble_readout_bwd(spa3d_ctx*, const T* dseq, const int32_t* qframe, int64_t B, int Q, int L, int Cl, int D,
                                                  T* dqtok, float* dlat);
void k_loss_fwd(spa3d_ctx*, const float* head, int64_t nq, int T_, const float* tgt, const float* tvis, float* tracks, float* vlog,
                float* clog, float* sums, unsigned* poison, int NC = 3);
void k_loss_from_preds(spa3d_ctx*, const float* tracks, const float* vlog, int64_t n, const float* tgt, const float* tvis, float* sums,
                       unsigned* poison, int NC = 3);
template <typename T> void k_loss_bwd(spa3d_ctx*, const float* head, int64_t nq, int T_, const float* tgt, const float* tvis,
                                      const float* denom_dev, float l1w, float bcew, T* dhead, int NC = 3, const float* scale_dev = nullptr);
void k_vis_count(spa3d_ctx*, const float* tvis, int64_t n, float* out, unsigned* poison);
void k_set_denom(spa3d_ctx*, const float* sums, const unsigned* poison, float denom_host, float* denom_dev);
void k_loss_finalize(spa3d_ctx*, const float* sums, const unsigned* poison, const float* denom_dev, float l1w, float bcew, float* loss3);
void k_adamw(spa3d_ctx*, float* p, const float* g, float* m, float* v, int64_t n, float lr, int64_t step, float clip, float b1, float b2,
             float eps, float wd, float* scratch);
void k_uniform_noise(spa3d_ctx*, float* out, int64_t n, uint32_t k0, uint32_t k1);
void k_det_flush(spa3d_ctx*, float* g, long long* shadow, const unsigned* flag, int64_t n);  // g[i] += shadow[i] * 2^-32; shadow[i] = 0; NaN when *flag
// single-query attention of the pruned last block (kernels.hip)
template <typename T> void k_attn_q1_fwd(spa3d_ctx*, const T* q0, int64_t ldq0, const T* k, const T* v, int64_t ldk, int64_t ldv,
                                         const float* sq, const float* sk, const float* km, int64_t nseq, int S, int H, int Dh, T* o0,
                                         float* p0, const int32_t* seq_off = nullptr);
template <typename T> void k_attn_q1_bwd(spa3d_ctx*, const T* q0, int64_t ldq0, const T* k, const T* v, int64_t ldk, int64_t ldv,
                                         const float* sq, const float* sk, const float* km, int64_t nseq, int S, int H, int Dh,
                                         const float* p0, const T* d_o0, T* dq0, T* dk, T* dv, float* dsq, float* dsk,
                                         const int32_t* seq_off = nullptr);
template <typename T> void k_add_rows_strided(spa3d_ctx*, T* dst, const T* src, int64_t dst_stride_rows, int64_t n, int d);
// 2-D TRAJAN twin (track_autoencoder.py:117-390)
template <typename T> void k_vis_mean_pool(spa3d_ctx*, const T* tok, const float* vis, int64_t nseq, int T_, int d, T* out);
template <typename T> void k_vis_mean_pool_bwd(spa3d_ctx*, const T* dout, const float* vis, int64_t nseq, int T_, int d, T* dtok);
void k_keymask2d(spa3d_ctx*, const float* visible, const int32_t* boundary, int64_t nseq, int N, int T_, float* km);
}  // namespace SPA_NS
using namespace SPA_NS;  // one 16-bit type per translation unit
