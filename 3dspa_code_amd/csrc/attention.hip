// attention.hip -- attention core of ImprovedMHDPAttention (attention.py:166-175): per-head RMSNorm of q,k,
// q/sqrt(Dh), key mask with finfo.min fill, softmax, PV -- forward and backward.
// impl 1 ("generic"): composition of the strided batched MFMA GEMM with row-wise kernels; any Sq/Sk/Dh<=128,
// both dtypes; the F32 parity path.  impl 2: fused LDS-resident kernels (attention_fused.hip) for bf16.
#include <algorithm>
#include <cmath>

#include "common.hpp"

namespace SPA_NS {

bool attn_fused_fwd_bf16(spa3d_ctx* c, const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ldq, int64_t ldk, int64_t ldv,
                         const float* sq, const float* sk, const float* km, int64_t nseq, int Sq, int Sk, int H, int Dh, bf16_t* o,
                         float* lse, const int32_t* seq_off, int64_t total_rows);
bool attn_fused_bwd_bf16(spa3d_ctx* c, const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ldq, int64_t ldk, int64_t ldv,
                         const float* sq, const float* sk, const float* km, int64_t nseq, int Sq, int Sk, int H, int Dh, const bf16_t* o,
                         const float* lse, const bf16_t* d_o, bf16_t* dq, bf16_t* dk, bf16_t* dv, float* dsq, float* dsk,
                         const int32_t* seq_off, int64_t total_rows);

template <typename T> static T* aalloc(spa3d_ctx* c, int64_t n) { return (T*)c->ar.alloc(n * (int64_t)sizeof(T)); }

static int64_t attn_chunk(int64_t nseq, int Sq, int Sk, int H, int E, int esz) {
  int64_t per = (int64_t)H * Sq * Sk + (int64_t)(Sq + Sk) * E;
  int64_t cs = std::max<int64_t>(1, (int64_t)(768ll << 20) / (per * esz));
  return std::min(cs, nseq);
}
template <typename T>
static void scores(spa3d_ctx* c, const T* qn, const T* kn, T* s, int64_t ns, int Sq, int Sk, int H, int Dh) {
  const int E = H * Dh;
  GemmDesc d{};
  d.A = qn; d.B = kn; d.C = s; d.M = Sq; d.N = Sk; d.K = Dh;
  d.sAm = E; d.sAk = 1; d.sBk = 1; d.sBn = E; d.sCm = Sk;
  d.nb1 = (int)ns; d.nb2 = H; d.bA1 = (int64_t)Sq * E; d.bA2 = Dh; d.bB1 = (int64_t)Sk * E; d.bB2 = Dh;
  d.bC1 = (int64_t)H * Sq * Sk; d.bC2 = (int64_t)Sq * Sk;
  d.alpha = 1.0f / sqrtf((float)Dh);
  gemm_generic<T>(c, d);
}

template <typename T>
void attention_fwd(spa3d_ctx* c, const T* q, const T* k, const T* v, int64_t ldq, int64_t ldk, int64_t ldv, const float* sq, const float* sk,
              const float* km, int64_t nseq, int Sq, int Sk, int H, int Dh, T* o, float* lse, int impl, const int32_t* seq_off,
              int64_t total_rows) {
  const int E = H * Dh;
  if constexpr (sizeof(T) == 2) {
    if (impl != 1 && attn_fused_fwd_bf16(c, q, k, v, ldq, ldk, ldv, sq, sk, km, nseq, Sq, Sk, H, Dh, o, lse, seq_off, total_rows)) return;
  }
  if (seq_off) { if (!c->hip_err) { c->hip_err = -2; c->err = "ragged sequences need the fused attention kernels"; } return; }
  if (impl == 2) { if (!c->hip_err) { c->hip_err = -2; c->err = "fused attention forward does not cover this shape/dtype"; } return; }
  const int64_t cs = attn_chunk(nseq, Sq, Sk, H, E, (int)sizeof(T));
  int64_t mk = c->ar.mark();
  T* qn = aalloc<T>(c, cs * Sq * E); T* kn = aalloc<T>(c, cs * Sk * E); T* s = aalloc<T>(c, cs * H * Sq * Sk);
  for (int64_t s0 = 0; s0 < nseq; s0 += cs) {
    const int64_t ns = std::min(cs, nseq - s0);
    k_rmsnorm_heads<T>(c, q + s0 * Sq * ldq, ldq, sq, qn, E, ns * Sq, H, Dh);
    k_rmsnorm_heads<T>(c, k + s0 * Sk * ldk, ldk, sk, kn, E, ns * Sk, H, Dh);
    scores<T>(c, qn, kn, s, ns, Sq, Sk, H, Dh);
    k_softmax<T>(c, s, km ? km + s0 * Sk : nullptr, ns, H, Sq, Sk);
    GemmDesc d{};  // O = P V
    d.A = s; d.B = v + s0 * Sk * ldv; d.C = o + s0 * Sq * E; d.M = Sq; d.N = Dh; d.K = Sk;
    d.sAm = Sk; d.sAk = 1; d.sBk = ldv; d.sBn = 1; d.sCm = E;
    d.nb1 = (int)ns; d.nb2 = H; d.bA1 = (int64_t)H * Sq * Sk; d.bA2 = (int64_t)Sq * Sk; d.bB1 = (int64_t)Sk * ldv; d.bB2 = Dh;
    d.bC1 = (int64_t)Sq * E; d.bC2 = Dh;
    gemm_generic<T>(c, d);
  }
  c->ar.release(mk);
}
template <typename T>
void attention_bwd(spa3d_ctx* c, const T* q, const T* k, const T* v, int64_t ldq, int64_t ldk, int64_t ldv, const float* sq, const float* sk,
              const float* km, int64_t nseq, int Sq, int Sk, int H, int Dh, const T* o, const float* lse, const T* d_o, T* dq, T* dk,
              T* dv, float* dsq, float* dsk, int impl, const int32_t* seq_off, int64_t total_rows) {
  const int E = H * Dh;
  if constexpr (sizeof(T) == 2) {
    if (impl != 1 && o && lse &&
        attn_fused_bwd_bf16(c, q, k, v, ldq, ldk, ldv, sq, sk, km, nseq, Sq, Sk, H, Dh, o, lse, d_o, dq, dk, dv, dsq, dsk, seq_off, total_rows))
      return;
  }
  if (seq_off) { if (!c->hip_err) { c->hip_err = -2; c->err = "ragged sequences need the fused attention kernels"; } return; }
  if (impl == 2) { if (!c->hip_err) { c->hip_err = -2; c->err = "fused attention backward does not cover this shape/dtype"; } return; }
  const int64_t cs = attn_chunk(nseq, Sq, Sk, H, E, (int)sizeof(T));
  int64_t mk = c->ar.mark();
  T* qn = aalloc<T>(c, cs * Sq * E); T* kn = aalloc<T>(c, cs * Sk * E); T* s = aalloc<T>(c, cs * H * Sq * Sk);
  T* dp = aalloc<T>(c, cs * H * Sq * Sk); T* dqn = aalloc<T>(c, cs * Sq * E); T* dkn = aalloc<T>(c, cs * Sk * E);
  const float alpha = 1.0f / sqrtf((float)Dh);
  for (int64_t s0 = 0; s0 < nseq; s0 += cs) {
    const int64_t ns = std::min(cs, nseq - s0);
    const T* q0 = q + s0 * Sq * ldq; const T* k0 = k + s0 * Sk * ldk; const T* v0 = v + s0 * Sk * ldv;
    const T* do0 = d_o + s0 * Sq * E;
    k_rmsnorm_heads<T>(c, q0, ldq, sq, qn, E, ns * Sq, H, Dh);
    k_rmsnorm_heads<T>(c, k0, ldk, sk, kn, E, ns * Sk, H, Dh);
    scores<T>(c, qn, kn, s, ns, Sq, Sk, H, Dh);
    k_softmax<T>(c, s, km ? km + s0 * Sk : nullptr, ns, H, Sq, Sk);
    const int64_t bS1 = (int64_t)H * Sq * Sk, bS2 = (int64_t)Sq * Sk;
    {  // dP = dO V^T
      GemmDesc d{};
      d.A = do0; d.B = v0; d.C = dp; d.M = Sq; d.N = Sk; d.K = Dh;
      d.sAm = E; d.sAk = 1; d.sBk = 1; d.sBn = ldv; d.sCm = Sk;
      d.nb1 = (int)ns; d.nb2 = H; d.bA1 = (int64_t)Sq * E; d.bA2 = Dh; d.bB1 = (int64_t)Sk * ldv; d.bB2 = Dh; d.bC1 = bS1; d.bC2 = bS2;
      gemm_generic<T>(c, d);
    }
    {  // dV = P^T dO
      GemmDesc d{};
      d.A = s; d.B = do0; d.C = dv + s0 * Sk * ldv; d.M = Sk; d.N = Dh; d.K = Sq;
      d.sAm = 1; d.sAk = Sk; d.sBk = E; d.sBn = 1; d.sCm = ldv;
      d.nb1 = (int)ns; d.nb2 = H; d.bA1 = bS1; d.bA2 = bS2; d.bB1 = (int64_t)Sq * E; d.bB2 = Dh; d.bC1 = (int64_t)Sk * ldv; d.bC2 = Dh;
      gemm_generic<T>(c, d);
    }
    k_softmax_bwd<T>(c, s, dp, ns * H * Sq, Sk, km ? km + s0 * Sk : nullptr, (int64_t)H * Sq);  // dp := dS
    {  // dQn = alpha dS Kn
      GemmDesc d{};
      d.A = dp; d.B = kn; d.C = dqn; d.M = Sq; d.N = Dh; d.K = Sk;
      d.sAm = Sk; d.sAk = 1; d.sBk = E; d.sBn = 1; d.sCm = E; d.alpha = alpha;
      d.nb1 = (int)ns; d.nb2 = H; d.bA1 = bS1; d.bA2 = bS2; d.bB1 = (int64_t)Sk * E; d.bB2 = Dh; d.bC1 = (int64_t)Sq * E; d.bC2 = Dh;
      gemm_generic<T>(c, d);
    }
    {  // dKn = alpha dS^T Qn
      GemmDesc d{};
      d.A = dp; d.B = qn; d.C = dkn; d.M = Sk; d.N = Dh; d.K = Sq;
      d.sAm = 1; d.sAk = Sk; d.sBk = E; d.sBn = 1; d.sCm = E; d.alpha = alpha;
      d.nb1 = (int)ns; d.nb2 = H; d.bA1 = bS1; d.bA2 = bS2; d.bB1 = (int64_t)Sq * E; d.bB2 = Dh; d.bC1 = (int64_t)Sk * E; d.bC2 = Dh;
      gemm_generic<T>(c, d);
    }
    k_rmsnorm_heads_bwd<T>(c, q0, ldq, sq, dqn, E, dq + s0 * Sq * ldq, ldq, dsq, ns * Sq, H, Dh);
    k_rmsnorm_heads_bwd<T>(c, k0, ldk, sk, dkn, E, dk + s0 * Sk * ldk, ldk, dsk, ns * Sk, H, Dh);
  }
  c->ar.release(mk);
}


#define INST_ATTN(T)                                                                                                                  \
  template void attention_fwd<T>(spa3d_ctx*, const T*, const T*, const T*, int64_t, int64_t, int64_t, const float*, const float*,     \
                                 const float*, int64_t, int, int, int, int, T*, float*, int, const int32_t*, int64_t);                       \
  template void attention_bwd<T>(spa3d_ctx*, const T*, const T*, const T*, int64_t, int64_t, int64_t, const float*, const float*,     \
                                 const float*, int64_t, int, int, int, int, const T*, const float*, const T*, T*, T*, T*, float*, float*, int, \
                                 const int32_t*, int64_t);
INST_ATTN(float)
INST_ATTN(bf16_t)
}  // namespace SPA_NS
