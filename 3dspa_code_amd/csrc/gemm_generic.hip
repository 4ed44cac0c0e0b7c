// gemm_generic.hip -- batched, arbitrarily strided MFMA GEMM for T in {float, bf16}.
//   C[b][m][n] (op)= epi(alpha * sum_k A[b][m][k] * B[b][k][n] + bias[n])
// float : v_mfma_f32_16x16x4_f32  (bit-exact k-ordered fmaf chain -> the 1e-4 parity path)
// bf16  : v_mfma_f32_16x16x32_bf16, fp32 accumulate
// 64x64 output tile per 256-thread workgroup (4 waves as 2x2, 32x32 per wave), operands staged
// through LDS as [m][k] / [n][k] with element-wise strided loads so every layout (NN/NT/TN,
// batched heads, strided sub-views) works.  This is the fallback for odd shapes and the whole
// F32 mode; the hot bf16 shapes use gemm_fast.hip.
#include "common.hpp"

namespace SPA_NS {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef mfma16x8 bf16x8;  // 8 activation elements: the MFMA A/B operand of one lane

template <typename T> struct GemmCfg;
template <> struct GemmCfg<float> { static constexpr int BK = 16, PAD = 4; };
template <> struct GemmCfg<bf16_t> { static constexpr int BK = 32, PAD = 8; };

struct GemmArgs {
  const void* A; const void* B; void* C;
  int64_t M; int N; int64_t K;
  int64_t sAm, sAk, sBk, sBn, sCm;
  int nb2; int64_t bA1, bA2, bB1, bB2, bC1, bC2;
  float alpha; const float* bias; int epi; const void* aux; int aux_is_residual; int out_f32; int accumulate; int atomic;
  int tiles_m, tiles_n, ksplit; int64_t kchunk;
  int crow_group, crow_skip, brow_group, brow_skip;
  void* pre_out;
};

template <typename T>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmArgs g) {
  constexpr int BK = GemmCfg<T>::BK, LDS_LD = BK + GemmCfg<T>::PAD;
  __shared__ __attribute__((aligned(16))) T As[64 * LDS_LD];
  __shared__ __attribute__((aligned(16))) T Bs[64 * LDS_LD];
  // decode block id: ntile fastest, then mtile, then ksplit, then batch
  int64_t bid = blockIdx.x;
  const int tn = (int)(bid % g.tiles_n); bid /= g.tiles_n;
  const int tm = (int)(bid % g.tiles_m); bid /= g.tiles_m;
  const int ks = (int)(bid % g.ksplit); bid /= g.ksplit;
  const int64_t b1 = bid / g.nb2, b2 = bid % g.nb2;
  const T* A = (const T*)g.A + b1 * g.bA1 + b2 * g.bA2;
  const T* B = (const T*)g.B + b1 * g.bB1 + b2 * g.bB2;
  const int64_t coff = b1 * g.bC1 + b2 * g.bC2;
  const int64_t m0 = (int64_t)tm * 64;
  const int n0 = tn * 64;
  const int64_t kbeg = (int64_t)ks * g.kchunk, kend = min(g.K, kbeg + g.kchunk);

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 1) * 32, wn = (w & 1) * 32;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool a_kfast = (g.sAk == 1), b_kfast = (g.sBk == 1);
  T zero; st(&zero, 0.f);

  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    // ---- stage A tile [64][BK]
    if (a_kfast) {
      const int kk = tid % BK, mr = tid / BK;
#pragma unroll
      for (int it = 0; it < 64 / (256 / BK); ++it) {
        int m = mr + it * (256 / BK);
        int64_t gm = m0 + m, gk = k0 + kk;
        As[m * LDS_LD + kk] = (gm < g.M && gk < kend) ? A[gm * g.sAm + gk] : zero;
      }
    } else {
      const int m = tid & 63, kr = tid >> 6;
#pragma unroll
      for (int it = 0; it < BK / 4; ++it) {
        int kk = kr + it * 4;
        int64_t gm = m0 + m, gk = k0 + kk;
        As[m * LDS_LD + kk] = (gm < g.M && gk < kend) ? A[gm * g.sAm + gk * g.sAk] : zero;
      }
    }
    // ---- stage B tile as [n][k]
    if (b_kfast) {
      const int kk = tid % BK, nr = tid / BK;
#pragma unroll
      for (int it = 0; it < 64 / (256 / BK); ++it) {
        int n = nr + it * (256 / BK);
        int gn = n0 + n; int64_t gk = k0 + kk;
        int64_t pk = gk; if (g.brow_group > 0) pk = gk + (gk / g.brow_group + 1) * (int64_t)g.brow_skip;
        Bs[n * LDS_LD + kk] = (gn < g.N && gk < kend) ? B[pk + (int64_t)gn * g.sBn] : zero;
      }
    } else {
      const int n = tid & 63, kr = tid >> 6;
#pragma unroll
      for (int it = 0; it < BK / 4; ++it) {
        int kk = kr + it * 4;
        int gn = n0 + n; int64_t gk = k0 + kk;
        int64_t pk = gk; if (g.brow_group > 0) pk = gk + (gk / g.brow_group + 1) * (int64_t)g.brow_skip;
        Bs[n * LDS_LD + kk] = (gn < g.N && gk < kend) ? B[pk * g.sBk + (int64_t)gn * g.sBn] : zero;
      }
    }
    __syncthreads();
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int kk = 0; kk < BK; kk += 4) {
        float a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = ((const float*)As)[(wm + i * 16 + fr) * LDS_LD + kk + fq];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = ((const float*)Bs)[(wn + j * 16 + fr) * LDS_LD + kk + fq];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *(const bf16x8*)((const bf16_t*)As + (wm + i * 16 + fr) * LDS_LD + fq * 8);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *(const bf16x8*)((const bf16_t*)Bs + (wn + j * 16 + fr) * LDS_LD + fq * 8);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = MFMA16(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }

  // ---- epilogue.  C/D map (16x16): col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int64_t gm = m0 + wm + i * 16 + fq * 4 + r;
        int gn = n0 + wn + j * 16 + fr;
        if (gm >= g.M || gn >= g.N) continue;
        float v = g.alpha * acc[i][j][r];
        if (g.bias && ks == 0) v += g.bias[gn];
        int64_t crow = gm; if (g.crow_group > 0) crow = gm + (gm / g.crow_group + 1) * (int64_t)g.crow_skip;
        const int64_t ci = coff + crow * g.sCm + gn;
        if (g.pre_out) st((T*)g.pre_out + ci, v);
        if (g.epi == EPI_GELU) v = gelu_tanh_f(v);
        if (g.aux) {
          float x = ld((const T*)g.aux + ci);
          if (g.epi == EPI_MUL_GELU_GRAD) v *= gelu_tanh_grad_f(x);
          else if (ks == 0) v += x;
        }
        if (g.out_f32) {
          float* cp = (float*)g.C + ci;
          if (g.atomic) grad_add(cp, v);
          else if (g.accumulate) *cp += v;
          else *cp = v;
        } else {
          T* cp = (T*)g.C + ci;
          if (g.accumulate) st(cp, ld(cp) + v); else st(cp, v);
        }
      }
}

template <typename T>
void gemm_generic(spa3d_ctx* c, const GemmDesc& d) {
  if (c->dry || d.M == 0 || d.N == 0) return;
  constexpr int BK = GemmCfg<T>::BK;
  GemmArgs g;
  g.A = d.A; g.B = d.B; g.C = d.C; g.M = d.M; g.N = d.N; g.K = d.K;
  g.sAm = d.sAm; g.sAk = d.sAk; g.sBk = d.sBk; g.sBn = d.sBn; g.sCm = d.sCm;
  g.nb2 = d.nb2; g.bA1 = d.bA1; g.bA2 = d.bA2; g.bB1 = d.bB1; g.bB2 = d.bB2; g.bC1 = d.bC1; g.bC2 = d.bC2;
  g.alpha = d.alpha; g.bias = d.bias; g.epi = d.epi; g.aux = d.aux; g.aux_is_residual = d.aux_is_residual;
  g.out_f32 = d.out_f32; g.accumulate = d.accumulate; g.atomic = 0;
  g.crow_group = d.crow_group; g.crow_skip = d.crow_skip; g.brow_group = d.brow_group; g.brow_skip = d.brow_skip; g.pre_out = d.pre_out;
  g.tiles_m = (int)((d.M + 63) / 64); g.tiles_n = (d.N + 63) / 64;
  int64_t nbatch = (int64_t)d.nb1 * d.nb2;
  int64_t blocks = nbatch * g.tiles_m * g.tiles_n;
  // split-K with f32 atomics when the output is small and the reduction long (dW = X^T dY)
  int ksplit = 1;
  if (d.out_f32 && d.accumulate && d.epi == EPI_NONE && d.aux == nullptr && blocks < 1024 && d.K >= 4096) {
    ksplit = (int)std::min<int64_t>(std::max<int64_t>(1, 2048 / blocks), (d.K + 2047) / 2048);
  }
  g.ksplit = ksplit;
  int64_t kchunk = ((d.K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
  g.kchunk = kchunk;
  if (ksplit > 1) { g.atomic = 1; }
  blocks *= ksplit;
  if (blocks > 0x7fffffffLL) {
    if (!c->hip_err) { c->hip_err = -1; c->err = "gemm_generic: grid too large"; }
    return;
  }
  ProfScope ps(c, PROF_GEMM_GENERIC, 2.0 * (double)d.M * d.N * d.K * (double)nbatch,
               (double)nbatch * ((double)d.M * d.K + (double)d.K * d.N + (double)d.M * d.N) * sizeof(T));
  ps.tag(d.M, d.N, d.K, (int64_t)nbatch * 1000 + (d.sAm == 1 ? 1 : 0) + (d.sBn != 1 ? 2 : 0) + (d.out_f32 ? 4 : 0) + (d.accumulate ? 8 : 0) + (ksplit > 1 ? 16 : 0));
  gemm_generic_kernel<T><<<(unsigned)blocks, 256, 0, c->stream>>>(g);
  SPA_LAUNCH_CHECK(c);
}
template void gemm_generic<float>(spa3d_ctx*, const GemmDesc&);
template void gemm_generic<bf16_t>(spa3d_ctx*, const GemmDesc&);
SPA_DET_UPLOAD_DEF(det_upload_gemm_generic)
}  // namespace SPA_NS
