// kernels.hip -- elementwise / normalisation / reduction / assembly kernels of the 3DSPA hot path (gfx950).
// HBM-bound kernels: one wave per row where a row reduction is needed, coalesced lane-contiguous access,
// fp32 math, T in {float, bf16_t} storage.  References are to /root/reference files.
#include "common.hpp"

namespace SPA_NS {

#define GRID1D(n, bs) dim3((unsigned)std::min<int64_t>(((n) + (bs)-1) / (bs), 1 << 20))

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// LayerNorm (flax nn.LayerNorm(use_bias=False), eps 1e-6, fast variance clamped at 0) attention.py:49,76,103
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale, T* __restrict__ y,
                                                     float* __restrict__ stats, int64_t rows, int d) {
  const int lane = threadIdx.x & 63;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (; row < rows; row += stride) {
    const T* xr = x + row * d;
    float s = 0.f, ss = 0.f;
    for (int i = lane; i < d; i += 64) { float v = ld(xr + i); s += v; ss += v * v; }
    s = wave_sum(s); ss = wave_sum(ss);
    float mu = s / d;
    float var = fmaxf(ss / d - mu * mu, 0.f);
    float r = rsqrtf(var + 1e-6f);
    if (stats && lane == 0) { stats[row * 2] = mu; stats[row * 2 + 1] = r; }
    T* yr = y + row * d;
    for (int i = lane; i < d; i += 64) st(yr + i, (ld(xr + i) - mu) * r * scale[i]);
  }
}
// dx = [add +] r*(g - mean(g) - xhat*mean(g*xhat)), g = dy*scale ; dscale += sum_rows dy*xhat   
#define LN_MAXJ 32  // d <= 2048
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                     const float* __restrict__ stats, const T* __restrict__ dy, const T* add,
                                                     T* dx, float* __restrict__ dscale, int64_t rows, int d) {
  __shared__ float red[4][64 * LN_MAXJ / 4];  // reused per quarter below
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float acc[LN_MAXJ];
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) acc[j] = 0.f;
  int64_t row = (int64_t)blockIdx.x * 4 + w;
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (; row < rows; row += stride) {
    const T* xr = x + row * d;
    const T* dyr = dy + row * d;
    const float mu = stats[row * 2], r = stats[row * 2 + 1];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
      int i = lane + 64 * j;
      if (i < d) {
        float xh = (ld(xr + i) - mu) * r;
        float dyv = ld(dyr + i);
        float g = dyv * scale[i];
        sg += g; sgx += g * xh;
        acc[j] += dyv * xh;
      }
    }
    sg = wave_sum(sg) / d; sgx = wave_sum(sgx) / d;
    T* dxr = dx + row * d;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
      int i = lane + 64 * j;
      if (i < d) {
        float xh = (ld(xr + i) - mu) * r;
        float g = ld(dyr + i) * scale[i];
        float v = r * (g - sg - xh * sgx);
        if (add) v += ld(add + row * d + i);
        st(dxr + i, v);
      }
    }
  }
  // block reduce acc over the 4 waves, 8 columns-groups at a time to bound LDS
  for (int j0 = 0; j0 < LN_MAXJ; j0 += 8) {
    if (j0 * 64 >= d) break;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) red[w][j * 64 + lane] = acc[j0 + j];
    __syncthreads();
    if (w == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int i = lane + 64 * (j0 + j);
        if (i < d) {
          float s = red[0][j * 64 + lane] + red[1][j * 64 + lane] + red[2][j * 64 + lane] + red[3][j * 64 + lane];
          grad_add(dscale + i, s);
        }
      }
    }
  }
}
// vectorised variant: 16-byte loads (8 bf16 / 4 f32 per lane per step); lanes own fixed columns so the scale-gradient
// partials stay in registers across the rows a wave walks.  Needs d % VEC == 0 and d <= 2048.
template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<bf16_t> { static constexpr int N = 8; };
template <typename T, int NV>
__device__ __forceinline__ void load_vec(const T* p, float (&f)[NV]) {
  if constexpr (sizeof(T) == 4) { const float4 v = *(const float4*)p; f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
  else { const uint4 v = *(const uint4*)p; const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = unpack_lo(u[i]); f[2 * i + 1] = unpack_hi(u[i]); } }
}
template <typename T, int NV>
__device__ __forceinline__ void store_vec(T* p, const float (&f)[NV]) {
  if constexpr (sizeof(T) == 4) { *(float4*)p = make_float4(f[0], f[1], f[2], f[3]); }
  else { unsigned u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = f2bf_pack2(f[2 * i], f[2 * i + 1]);
    *(uint4*)p = make_uint4(u[0], u[1], u[2], u[3]); }
}
// the 16 bytes of load_vec kept packed (half the registers of the unpacked floats while several rows are in flight)
template <typename T, int NV>
__device__ __forceinline__ void unpack_vec(const uint4& v, float (&f)[NV]) {
  if constexpr (sizeof(T) == 4) { f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w); }
  else { const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = unpack_lo(u[i]); f[2 * i + 1] = unpack_hi(u[i]); } }
}
// forward, vectorised: a wave per row, 16-byte loads held in registers between the statistics and the normalisation (the row is read
// once: the scalar kernel above reads it twice with 2-byte loads and measured 2.2 TB/s), two rows in flight per wave.
template <typename T, int STEPS, int U = 2>
__global__ __launch_bounds__(256) void ln_fwd_vec_kernel(const T* __restrict__ x, const float* __restrict__ scale, T* __restrict__ y,
                                                         float* __restrict__ stats, int64_t rows, int d) {
  constexpr int NV = VecOf<T>::N;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nch = d / NV;
  float sc[STEPS][NV];
#pragma unroll
  for (int s_ = 0; s_ < STEPS; ++s_) {
    const int c = lane + 64 * s_;
#pragma unroll
    for (int j = 0; j < NV; ++j) sc[s_][j] = c < nch ? scale[c * NV + j] : 0.f;
  }
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t row0 = (int64_t)blockIdx.x * 4 + w; row0 < rows; row0 += stride * U) {
    float xv[U][STEPS][NV];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = row0 + u * stride;
#pragma unroll
      for (int s_ = 0; s_ < STEPS; ++s_) {
        const int c = lane + 64 * s_;
        if (row < rows && c < nch) load_vec<T, NV>(x + row * d + c * NV, xv[u][s_]);
        else {
#pragma unroll
          for (int j = 0; j < NV; ++j) xv[u][s_][j] = 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = row0 + u * stride;
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int s_ = 0; s_ < STEPS; ++s_)
#pragma unroll
        for (int j = 0; j < NV; ++j) { s += xv[u][s_][j]; ss += xv[u][s_][j] * xv[u][s_][j]; }
      s = wave_sum(s); ss = wave_sum(ss);
      const float mu = s / d;
      const float var = fmaxf(ss / d - mu * mu, 0.f);
      const float r = rsqrtf(var + 1e-6f);
      if (row < rows) {
        if (stats && lane == 0) { stats[row * 2] = mu; stats[row * 2 + 1] = r; }
#pragma unroll
        for (int s_ = 0; s_ < STEPS; ++s_) {
          const int c = lane + 64 * s_;
          if (c < nch) {
            float o[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) o[j] = (xv[u][s_][j] - mu) * r * sc[s_][j];
            store_vec<T, NV>(y + row * d + c * NV, o);
          }
        }
      }
    }
  }
}
// d = LPR * CH * NV: a row per LPR lanes (16 at d = 384, 32 at d = 1280 in 16-bit types), 64 / LPR consecutive rows per wave and step -- every lane
// works (a wave per 768-byte row leaves 16 of 64 lanes idle) and a wave keeps 64 / LPR rows in flight: 4.5 -> 5.2 TB/s at d = 384.
template <typename T, int LPR, int CH>
__global__ __launch_bounds__(256) void ln_fwd_part_kernel(const T* __restrict__ x, const float* __restrict__ scale, T* __restrict__ y,
                                                          float* __restrict__ stats, int64_t rows, int d) {
  constexpr int NV = VecOf<T>::N, RPB = 256 / LPR;
  const int sub = threadIdx.x & (LPR - 1);
  float sc[CH][NV];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int j = 0; j < NV; ++j) sc[c][j] = scale[(sub + LPR * c) * NV + j];
  for (int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR; row < rows; row += (int64_t)gridDim.x * RPB) {
    uint4 xr[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) xr[c] = *(const uint4*)(x + row * d + (sub + LPR * c) * NV);
    float xv[CH][NV]; float s = 0.f, ss = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      unpack_vec<T, NV>(xr[c], xv[c]);
#pragma unroll
      for (int j = 0; j < NV; ++j) { s += xv[c][j]; ss += xv[c][j] * xv[c][j]; }
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
    const float mu = s / d;
    const float var = fmaxf(ss / d - mu * mu, 0.f);
    const float r = rsqrtf(var + 1e-6f);
    if (stats && sub == 0) { stats[row * 2] = mu; stats[row * 2 + 1] = r; }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float o[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) o[j] = (xv[c][j] - mu) * r * sc[c][j];
      store_vec<T, NV>(y + row * d + (sub + LPR * c) * NV, o);
    }
  }
}
// backward in the same row partition (d = 384: LPR = 16, CH = 3)
template <typename T, int LPR, int CH>
__global__ __launch_bounds__(256) void ln_bwd_part_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ stats,
                                                          const T* __restrict__ dy, const T* add, T* dx, float* __restrict__ dscale, int64_t rows, int d) {
  constexpr int NV = VecOf<T>::N, RPB = 256 / LPR, DV = LPR * CH * NV;
  __shared__ float red[4][DV];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, sub = threadIdx.x & (LPR - 1);
  float acc[CH][NV], sc[CH][NV];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int j = 0; j < NV; ++j) { acc[c][j] = 0.f; sc[c][j] = scale[(sub + LPR * c) * NV + j]; }
  for (int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR; row < rows; row += (int64_t)gridDim.x * RPB) {
    uint4 xr[CH], dr[CH], ar[CH];
    const float mu = stats[row * 2], r = stats[row * 2 + 1];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int64_t o = row * d + (sub + LPR * c) * NV;
      xr[c] = *(const uint4*)(x + o); dr[c] = *(const uint4*)(dy + o);
      ar[c] = add ? *(const uint4*)(add + o) : make_uint4(0u, 0u, 0u, 0u);
    }
    float xh[CH][NV], gg[CH][NV]; float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float xv[NV], dv[NV];
      unpack_vec<T, NV>(xr[c], xv); unpack_vec<T, NV>(dr[c], dv);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        xh[c][j] = (xv[j] - mu) * r; gg[c][j] = dv[j] * sc[c][j];
        sg += gg[c][j]; sgx += gg[c][j] * xh[c][j]; acc[c][j] += dv[j] * xh[c][j];
      }
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) { sg += __shfl_xor(sg, o, 64); sgx += __shfl_xor(sgx, o, 64); }
    sg /= d; sgx /= d;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float o[NV], av[NV];
      unpack_vec<T, NV>(ar[c], av);
#pragma unroll
      for (int j = 0; j < NV; ++j) o[j] = r * (gg[c][j] - sg - xh[c][j] * sgx) + av[j];
      store_vec<T, NV>(dx + row * d + (sub + LPR * c) * NV, o);
    }
  }
  // scale gradient: the 64 / LPR row groups of a wave, then the four waves, then one atomic per column and workgroup
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      float a = acc[c][j];
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
      if (lane < LPR) red[w][(sub + LPR * c) * NV + j] = a;
    }
  __syncthreads();
  for (int t = threadIdx.x; t < DV; t += 256) grad_add(dscale + t, red[0][t] + red[1][t] + red[2][t] + red[3][t]);
}
template <typename T>
void k_layernorm(spa3d_ctx* c, const T* x, const float* scale, T* y, float* stats, int64_t rows, int d) {
  if (c->dry || rows == 0) return;
  ProfScope ps(c, PROF_LN_FWD, 8.0 * (double)rows * d, (double)rows * d * 2.0 * sizeof(T) + rows * 8.0);  // read x, write y (+ stats)
  ps.tag(rows, d, 0, 0);
  constexpr int NV = VecOf<T>::N;
  const bool al = ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
  if (d % NV == 0 && al && d <= 64 * NV * 4) {
    constexpr int gcap = 4096;  // measured (tools/bench_ln.py)
    const unsigned g = (unsigned)std::min<int64_t>(cdiv(rows, 8), gcap);
    const int steps = (d / NV + 63) / 64;
    if constexpr (sizeof(T) == 2) {  // the two widths of the step's large LayerNorms: row-partitioned kernels (4.6 -> 5.3 and 3.9 -> 5.2 TB/s)
      if (d == 384) { ln_fwd_part_kernel<T, 16, 3><<<(unsigned)std::min<int64_t>(cdiv(rows, 16), 2 * gcap), 256, 0, c->stream>>>(x, scale, y, stats, rows, d); SPA_LAUNCH_CHECK(c); return; }
      if (d == 1280) { ln_fwd_part_kernel<T, 32, 5><<<(unsigned)std::min<int64_t>(cdiv(rows, 8), 2 * gcap), 256, 0, c->stream>>>(x, scale, y, stats, rows, d); SPA_LAUNCH_CHECK(c); return; }
    }
    if (steps == 1) ln_fwd_vec_kernel<T, 1><<<g, 256, 0, c->stream>>>(x, scale, y, stats, rows, d);  // (four rows in flight per wave measured 3.5 vs 4.4 TB/s: occupancy)
    else if (steps == 2) ln_fwd_vec_kernel<T, 2><<<g, 256, 0, c->stream>>>(x, scale, y, stats, rows, d);
    else if (steps == 3) ln_fwd_vec_kernel<T, 3><<<g, 256, 0, c->stream>>>(x, scale, y, stats, rows, d);
    else ln_fwd_vec_kernel<T, 4><<<g, 256, 0, c->stream>>>(x, scale, y, stats, rows, d);
  } else {
    const unsigned g = (unsigned)std::min<int64_t>(cdiv(rows, 4), 65536);
    ln_fwd_kernel<T><<<g, 256, 0, c->stream>>>(x, scale, y, stats, rows, d);
  }
  SPA_LAUNCH_CHECK(c);
}

template <typename T, int STEPS, int U>
__global__ __launch_bounds__(256) void ln_bwd_vec_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ stats, const T* __restrict__ dy, const T* add, T* dx,
                                                         float* __restrict__ dscale, int64_t rows, int d) {
  constexpr int NV = VecOf<T>::N;
  __shared__ float red[4][64 * NV];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nch = d / NV;
  float acc[STEPS][NV], sc[STEPS][NV];
#pragma unroll
  for (int s_ = 0; s_ < STEPS; ++s_) {
    const int c = lane + 64 * s_;
#pragma unroll
    for (int j = 0; j < NV; ++j) { acc[s_][j] = 0.f; sc[s_][j] = c < nch ? scale[c * NV + j] : 0.f; }
  }
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t row0 = (int64_t)blockIdx.x * 4 + w; row0 < rows; row0 += stride * U) {
    // every load of U rows (x, dy and the residual-path gradient) is requested before the first is used
    uint4 xr[U][STEPS], dr[U][STEPS], ar[U][STEPS]; float mu[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = row0 + u * stride;
      const bool live = row < rows;
      mu[u] = live ? stats[row * 2] : 0.f; r[u] = live ? stats[row * 2 + 1] : 0.f;
#pragma unroll
      for (int s_ = 0; s_ < STEPS; ++s_) {
        const int c = lane + 64 * s_;
        xr[u][s_] = dr[u][s_] = ar[u][s_] = make_uint4(0u, 0u, 0u, 0u);
        if (live && c < nch) {
          xr[u][s_] = *(const uint4*)(x + row * d + c * NV); dr[u][s_] = *(const uint4*)(dy + row * d + c * NV);
          if (add) ar[u][s_] = *(const uint4*)(add + row * d + c * NV);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = row0 + u * stride;
      float xh[STEPS][NV], gg[STEPS][NV];
      float sg = 0.f, sgx = 0.f;
#pragma unroll
      for (int s_ = 0; s_ < STEPS; ++s_) {
        float xv[NV], dv[NV];
        unpack_vec<T, NV>(xr[u][s_], xv); unpack_vec<T, NV>(dr[u][s_], dv);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          xh[s_][j] = (xv[j] - mu[u]) * r[u]; gg[s_][j] = dv[j] * sc[s_][j];
          sg += gg[s_][j]; sgx += gg[s_][j] * xh[s_][j]; acc[s_][j] += dv[j] * xh[s_][j];
        }
      }
      sg = wave_sum(sg) / d; sgx = wave_sum(sgx) / d;
      if (row < rows) {
#pragma unroll
        for (int s_ = 0; s_ < STEPS; ++s_) {
          const int c = lane + 64 * s_;
          if (c < nch) {
            float o[NV], av[NV];
            unpack_vec<T, NV>(ar[u][s_], av);
#pragma unroll
            for (int j = 0; j < NV; ++j) o[j] = r[u] * (gg[s_][j] - sg - xh[s_][j] * sgx) + av[j];
            store_vec<T, NV>(dx + row * d + c * NV, o);
          }
        }
      }
    }
  }
#pragma unroll
  for (int s_ = 0; s_ < STEPS; ++s_) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j) red[w][lane * NV + j] = acc[s_][j];
    __syncthreads();
    if (w == 0) {
      const int c = lane + 64 * s_;
      if (c < nch)
#pragma unroll
        for (int j = 0; j < NV; ++j)
          grad_add(dscale + c * NV + j, red[0][lane * NV + j] + red[1][lane * NV + j] + red[2][lane * NV + j] + red[3][lane * NV + j]);
    }
  }
}
template <typename T>
void k_layernorm_bwd(spa3d_ctx* c, const T* x, const float* scale, const float* stats, const T* dy, T* dx, float* dscale,
                     int64_t rows, int d, const T* add) {
  if (c->dry || rows == 0) return;
  ProfScope ps(c, PROF_LN_BWD, 16.0 * (double)rows * d, (double)rows * d * (add ? 4.0 : 3.0) * sizeof(T) + rows * 8.0);  // x, dy, (add), dx
  ps.tag(rows, d, add ? 1 : 0, 0);
  constexpr int gcapb = 1024;
  unsigned g = (unsigned)std::min<int64_t>(cdiv(rows, 4), d <= 512 ? gcapb : 2 * gcapb);  // measured: 1024 blocks at d = 384, 2048 at d = 1280
  // few rows (the latent stacks: 1 408): one block per 4 rows means 352 blocks each adding its d partial sums into the SAME d addresses -- 103 us for 3 MB;
  // 32 rows per block there
  if (rows <= 16384) g = (unsigned)std::min<int64_t>(g, std::max<int64_t>(64, cdiv(rows, 32)));
  constexpr int NV = VecOf<T>::N;
  const bool al = ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)add)) & 15) == 0;
  if (d % NV == 0 && al && d <= 64 * NV * 4) {
    const int steps = (d / NV + 63) / 64;
    if constexpr (sizeof(T) == 2) {  // row-partitioned kernels: 4.9 -> 5.2 TB/s at d = 384 (2.8 -> 5.3 below 0.5 M rows), 4.85 -> 5.45 at d = 1280
      if (d == 384) { ln_bwd_part_kernel<T, 16, 3><<<(unsigned)std::min<int64_t>(cdiv(rows, 16), 2 * gcapb), 256, 0, c->stream>>>(x, scale, stats, dy, add, dx, dscale, rows, d); SPA_LAUNCH_CHECK(c); return; }
      if (d == 1280) { ln_bwd_part_kernel<T, 32, 5><<<(unsigned)std::min<int64_t>(cdiv(rows, 8), 2 * gcapb), 256, 0, c->stream>>>(x, scale, stats, dy, add, dx, dscale, rows, d); SPA_LAUNCH_CHECK(c); return; }
    }
    if (steps == 1) ln_bwd_vec_kernel<T, 1, 1><<<g, 256, 0, c->stream>>>(x, scale, stats, dy, add, dx, dscale, rows, d);  // U = 4 measured 3.7 vs 4.7 TB/s
    else if (steps == 2) ln_bwd_vec_kernel<T, 2, 2><<<g, 256, 0, c->stream>>>(x, scale, stats, dy, add, dx, dscale, rows, d);
    else if (steps == 3) ln_bwd_vec_kernel<T, 3, 2><<<g, 256, 0, c->stream>>>(x, scale, stats, dy, add, dx, dscale, rows, d);
    else ln_bwd_vec_kernel<T, 4, 1><<<g, 256, 0, c->stream>>>(x, scale, stats, dy, add, dx, dscale, rows, d);
  } else {
    ln_bwd_kernel<T><<<g, 256, 0, c->stream>>>(x, scale, stats, dy, add, dx, dscale, rows, d);
  }
  SPA_LAUNCH_CHECK(c);
}

// ---------------------------------------------------------------------------------------------
// per-head RMSNorm over Dh (flax nn.RMSNorm eps 1e-6) attention.py:166-167.  one wave per (row, head)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rms_heads_fwd_kernel(const T* __restrict__ x, int64_t ldx, const float* __restrict__ scale,
                                                            T* __restrict__ y, int64_t ldy, int64_t rows, int H, int Dh) {
  const int lane = threadIdx.x & 63;
  int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n = rows * H, stride = (int64_t)gridDim.x * 4;
  for (; item < n; item += stride) {
    int64_t row = item / H; int h = (int)(item - row * H);
    const T* xr = x + row * ldx + h * Dh;
    float v0 = lane < Dh ? ld(xr + lane) : 0.f;
    float v1 = lane + 64 < Dh ? ld(xr + lane + 64) : 0.f;
    float ms = wave_sum(v0 * v0 + v1 * v1) / Dh;
    float r = rsqrtf(ms + 1e-6f);
    T* yr = y + row * ldy + h * Dh;
    if (lane < Dh) st(yr + lane, v0 * r * scale[lane]);
    if (lane + 64 < Dh) st(yr + lane + 64, v1 * r * scale[lane + 64]);
  }
}
template <typename T>
void k_rmsnorm_heads(spa3d_ctx* c, const T* x, int64_t ldx, const float* scale, T* y, int64_t ldy, int64_t rows, int H, int Dh) {
  if (c->dry || rows == 0) return;
  unsigned g = (unsigned)std::min<int64_t>(cdiv(rows * H, 4), 65536);
  rms_heads_fwd_kernel<T><<<g, 256, 0, c->stream>>>(x, ldx, scale, y, ldy, rows, H, Dh);
  SPA_LAUNCH_CHECK(c);
}
// dx = r*(g - xhat*mean(g*xhat)), g=dy*scale ; dscale += sum dy*xhat
template <typename T>
__global__ __launch_bounds__(256) void rms_heads_bwd_kernel(const T* __restrict__ x, int64_t ldx, const float* __restrict__ scale,
                                                            const T* __restrict__ dy, int64_t lddy, T* __restrict__ dx, int64_t lddx,
                                                            float* __restrict__ dscale, int64_t rows, int H, int Dh) {
  __shared__ float red[4][128];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float a0 = 0.f, a1 = 0.f;
  int64_t item = (int64_t)blockIdx.x * 4 + w;
  const int64_t n = rows * H, stride = (int64_t)gridDim.x * 4;
  const float s0 = lane < Dh ? scale[lane] : 0.f, s1 = lane + 64 < Dh ? scale[lane + 64] : 0.f;
  for (; item < n; item += stride) {
    int64_t row = item / H; int h = (int)(item - row * H);
    const T* xr = x + row * ldx + h * Dh;
    const T* dyr = dy + row * lddy + h * Dh;
    float v0 = lane < Dh ? ld(xr + lane) : 0.f, v1 = lane + 64 < Dh ? ld(xr + lane + 64) : 0.f;
    float d0 = lane < Dh ? ld(dyr + lane) : 0.f, d1 = lane + 64 < Dh ? ld(dyr + lane + 64) : 0.f;
    float r = rsqrtf(wave_sum(v0 * v0 + v1 * v1) / Dh + 1e-6f);
    float xh0 = v0 * r, xh1 = v1 * r;
    float g0 = d0 * s0, g1 = d1 * s1;
    float mg = wave_sum(g0 * xh0 + g1 * xh1) / Dh;
    a0 += d0 * xh0; a1 += d1 * xh1;
    T* dxr = dx + row * lddx + h * Dh;
    if (lane < Dh) st(dxr + lane, r * (g0 - xh0 * mg));
    if (lane + 64 < Dh) st(dxr + lane + 64, r * (g1 - xh1 * mg));
  }
  red[w][lane] = a0; red[w][lane + 64] = a1;
  __syncthreads();
  if (w == 0) {
    if (lane < Dh) grad_add(dscale + lane, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
    if (lane + 64 < Dh) grad_add(dscale + lane + 64, red[0][lane + 64] + red[1][lane + 64] + red[2][lane + 64] + red[3][lane + 64]);
  }
}
template <typename T>
void k_rmsnorm_heads_bwd(spa3d_ctx* c, const T* x, int64_t ldx, const float* scale, const T* dy, int64_t lddy, T* dx, int64_t lddx,
                         float* dscale, int64_t rows, int H, int Dh) {
  if (c->dry || rows == 0) return;
  unsigned g = (unsigned)std::min<int64_t>(cdiv(rows * H, 4), 2048);
  rms_heads_bwd_kernel<T><<<g, 256, 0, c->stream>>>(x, ldx, scale, dy, lddy, dx, lddx, dscale, rows, H, Dh);
  SPA_LAUNCH_CHECK(c);
}

// ---------------------------------------------------------------------------------------------
// softmax over keys with key mask: where(mask, logit, finfo.min) -> softmax (flax dot_product_attention)
// s: [nseq][H][Sq][Sk] in place.  one wave per row.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_kernel(T* __restrict__ s, const float* __restrict__ km, int64_t nseq, int H, int Sq, int Sk) {
  const int lane = threadIdx.x & 63;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nrows = nseq * H * Sq, stride = (int64_t)gridDim.x * 4;
  for (; row < nrows; row += stride) {
    int64_t seq = row / ((int64_t)H * Sq);
    T* sr = s + row * Sk;
    const float* kmr = km ? km + seq * Sk : nullptr;
    float m = -3.4028234663852886e38f;
    for (int k = lane; k < Sk; k += 64) {
      float v = ld(sr + k);
      if (kmr && kmr[k] == 0.f) v = -3.4028234663852886e38f;
      m = fmaxf(m, v);
    }
    m = wave_max(m);
    float sum = 0.f;
    for (int k = lane; k < Sk; k += 64) {
      float v = ld(sr + k);
      if (kmr && kmr[k] == 0.f) v = -3.4028234663852886e38f;
      sum += expf(v - m);
    }
    sum = wave_sum(sum);
    float inv = 1.f / sum;
    for (int k = lane; k < Sk; k += 64) {
      float v = ld(sr + k);
      if (kmr && kmr[k] == 0.f) v = -3.4028234663852886e38f;
      st(sr + k, expf(v - m) * inv);
    }
  }
}
template <typename T>
void k_softmax(spa3d_ctx* c, T* s, const float* keymask, int64_t nseq, int H, int Sq, int Sk) {
  if (c->dry || nseq == 0) return;
  unsigned g = (unsigned)std::min<int64_t>(cdiv(nseq * H * Sq, 4), 65536);
  softmax_kernel<T><<<g, 256, 0, c->stream>>>(s, keymask, nseq, H, Sq, Sk);
  SPA_LAUNCH_CHECK(c);
}
// dS = P o (dP - rowsum(dP o P)), in place on dp; masked keys get exactly 0: where(mask, logit, min) passes them no gradient
// (this matters only for fully masked rows, where P is uniform instead of 0)
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ p, T* __restrict__ dp, int64_t rows, int Sk,
                                                          const float* __restrict__ km, int64_t rows_per_seq) {
  const int lane = threadIdx.x & 63;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (; row < rows; row += stride) {
    const T* pr = p + row * Sk; T* dr = dp + row * Sk;
    float s = 0.f;
    for (int k = lane; k < Sk; k += 64) s += ld(pr + k) * ld(dr + k);
    s = wave_sum(s);
    const float* kmr = km ? km + (row / rows_per_seq) * Sk : nullptr;
    for (int k = lane; k < Sk; k += 64) {
      float pv = ld(pr + k);
      float v = pv * (ld(dr + k) - s);
      if (kmr && kmr[k] == 0.f) v = 0.f;
      st(dr + k, v);
    }
  }
}
template <typename T>
void k_softmax_bwd(spa3d_ctx* c, const T* p, T* dp, int64_t rows, int Sk, const float* km, int64_t rows_per_seq) {
  if (c->dry || rows == 0) return;
  unsigned g = (unsigned)std::min<int64_t>(cdiv(rows, 4), 65536);
  softmax_bwd_kernel<T><<<g, 256, 0, c->stream>>>(p, dp, rows, Sk, km, rows_per_seq);
  SPA_LAUNCH_CHECK(c);
}

// ---------------------------------------------------------------------------------------------
// SinusoidalEmbedding (track_autoencoder.py:18-38): v = fl32(x*s_f); out = sin([v, fl32(v + fl32(pi/2))])
// layout "(coords d)": out[r][c*2nf + f] , out[r][c*2nf + nf + f].  Never fused to fma, never cos.
// ---------------------------------------------------------------------------------------------
struct SinScales { float s[64]; };
static SinScales make_scales(int nf) {
  SinScales sc;
  for (int i = 0; i < 64; ++i) sc.s[i] = i < nf ? (float)pow(2.0, (double)i / 3.0) : 0.f;
  return sc;
}
__device__ __forceinline__ float sin_feat(float x, float s, bool shifted) {
  float v = __fmul_rn(x, s);
  if (shifted) v = __fadd_rn(v, 1.57079637050628662109375f);  // fl32(0.5*pi)
  return sinf(v);
}
template <typename T>
__global__ __launch_bounds__(256) void sin_embed_kernel(const float* __restrict__ x, int64_t rows, int C, int nf, float prescale,
                                                        SinScales sc, T* __restrict__ out) {
  const int W = C * 2 * nf;
  const int64_t n = rows * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int64_t r = i / W; int j = (int)(i - r * W);
    int cc = j / (2 * nf); int f = j - cc * 2 * nf;
    float xv = x[r * C + cc] / prescale;
    st(out + i, sin_feat(xv, sc.s[f < nf ? f : f - nf], f >= nf));
  }
}
template <typename T>
void k_sin_embed(spa3d_ctx* c, const float* x, int64_t rows, int C, int nf, float prescale, T* out) {
  if (c->dry || rows == 0) return;
  sin_embed_kernel<T><<<GRID1D(rows * C * 2 * nf, 256), 256, 0, c->stream>>>(x, rows, C, nf, prescale, make_scales(nf), out);
  SPA_LAUNCH_CHECK(c);
}

// E1+E2 for the track tokens (track_autoencoder_3d.py:126-134): x4 = [x,y,z,t/T] -> sinbuf[nseq*T][4*2nf]
template <typename T>
__global__ __launch_bounds__(256) void embed_tokens_kernel(const float* __restrict__ tracks, int64_t nrows, int T_, int nf, float prescale,
                                                           SinScales sc, T* __restrict__ out, int NC) {
  const int W = (NC + 1) * 2 * nf;
  const int64_t n = nrows * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int64_t r = i / W; int j = (int)(i - r * W);
    int cc = j / (2 * nf); int f = j - cc * 2 * nf;
    float xv;
    if (cc < NC) xv = tracks[r * NC + cc];
    else xv = (float)(int)(r % T_) / (float)T_;  // jnp.arange(T)/T
    xv = xv / prescale;
    st(out + i, sin_feat(xv, sc.s[f < nf ? f : f - nf], f >= nf));
  }
}
// vectorised: 8 consecutive features (one coordinate, one sin/cos half) per thread, one 16-B store, 32-bit index math, the scale table
// in LDS (indexing the by-value kernel argument per lane went through scratch).  Same arithmetic as sin_feat(): results are identical.
template <typename T>
__global__ __launch_bounds__(256) void embed_tokens_vec_kernel(const float* __restrict__ tracks, unsigned nrows, int T_, int nf, float prescale,
                                                               SinScales sc, T* __restrict__ out, int NC) {
  constexpr int NV = VecOf<T>::N;
  __shared__ float ssc[64];
  if (threadIdx.x < 64) ssc[threadIdx.x] = sc.s[threadIdx.x];
  __syncthreads();
  const unsigned W = (unsigned)(NC + 1) * 2u * nf, gpr = W / NV;   // groups per row
  const unsigned total = nrows * gpr;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned r = i / gpr; const int j0 = (int)(i - r * gpr) * NV;
    const int cc = j0 / (2 * nf); const int f0 = j0 - cc * 2 * nf;
    const bool shifted = f0 >= nf; const int fb = shifted ? f0 - nf : f0;
    float xv;
    if (cc < NC) xv = tracks[(int64_t)r * NC + cc];
    else xv = (float)(int)(r % (unsigned)T_) / (float)T_;  // jnp.arange(T)/T
    xv = xv / prescale;
    float o[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) o[e] = sin_feat(xv, ssc[fb + e], shifted);
    store_vec<T, NV>(out + (int64_t)r * W + j0, o);
  }
}
template <typename T>
void k_embed_tokens(spa3d_ctx* c, const float* tracks, int64_t nrows, int T_, int nf, float prescale, T* sinbuf, int NC) {
  if (c->dry || nrows == 0) return;
  constexpr int NV = VecOf<T>::N;
  const int64_t W = (int64_t)(NC + 1) * 2 * nf;
  if (nf % NV == 0 && nf <= 64 && nrows * (W / NV) < 0x7fffffffLL && nrows < 0x7fffffffLL && (((uintptr_t)sinbuf) & 15) == 0) {
    embed_tokens_vec_kernel<T><<<GRID1D(nrows * (W / NV), 256), 256, 0, c->stream>>>(tracks, (unsigned)nrows, T_, nf, prescale, make_scales(nf), sinbuf, NC);
  } else {
    embed_tokens_kernel<T><<<GRID1D(nrows * W, 256), 256, 0, c->stream>>>(tracks, nrows, T_, nf, prescale, make_scales(nf), sinbuf, NC);
  }
  SPA_LAUNCH_CHECK(c);
}

// get_decoder_context + first-level query features (track_autoencoder_3d.py:209-233,265-272):
// feat[q][0:6nf] = sin-embed(xyz/track_scale); feat[q][6nf] = floor(round(t)/time_scale); qframe = round(t) (half-even)
__global__ __launch_bounds__(256) void query_embed1_kernel(const float* __restrict__ qp, int64_t nq, int nf, float track_scale,
                                                           float time_scale, SinScales sc, float* __restrict__ feat, int32_t* __restrict__ qframe,
                                                           int NC) {
  const int W = NC * 2 * nf + 1;
  const int64_t n = nq * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int64_t q = i / W; int j = (int)(i - q * W);
    if (j == NC * 2 * nf) {
      int32_t fr = (int32_t)rintf(qp[q * (NC + 1)]);
      qframe[q] = fr;
      feat[i] = floorf((float)fr / time_scale);
    } else {
      int cc = j / (2 * nf); int f = j - cc * 2 * nf;
      float xv = qp[q * (NC + 1) + 1 + cc] / track_scale;
      feat[i] = sin_feat(xv, sc.s[f < nf ? f : f - nf], f >= nf);
    }
  }
}
void k_query_embed1(spa3d_ctx* c, const float* qp, int64_t nq, int nf, float track_scale, float time_scale, float* feat, int32_t* qframe,
                    int NC) {
  if (c->dry || nq == 0) return;
  query_embed1_kernel<<<GRID1D(nq * (NC * 2 * nf + 1), 256), 256, 0, c->stream>>>(qp, nq, nf, track_scale, time_scale, make_scales(nf), feat, qframe, NC);
  SPA_LAUNCH_CHECK(c);
}

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, int64_t rows, int n, int64_t ld_, float* __restrict__ out, int64_t rows_per_block,
                              int rgroup, int rskip) {
  // block (64 cols x 4 row-lanes); grid (ceil(n/64), row_splits)
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = std::min<int64_t>(rows, r0 + rows_per_block);
  float s = 0.f;
  if (col < n) for (int64_t r = r0 + w; r < r1; r += 4) {
    int64_t pr = r; if (rgroup > 0) pr = r + (r / rgroup + 1) * (int64_t)rskip;
    s += ld(x + pr * ld_ + col);
  }
  red[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && col < n) grad_add(out + col, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// vectorised: a thread owns one 16-byte column chunk and walks rows; block = 32 chunks x 8 row lanes
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ x, int64_t rows, int n, int64_t ld_, float* __restrict__ out,
                                                         int64_t rows_per_block, int rgroup, int rskip) {
  constexpr int NV = VecOf<T>::N;
  __shared__ float red[8][32 * NV];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int ch = blockIdx.x * 32 + cx;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = std::min<int64_t>(rows, r0 + rows_per_block);
  float a[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) a[j] = 0.f;
  if (ch * NV < n)
    for (int64_t r = r0 + ry; r < r1; r += 8) {
      int64_t pr = r; if (rgroup > 0) pr = r + (r / rgroup + 1) * (int64_t)rskip;
      float v[NV]; load_vec<T, NV>(x + pr * ld_ + ch * NV, v);
#pragma unroll
      for (int j = 0; j < NV; ++j) a[j] += v[j];
    }
#pragma unroll
  for (int j = 0; j < NV; ++j) red[ry][cx * NV + j] = a[j];
  __syncthreads();
  if (ry == 0 && ch * NV < n)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      float s_ = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s_ += red[k][cx * NV + j];
      grad_add(out + ch * NV + j, s_);
    }
}
template <typename T>
void k_colsum(spa3d_ctx* c, const T* x, int64_t rows, int n, int64_t ld_, float* out, int rgroup, int rskip) {
  if (c->dry || rows == 0) return;
  constexpr int NV = VecOf<T>::N;
  if (n % NV == 0 && ld_ % NV == 0 && (((uintptr_t)x) & 15) == 0) {
    const int64_t gx = cdiv(n / NV, 32);
    int64_t sp = std::max<int64_t>(1, std::min<int64_t>(cdiv(rows, 64), 2048 / gx + 1));   // (was rows / 512: at 1 408 rows three row splits, 59 dependent loads per thread)
    int64_t rpb_ = cdiv(rows, sp);
    colsum_vec_kernel<T><<<dim3((unsigned)gx, (unsigned)cdiv(rows, rpb_)), 256, 0, c->stream>>>(x, rows, n, ld_, out, rpb_, rgroup, rskip);
    SPA_LAUNCH_CHECK(c);
    return;
  }
  int64_t splits = std::max<int64_t>(1, std::min<int64_t>(cdiv(rows, 256), 1024 / std::max<int64_t>(1, cdiv(n, 64)) + 1));
  int64_t rpb = cdiv(rows, splits);
  colsum_kernel<T><<<dim3((unsigned)cdiv(n, 64), (unsigned)cdiv(rows, rpb)), 256, 0, c->stream>>>(x, rows, n, ld_, out, rpb, rgroup, rskip);
  SPA_LAUNCH_CHECK(c);
}

template <typename T>
__global__ void gelu_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) st(y + i, gelu_tanh_f(ld(x + i)));
}
template <typename T> void k_gelu(spa3d_ctx* c, const T* x, T* y, int64_t n) {
  if (c->dry || n == 0) return;
  gelu_kernel<T><<<GRID1D(n, 256), 256, 0, c->stream>>>(x, y, n); SPA_LAUNCH_CHECK(c);
}
template <typename T>
__global__ void add_kernel(T* __restrict__ dst, const T* __restrict__ src, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) st(dst + i, ld(dst + i) + ld(src + i));
}
template <typename T> void k_add(spa3d_ctx* c, T* dst, const T* src, int64_t n) {
  if (c->dry || n == 0) return;
  add_kernel<T><<<GRID1D(n, 256), 256, 0, c->stream>>>(dst, src, n); SPA_LAUNCH_CHECK(c);
}
__global__ void fill_kernel(float* p, float v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}
void k_fill(spa3d_ctx* c, float* p, float v, int64_t n) {
  if (c->dry || n == 0) return;
  fill_kernel<<<GRID1D(n, 256), 256, 0, c->stream>>>(p, v, n); SPA_LAUNCH_CHECK(c);
}
void k_zero(spa3d_ctx* c, void* p, int64_t bytes) {
  if (c->dry || bytes == 0) return;
  hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, c->stream);
  if (e != hipSuccess && !c->hip_err) { c->hip_err = (int)e; c->err = std::string("hipMemsetAsync: ") + hipGetErrorString(e); }
}
template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ s, T* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) st(d + i, s[i]);
}
template <typename T> void k_cast_from_f32(spa3d_ctx* c, const float* s, T* d, int64_t n) {
  if (c->dry || n == 0) return;
  cast_from_f32_kernel<T><<<GRID1D(n, 256), 256, 0, c->stream>>>(s, d, n); SPA_LAUNCH_CHECK(c);
}
template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ s, float* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = ld(s + i);
}
template <typename T> void k_cast_to_f32(spa3d_ctx* c, const T* s, float* d, int64_t n) {
  if (c->dry || n == 0) return;
  cast_to_f32_kernel<T><<<GRID1D(n, 256), 256, 0, c->stream>>>(s, d, n); SPA_LAUNCH_CHECK(c);
}

// weight shadows: src f32 [rows][cols] (row stride src_ld) -> native T [rows][ldn-strided], transposed T [cols][ldt-strided]
template <typename T>
__global__ void pack_kernel(const float* __restrict__ src, int64_t src_ld, int rows, int cols, T* dn, int64_t ldn, T* dt, int64_t ldt) {
  __shared__ float tile[32][33];
  int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    int r = r0 + i, cc = c0 + tx;
    float v = (r < rows && cc < cols) ? src[(int64_t)r * src_ld + cc] : 0.f;
    tile[i][tx] = v;
    if (dn && r < rows && cc < cols) st(dn + (int64_t)r * ldn + cc, v);
  }
  __syncthreads();
  if (dt)
    for (int i = ty; i < 32; i += 8) {
      int cc = c0 + i, r = r0 + tx;
      if (r < rows && cc < cols) st(dt + (int64_t)cc * ldt + r, tile[tx][i]);
    }
}
template <typename T>
void k_pack(spa3d_ctx* c, const float* src, int64_t src_ld, int rows, int cols, T* dn, int64_t ldn, T* dt, int64_t ldt) {
  if (c->dry) return;
  pack_kernel<T><<<dim3((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32)), 256, 0, c->stream>>>(src, src_ld, rows, cols, dn, ldn, dt, ldt);
  SPA_LAUNCH_CHECK(c);
}

template <typename T>
__global__ void transpose_kernel(const T* __restrict__ src, int rows, int cols, T* __restrict__ dst) {
  __shared__ T tile[32][33];
  int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = src[(int64_t)(r0 + i) * cols + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8) if (c0 + i < cols && r0 + tx < rows) dst[(int64_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}
template <typename T> void k_transpose(spa3d_ctx* c, const T* src, int rows, int cols, T* dst) {
  if (c->dry) return;
  transpose_kernel<T><<<dim3((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32)), 256, 0, c->stream>>>(src, rows, cols, dst);
  SPA_LAUNCH_CHECK(c);
}

// ---------------------------------------------------------------------------------------------
// token bookkeeping
// ---------------------------------------------------------------------------------------------
// tok[seq][0][:] = readout param (track_autoencoder_3d.py:161-165)
template <typename T>
__global__ void set_readout_kernel(T* tok, const float* __restrict__ ro, int64_t nseq, int S, int d) {
  const int64_t n = nseq * d;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int64_t s = i / d; int j = (int)(i - s * d);
    st(tok + s * S * d + j, ro[j]);
  }
}
template <typename T> void k_set_readout_rows(spa3d_ctx* c, T* tok, const float* readout, int64_t nseq, int S, int d) {
  if (c->dry || nseq == 0) return;
  set_readout_kernel<T><<<GRID1D(nseq * d, 256), 256, 0, c->stream>>>(tok, readout, nseq, S, d); SPA_LAUNCH_CHECK(c);
}
// One-pass embedding (model.hip encode_chunk): row maps of the token rows the encoder keeps.  Row j of the (compact or dense) token buffer is dense
// token q = row_src[j] (or j) = (seq, s): a frame token (s >= 1) reads input row seq * T + s - 1 and is written to row j; the readout token (s == 0)
// has no input -- its GEMM row is dropped (crow = -1, arow = any valid row) and row j receives the readout parameter here (3d:161-165).
template <typename T>
__global__ void embed_maps_kernel(const int32_t* __restrict__ row_src, int64_t rows, int S, int T_, int32_t* __restrict__ arow, int32_t* __restrict__ crow,
                                  T* __restrict__ tok, const float* __restrict__ ro, int d) {
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < rows; j += (int64_t)gridDim.x * 256) {
    const int64_t q = row_src ? row_src[j] : j;
    const int64_t seq = q / S; const int s_ = (int)(q - seq * S);
    if (s_ == 0) {
      arow[j] = (int32_t)(seq * T_); crow[j] = -1;
      for (int k = 0; k < d; ++k) st(tok + j * d + k, ro[k]);
    } else { arow[j] = (int32_t)(seq * T_ + s_ - 1); crow[j] = (int32_t)j; }
  }
}
template <typename T>
void k_embed_maps(spa3d_ctx* c, const int32_t* row_src, int64_t rows, int S, int T_, int32_t* arow, int32_t* crow, T* tok, const float* readout, int d) {
  if (c->dry || rows == 0) return;
  embed_maps_kernel<T><<<GRID1D(rows, 256), 256, 0, c->stream>>>(row_src, rows, S, T_, arow, crow, tok, readout, d); SPA_LAUNCH_CHECK(c);
}
__global__ void sum3_kernel(const float* a, const float* b, const float* c3, float* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (a ? a[i] : 0.f) + (b ? b[i] : 0.f) + (c3 ? c3[i] : 0.f);
}
void k_sum3(spa3d_ctx* c, const float* a, const float* b, const float* c3, float* out, int n) {
  if (c->dry) return;
  sum3_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(a, b, c3, out, n); SPA_LAUNCH_CHECK(c);
}
// key mask (repairs R2/R3): km[seq][0]=1 ; km[seq][1+t] = visible[seq][t]!=0 && t < boundary[b]
__global__ void keymask_kernel(const float* __restrict__ vis, const int32_t* __restrict__ boundary, int64_t nseq, int N, int T_, float* km) {
  const int S = T_ + 1;
  const int64_t n = nseq * S;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int64_t s = i / S; int k = (int)(i - s * S);
    float v = 1.f;
    if (k > 0) { int t = k - 1; v = (vis[s * T_ + t] != 0.f && t < boundary[s / N]) ? 1.f : 0.f; }
    km[i] = v;
  }
}
void k_keymask(spa3d_ctx* c, const float* visible, const int32_t* boundary, int64_t nseq, int N, int T_, float* km) {
  if (c->dry || nseq == 0) return;
  keymask_kernel<<<GRID1D(nseq * (T_ + 1), 256), 256, 0, c->stream>>>(visible, boundary, nseq, N, T_, km); SPA_LAUNCH_CHECK(c);
}
// dst[i][:] = src[i*stride_rows][:]
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ src, int64_t srows, T* __restrict__ dst, int64_t n, int d) {
  const int64_t tot = n * d;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    int64_t r = i / d; int j = (int)(i - r * d);
    dst[i] = src[r * srows * d + j];
  }
}
template <typename T> void k_gather_rows(spa3d_ctx* c, const T* src, int64_t srows, T* dst, int64_t n, int d) {
  if (c->dry || n == 0) return;
  gather_rows_kernel<T><<<GRID1D(n * d, 256), 256, 0, c->stream>>>(src, srows, dst, n, d); SPA_LAUNCH_CHECK(c);
}
template <typename T>
__global__ void scatter_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t drows, int64_t n, int d) {
  const int64_t tot = n * d;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    int64_t r = i / d; int j = (int)(i - r * d);
    dst[r * drows * d + j] = src[i];
  }
}
template <typename T> void k_scatter_rows(spa3d_ctx* c, const T* src, T* dst, int64_t drows, int64_t n, int d) {
  if (c->dry || n == 0) return;
  scatter_rows_kernel<T><<<GRID1D(n * d, 256), 256, 0, c->stream>>>(src, dst, drows, n, d); SPA_LAUNCH_CHECK(c);
}
// ---------------------------------------------------------------------------------------------
// Token pruning of the track encoder (3DSPA model, 16-bit fused path).  A frame token whose key is masked (occluded, or at / past
// boundary_frame: track_autoencoder_3d.py:167-184) is never attended to, and only token 0 leaves the encoder (:187-188), so a masked
// token's own row influences nothing: the stack runs on COMPACTED, ragged sequences (token 0 + the visible frames, in time order).
//   seq_off [nseq + 1]  first compact row of every sequence (exclusive scan of the kept counts; seq_off[nseq] = kept rows in all)
//   row_src [kept]      dense row (seq * S + t) behind every compact row
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prune_count_kernel(const float* __restrict__ km, int64_t nseq, int S, int32_t* __restrict__ cnt) {
  const int lane = threadIdx.x & 63;
  for (int64_t seq = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); seq < nseq; seq += (int64_t)gridDim.x * 4) {
    int n = 0;
    for (int t = lane; t < S; t += 64) n += km[seq * S + t] != 0.f ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if (lane == 0) cnt[seq] = n;
  }
}
// exclusive scan of cnt[0..n) into off[0..n] by ONE workgroup (n <= a few hundred thousand sequences): per-thread chunk sums, a block
// scan of the 1024 partials, then the chunk is re-walked.  off may alias cnt only if they are the same buffer shifted: they are not.
__global__ __launch_bounds__(1024) void prune_scan_kernel(const int32_t* __restrict__ cnt, int64_t n, int32_t* __restrict__ off) {
  __shared__ int32_t part[1024];
  const int tid = threadIdx.x;
  const int64_t per = (n + 1023) / 1024, a = tid * per, b = a + per < n ? a + per : n;
  int32_t s = 0;
  for (int64_t i = a; i < b; ++i) s += cnt[i];
  part[tid] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int32_t v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int32_t run = tid ? part[tid - 1] : 0;
  for (int64_t i = a; i < b; ++i) { off[i] = run; run += cnt[i]; }
  if (tid == 1023) off[n] = part[1023];
}
__global__ __launch_bounds__(256) void prune_fill_kernel(const float* __restrict__ km, int64_t nseq, int S, const int32_t* __restrict__ off,
                                                         int32_t* __restrict__ row_src) {
  const int lane = threadIdx.x & 63;
  for (int64_t seq = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); seq < nseq; seq += (int64_t)gridDim.x * 4) {
    int base = off[seq];
    for (int t0 = 0; t0 < S; t0 += 64) {
      const int t = t0 + lane;
      const bool keep = t < S && km[seq * S + t] != 0.f;
      const unsigned long long m = __ballot(keep);
      if (keep) row_src[base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)(seq * S + t);
      base += __popcll(m);
    }
  }
}
// returns the number of kept rows (ONE stream synchronisation: the row count sizes every launch that follows); dense count when dry
int64_t k_prune_plan(spa3d_ctx* c, const float* km, int64_t nseq, int S, int32_t* cnt, int32_t* seq_off, int32_t* row_src) {
  if (c->dry) return nseq * S;
  prune_count_kernel<<<(unsigned)std::min<int64_t>(cdiv(nseq, 4), 8192), 256, 0, c->stream>>>(km, nseq, S, cnt); SPA_LAUNCH_CHECK(c);
  prune_scan_kernel<<<1, 1024, 0, c->stream>>>(cnt, nseq, seq_off); SPA_LAUNCH_CHECK(c);
  prune_fill_kernel<<<(unsigned)std::min<int64_t>(cdiv(nseq, 4), 8192), 256, 0, c->stream>>>(km, nseq, S, seq_off, row_src); SPA_LAUNCH_CHECK(c);
  int32_t kept = 0;
  if (hipMemcpyAsync(&kept, seq_off + nseq, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess) {
    if (!c->hip_err) { c->hip_err = -4; c->err = "prune plan: reading the kept-row count failed"; }
    return nseq * S;
  }
  return kept;
}
// rows by index, 16 bytes per thread: MODE 0 dst[i] = src[idx[i]] (gather), 1 dst[idx[i]] = src[i] (scatter), 2 dst[idx[i]] += src[i]
template <typename T, int MODE>
__global__ void rows_idx_kernel(const T* __restrict__ src, const int32_t* __restrict__ idx, T* __restrict__ dst, int64_t n, int d) {
  constexpr int NV = VecOf<T>::N;
  const int cpr = d / NV;  // 16-byte chunks per row (host: d % NV == 0)
  const int64_t tot = n * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cpr; const int ch = (int)(i - r * cpr);
    const int64_t other = idx[r];
    const int64_t so = (MODE == 0 ? other : r) * d + ch * NV, dofs = (MODE == 0 ? r : other) * d + ch * NV;
    if constexpr (MODE == 2) {
      float a[NV], b[NV];
      load_vec<T, NV>(src + so, a); load_vec<T, NV>(dst + dofs, b);
#pragma unroll
      for (int j = 0; j < NV; ++j) b[j] += a[j];
      store_vec<T, NV>(dst + dofs, b);
    } else {
      *(uint4*)(dst + dofs) = *(const uint4*)(src + so);
    }
  }
}
template <typename T> void k_rows_idx(spa3d_ctx* c, int mode, const T* src, const int32_t* idx, T* dst, int64_t n, int d) {
  if (c->dry || n == 0) return;
  constexpr int NV = VecOf<T>::N;
  if (d % NV) { if (!c->hip_err) { c->hip_err = -5; c->err = "rows_idx: row width must be a multiple of 16 bytes"; } return; }
  const dim3 g = GRID1D(n * (d / NV), 256);
  if (mode == 0) rows_idx_kernel<T, 0><<<g, 256, 0, c->stream>>>(src, idx, dst, n, d);
  else if (mode == 1) rows_idx_kernel<T, 1><<<g, 256, 0, c->stream>>>(src, idx, dst, n, d);
  else rows_idx_kernel<T, 2><<<g, 256, 0, c->stream>>>(src, idx, dst, n, d);
  SPA_LAUNCH_CHECK(c);
}

// ---------------------------------------------------------------------------------------------
// Shared latent rows of the readout stack's first block (track_autoencoder_3d.py:276-285, 235-246): token n >= 1 of the sequence of
// query (b, q) is [lat[b][n] | lat[b][n][5 t_q : 5 t_q + 128]] -- a function of (b, n, t_q) only, so every query of a sample with the same
// frame t_q carries the same 128 latent rows into the block's LayerNorm and QKV projection.  A "slot" is a distinct (sample, frame)
// pair; the block runs LN1 / QKV (and their backward) once per slot and expands / reduces through the slot index (model.hip, Share).
// ---------------------------------------------------------------------------------------------
// per sample: slot_local[q] = rank of q's frame among the distinct frames of the sample (order of first occurrence); nslot_b[b] = their number
__global__ __launch_bounds__(256) void share_plan_kernel(const int32_t* __restrict__ qframe, int Q, int32_t* __restrict__ slot, int32_t* __restrict__ nslot_b,
                                                         int32_t* __restrict__ first_q) {
  extern __shared__ int32_t sh[];  // [Q] first occurrence of q's frame, then its rank
  const int b = blockIdx.x;
  const int32_t* fr = qframe + (int64_t)b * Q;
  for (int q = threadIdx.x; q < Q; q += 256) {
    const int32_t f = fr[q];
    int first = q;
    for (int p = 0; p < q; ++p) if (fr[p] == f) { first = p; break; }
    sh[q] = first;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int n = 0;
    for (int q = 0; q < Q; ++q) {
      if (sh[q] == q) { first_q[(int64_t)b * Q + n] = q; sh[q] = -(n + 1); ++n; }  // firsts carry -(rank + 1)
    }
    nslot_b[b] = n;
  }
  __syncthreads();
  for (int q = threadIdx.x; q < Q; q += 256) {
    const int32_t v = sh[q];
    slot[(int64_t)b * Q + q] = v < 0 ? -v - 1 : -sh[v] - 1;  // local rank; made global by share_plan_finish_kernel
  }
}
// prefix over samples: slot -> global slot index; slot_b / slot_f / slot_q0: sample, frame and first member sequence of every slot; total[0] = number of slots
__global__ __launch_bounds__(256) void share_plan_finish_kernel(const int32_t* __restrict__ qframe, int B, int Q, int32_t* __restrict__ slot,
                                                                const int32_t* __restrict__ nslot_b, const int32_t* __restrict__ first_q,
                                                                int32_t* __restrict__ slot_b, int32_t* __restrict__ slot_f, int32_t* __restrict__ slot_q0,
                                                                int32_t* __restrict__ total) {
  __shared__ int32_t off[1025];
  if (threadIdx.x == 0) { int a = 0; for (int b = 0; b < B; ++b) { off[b] = a; a += nslot_b[b]; } off[B] = a; total[0] = a; }
  __syncthreads();
  for (int64_t i = threadIdx.x; i < (int64_t)B * Q; i += 256) {
    const int b = (int)(i / Q), j = (int)(i - (int64_t)b * Q);
    slot[i] += off[b];
    if (j < nslot_b[b]) { slot_b[off[b] + j] = b; slot_f[off[b] + j] = qframe[(int64_t)b * Q + first_q[i]]; slot_q0[off[b] + j] = b * Q + first_q[i]; }
  }
}
// returns the number of slots (ONE stream synchronisation); B * Q (every query its own slot) when dry
int64_t k_share_plan(spa3d_ctx* c, const int32_t* qframe, int64_t B, int Q, int32_t* slot, int32_t* slot_b, int32_t* slot_f, int32_t* slot_q0,
                     int32_t* scratch /*[B*Q + B + 1]*/) {
  if (c->dry) return B * Q;
  if (B > 1024 || Q > 12288) return B * Q;  // outside the plan kernels' LDS tables: every query its own slot, i.e. the caller keeps the dense path
  int32_t* first_q = scratch; int32_t* nslot_b = scratch + B * Q; int32_t* total = nslot_b + B;
  share_plan_kernel<<<(unsigned)B, 256, Q * sizeof(int32_t), c->stream>>>(qframe, Q, slot, nslot_b, first_q); SPA_LAUNCH_CHECK(c);
  share_plan_finish_kernel<<<1, 256, 0, c->stream>>>(qframe, (int)B, Q, slot, nslot_b, first_q, slot_b, slot_f, slot_q0, total); SPA_LAUNCH_CHECK(c);
  int32_t n = 0;
  if (hipMemcpyAsync(&n, total, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
    if (!c->hip_err) { c->hip_err = -4; c->err = "share plan: reading the slot count failed"; }
    return B * Q;
  }
  return n;
}
// xU = [nslot * L latent rows | B*Q query-token rows]: the distinct rows of the readout sequences (assemble_vec_kernel's values)
template <typename T>
__global__ void share_assemble_kernel(const T* __restrict__ qtok, const T* __restrict__ lat, const int32_t* __restrict__ slot_b,
                                      const int32_t* __restrict__ slot_f, int64_t nslot, int64_t BQ, int L, int Cl, int D, T* __restrict__ xU) {
  constexpr int NV = VecOf<T>::N;
  const int cpr = D / NV;
  const int64_t nlat = nslot * L, tot = (nlat + BQ) * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / cpr; const int j = (int)(i - row * cpr) * NV;
    float v[NV];
    if (row >= nlat) load_vec<T, NV>(qtok + (row - nlat) * D + j, v);
    else {
      const int64_t s_ = row / L; const int n = (int)(row - s_ * L);
      const T* lr = lat + ((int64_t)slot_b[s_] * L + n) * Cl;
      if (j < Cl) load_vec<T, NV>(lr + j, v);
      else {
        const int base = j - Cl + 5 * slot_f[s_];
#pragma unroll
        for (int e = 0; e < NV; ++e) { const int cc = base + e; v[e] = (cc >= 0 && cc < Cl) ? ld(lr + cc) : 0.f; }
      }
    }
    store_vec<T, NV>(xU + row * D + j, v);
  }
}
template <typename T>
void k_share_assemble(spa3d_ctx* c, const T* qtok, const T* lat, const int32_t* slot_b, const int32_t* slot_f, int64_t nslot, int64_t BQ, int L, int Cl,
                      int D, T* xU) {
  if (c->dry || BQ == 0) return;
  constexpr int NV = VecOf<T>::N;
  if (D % NV || Cl % NV) { if (!c->hip_err) { c->hip_err = -5; c->err = "share assemble: widths must be multiples of 16 bytes"; } return; }
  share_assemble_kernel<T><<<GRID1D((nslot * L + BQ) * (D / NV), 256), 256, 0, c->stream>>>(qtok, lat, slot_b, slot_f, nslot, BQ, L, Cl, D, xU);
  SPA_LAUNCH_CHECK(c);
}
// dense rows from slot rows.  Forward (add == nullptr): dst[(seq, tkn)] = src[tkn == 0 ? nslot*L + seq : slot[seq]*L + tkn - 1], a gather.
// Backward (add != nullptr): the slot row holds the SUM over the slot's member sequences (the LayerNorm backward is linear in its incoming
// gradient), so it is added to ONE member, the slot's first sequence slot_q0 -- every consumer downstream sums over the queries of a
// sample with the members' common frame (k_assemble_readout_bwd) --, and dst = add elsewhere.
template <typename T, bool ADD>
__global__ void share_expand_kernel(const T* __restrict__ src, const int32_t* __restrict__ slot, const int32_t* __restrict__ slot_q0, int64_t nslot,
                                    int64_t nseq, int S, int d, const T* add, T* dst) {
  constexpr int NV = VecOf<T>::N;
  const int cpr = d / NV, L = S - 1;
  const int64_t tot = nseq * S * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / cpr; const int ch = (int)(i - row * cpr);
    const int64_t seq = row / S; const int tkn = (int)(row - seq * S);
    const int32_t sl = slot[seq];
    const int64_t sr = tkn == 0 ? nslot * L + seq : (int64_t)sl * L + tkn - 1;
    if constexpr (ADD) {
      float b[NV];
      load_vec<T, NV>(add + row * d + ch * NV, b);
      if (tkn == 0 || slot_q0[sl] == (int32_t)seq) {
        float a[NV];
        load_vec<T, NV>(src + sr * d + ch * NV, a);
#pragma unroll
        for (int j = 0; j < NV; ++j) b[j] += a[j];
      }
      store_vec<T, NV>(dst + row * d + ch * NV, b);
    } else {
      *(uint4*)(dst + row * d + ch * NV) = *(const uint4*)(src + sr * d + ch * NV);
    }
  }
}
template <typename T>
void k_share_expand(spa3d_ctx* c, const T* src, const int32_t* slot, const int32_t* slot_q0, int64_t nslot, int64_t nseq, int S, int d, const T* add, T* dst) {
  if (c->dry || nseq == 0) return;
  constexpr int NV = VecOf<T>::N;
  if (d % NV) { if (!c->hip_err) { c->hip_err = -5; c->err = "share expand: row width must be a multiple of 16 bytes"; } return; }
  const dim3 g = GRID1D(nseq * S * (d / NV), 256);
  if (add) share_expand_kernel<T, true><<<g, 256, 0, c->stream>>>(src, slot, slot_q0, nslot, nseq, S, d, add, dst);
  else share_expand_kernel<T, false><<<g, 256, 0, c->stream>>>(src, slot, slot_q0, nslot, nseq, S, d, nullptr, dst);
  SPA_LAUNCH_CHECK(c);
}
// slot rows from dense rows (the transpose of the expansion): dstU[(s, n)] = sum over the sequences of slot s of src[(seq, 1 + n)] (fp32 sums);
// dstU[nslot*L + seq] = src[(seq, 0)].  One workgroup per slot: the member list is built once in LDS, then rows are walked 16 B per thread.
template <typename T>
__global__ __launch_bounds__(256) void share_reduce_kernel(const T* __restrict__ src, const int32_t* __restrict__ slot, const int32_t* __restrict__ slot_b,
                                                           int64_t nslot, int Q, int S, int d, T* __restrict__ dstU) {
  constexpr int NV = VecOf<T>::N;
  extern __shared__ int32_t members[];  // [Q]
  __shared__ int32_t wcnt[4], nmem;
  const int64_t s_ = blockIdx.x; const int L = S - 1, cpr = d / NV;
  const int b = slot_b[s_];
  // member list in ascending query order (ballot + prefix, as prune_fill_kernel): the fp32 sums below then have ONE order, run to run
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int base = 0;
  for (int q0 = 0; q0 < Q; q0 += 256) {
    const int q = q0 + threadIdx.x;
    const bool mine = q < Q && slot[(int64_t)b * Q + q] == (int32_t)s_;
    const unsigned long long m = __ballot(mine);
    if (lane == 0) wcnt[wv] = __popcll(m);
    __syncthreads();
    int pre = base;
    for (int w2 = 0; w2 < wv; ++w2) pre += wcnt[w2];
    if (mine) members[pre + __popcll(m & ((1ull << lane) - 1ull))] = q;
    base += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) nmem = base;
  __syncthreads();
  const int nm = nmem;
  for (int i = threadIdx.x; i < L * cpr; i += 256) {
    const int n = i / cpr, ch = i - n * cpr;
    float acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] = 0.f;
    for (int m = 0; m < nm; ++m) {
      float v[NV];
      load_vec<T, NV>(src + (((int64_t)b * Q + members[m]) * S + 1 + n) * d + ch * NV, v);
#pragma unroll
      for (int j = 0; j < NV; ++j) acc[j] += v[j];
    }
    store_vec<T, NV>(dstU + (s_ * L + n) * d + ch * NV, acc);
  }
}
template <typename T>
void k_share_reduce(spa3d_ctx* c, const T* src, const int32_t* slot, const int32_t* slot_b, int64_t nslot, int64_t nseq, int Q, int S, int d, T* dstU) {
  if (c->dry || nseq == 0) return;
  constexpr int NV = VecOf<T>::N;
  if (d % NV) { if (!c->hip_err) { c->hip_err = -5; c->err = "share reduce: row width must be a multiple of 16 bytes"; } return; }
  if (nslot > 0) share_reduce_kernel<T><<<(unsigned)nslot, 256, Q * sizeof(int32_t), c->stream>>>(src, slot, slot_b, nslot, Q, S, d, dstU);
  SPA_LAUNCH_CHECK(c);
  k_gather_rows<T>(c, src, S, dstU + nslot * (S - 1) * d, nseq, d);  // rows 0: one per sequence
}

// copy rows 1..S-1 of each sequence into a compact [nseq*(S-1)][d] buffer (drop the readout row)
template <typename T>
__global__ void compact_tokens_kernel(const T* __restrict__ tok, T* __restrict__ dst, int64_t nseq, int S, int d) {
  const int64_t tot = nseq * (S - 1) * d;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    int64_t r = i / d; int j = (int)(i - r * d);
    int64_t s = r / (S - 1); int t = (int)(r - s * (S - 1));
    dst[i] = tok[(s * S + 1 + t) * d + j];
  }
}
template <typename T> void k_compact_tokens(spa3d_ctx* c, const T* tok, T* dst, int64_t nseq, int S, int d) {
  if (c->dry || nseq == 0) return;
  compact_tokens_kernel<T><<<GRID1D(nseq * (S - 1) * d, 256), 256, 0, c->stream>>>(tok, dst, nseq, S, d); SPA_LAUNCH_CHECK(c);
}
// ParamStateInit broadcast (track_autoencoder.py:41-53) and its gradient
template <typename T>
__global__ void bcast_rows_kernel(const float* __restrict__ src, int64_t per, T* __restrict__ dst, int64_t B) {
  const int64_t tot = per * B;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) st(dst + i, src[i % per]);
}
template <typename T> void k_broadcast_rows(spa3d_ctx* c, const float* src, int rows, int d, T* dst, int64_t B) {
  if (c->dry || B == 0) return;
  bcast_rows_kernel<T><<<GRID1D((int64_t)rows * d * B, 256), 256, 0, c->stream>>>(src, (int64_t)rows * d, dst, B); SPA_LAUNCH_CHECK(c);
}
template <typename T>
__global__ void bcast_grad_kernel(const T* __restrict__ dsrc, int64_t per, int64_t B, int64_t bstride, float* __restrict__ dparam, int64_t bchunk) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= per) return;
  const int64_t b0 = (int64_t)blockIdx.y * bchunk; int64_t b1 = b0 + bchunk; if (b1 > B) b1 = B;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four independent chains: the strided rows are latency, not bandwidth
  int64_t b = b0;
  for (; b + 3 < b1; b += 4) {
    s0 += ld(dsrc + b * bstride + i); s1 += ld(dsrc + (b + 1) * bstride + i);
    s2 += ld(dsrc + (b + 2) * bstride + i); s3 += ld(dsrc + (b + 3) * bstride + i);
  }
  for (; b < b1; ++b) s0 += ld(dsrc + b * bstride + i);
  const float s = (s0 + s1) + (s2 + s3);
  if (gridDim.y == 1) dparam[i] += s; else grad_add(dparam + i, s);
}
// dparam[per] += sum_b dsrc[b*bstride + :per]   (B up to ~10^5 strided rows: split over blockIdx.y, one f32 atomic per column per slice)
template <typename T> void k_bcast_grad(spa3d_ctx* c, const T* dsrc, int64_t per, int64_t B, int64_t bstride, float* dparam) {
  if (c->dry || B == 0) return;
  const int64_t gx = cdiv(per, 256);
  int64_t gy = std::max<int64_t>(1, std::min<int64_t>(B / 32, std::max<int64_t>(1, 2048 / gx)));
  const int64_t bchunk = cdiv(B, gy); gy = cdiv(B, bchunk);
  bcast_grad_kernel<T><<<dim3((unsigned)gx, (unsigned)gy), 256, 0, c->stream>>>(dsrc, per, B, bstride, dparam, bchunk); SPA_LAUNCH_CHECK(c);
}

// ---------------------------------------------------------------------------------------------
// Dense layers with a tiny input width (K <= 4: the 1-channel depth feature, 3d:143-147).  As a GEMM this is a rank-K update of a
// 3.4 M x 384 tensor: pure streaming.  out[crow(m)][:] += x[m][:K] . w[K][N] + bias, crow(m) = m + (m / G + 1) * S (token rows
// 1..T of each sequence).  8 columns (16 B) per thread.
// ---------------------------------------------------------------------------------------------
template <typename T, int K>
__global__ void rank_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ out,
                                int64_t M, int N, int64_t ldo, int rgroup, int rskip) {
  constexpr int NV = 8;
  const int cpr = N / NV;                   // column groups per row
  const int slots = 256 / cpr;              // rows per block iteration (5 at N = 384)
  const int slot = threadIdx.x / cpr, c = (threadIdx.x - slot * cpr) * NV;
  if (slot >= slots) return;
  float wv[K][NV], bvv[NV];                 // this thread's weight / bias columns: loaded once
#pragma unroll
  for (int j = 0; j < NV; ++j) { bvv[j] = bias ? bias[c + j] : 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) wv[k][j] = ld(w + (int64_t)k * N + c + j); }
  // four rows per iteration, all loads first: one 16-B load in flight per thread is latency-bound (1.5 TB/s)
  constexpr int U = 4;
  const int64_t stride = (int64_t)gridDim.x * slots;
  for (int64_t m0 = (int64_t)blockIdx.x * slots + slot; m0 < M; m0 += stride * U) {
    float cur[U][NV]; float xv[U][K]; T* o[U]; bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t m = m0 + u * stride; ok[u] = m < M;
      const int64_t mm = ok[u] ? m : m0;
      const int64_t crow = rgroup > 0 ? mm + ((unsigned)mm / (unsigned)rgroup + 1) * (int64_t)rskip : mm;  // M < 2^31 (host)
      o[u] = out + crow * ldo + c;
      load_vec<T, NV>(o[u], cur[u]);
#pragma unroll
      for (int k = 0; k < K; ++k) xv[u][k] = ld(x + mm * K + k);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) acc = fmaf(xv[u][k], wv[k][j], acc);
        cur[u][j] += acc + bvv[j];
      }
      if (ok[u]) store_vec<T, NV>(o[u], cur[u]);
    }
  }
}
// gw[K][N] += x^T dY(rows remapped), gb[N] += colsum(dY): one pass over dY, per-block partials in registers, f32 atomics at the end
template <typename T, int K>
__global__ void rank_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, int64_t M, int N, int64_t ldy, int rgroup, int rskip,
                                float* __restrict__ gw, float* __restrict__ gb, int64_t rows_per_block) {
  constexpr int NV = 8;
  const int cpr = N / NV;                       // column groups per row
  const int slots = 256 / cpr;                  // rows processed per block iteration
  const int slot = threadIdx.x / cpr, c = (threadIdx.x - slot * cpr) * NV;
  const int64_t m0 = (int64_t)blockIdx.x * rows_per_block; int64_t m1 = m0 + rows_per_block; if (m1 > M) m1 = M;
  if (slot >= slots) m1 = m0;  // idle threads: no rows, but they take part in the barriers below
  float aw[K][NV], ab[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) { ab[j] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) aw[k][j] = 0.f; }
  constexpr int U = 4;  // four rows per iteration, loads first
  for (int64_t mb = m0 + slot; mb < m1; mb += (int64_t)slots * U) {
    float d[U][NV]; float xv[U][K];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t m = mb + (int64_t)u * slots;
      if (m < m1) {
        const int64_t row = rgroup > 0 ? m + ((unsigned)m / (unsigned)rgroup + 1) * (int64_t)rskip : m;
        load_vec<T, NV>(dy + row * ldy + c, d[u]);
#pragma unroll
        for (int k = 0; k < K; ++k) xv[u][k] = ld(x + m * K + k);
      } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) d[u][j] = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) xv[u][k] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int j = 0; j < NV; ++j) ab[j] += d[u][j];
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < NV; ++j) aw[k][j] = fmaf(xv[u][k], d[u][j], aw[k][j]);
    }
  }
  // block-level reduction over the row slots in LDS, then ONE global atomic per column per block (all blocks hit the same 2 N addresses)
  __shared__ float red[(K + 1) * 2048];  // (K + 1) x N floats, N <= 2048 (N / 8 <= 256 column groups)
  float* rw = red; float* rb = red + (int64_t)K * N;
  __syncthreads();
  for (int t = threadIdx.x; t < (K + 1) * N; t += 256) red[t] = 0.f;
  __syncthreads();
  if (slot < slots) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      atomicAdd(rb + c + j, ab[j]);
#pragma unroll
      for (int k = 0; k < K; ++k) atomicAdd(rw + k * N + c + j, aw[k][j]);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < N; t += 256) {
    if (gb) grad_add(gb + t, rb[t]);
#pragma unroll
    for (int k = 0; k < K; ++k) grad_add(gw + (int64_t)k * N + t, rw[k * N + t]);
  }
}
template <typename T>
bool k_rank_fwd(spa3d_ctx* c, const T* x, const T* w, const float* bias, T* out, int64_t M, int N, int K, int64_t ldo, int rgroup, int rskip) {
  if (K < 1 || K > 4 || N % 8 || N / 8 > 256 || ldo % 8 || (((uintptr_t)out) & 15) || M >= 0x7fffffffLL) return false;
  if (c->dry || M == 0) return true;
  const dim3 g = GRID1D(cdiv(M, 4 * (256 / (N / 8))) * 256, 256);
  switch (K) {
    case 1: rank_fwd_kernel<T, 1><<<g, 256, 0, c->stream>>>(x, w, bias, out, M, N, ldo, rgroup, rskip); break;
    case 2: rank_fwd_kernel<T, 2><<<g, 256, 0, c->stream>>>(x, w, bias, out, M, N, ldo, rgroup, rskip); break;
    case 3: rank_fwd_kernel<T, 3><<<g, 256, 0, c->stream>>>(x, w, bias, out, M, N, ldo, rgroup, rskip); break;
    default: rank_fwd_kernel<T, 4><<<g, 256, 0, c->stream>>>(x, w, bias, out, M, N, ldo, rgroup, rskip); break;
  }
  SPA_LAUNCH_CHECK(c);
  return true;
}
template <typename T>
bool k_rank_bwd(spa3d_ctx* c, const T* x, const T* dy, int64_t M, int N, int K, int64_t ldy, int rgroup, int rskip, float* gw, float* gb) {
  if (c->det_grads) return false;  // its workgroup-level reduction uses LDS float atomics (arrival order): the deterministic mode takes the GEMM path
  if (K < 1 || K > 4 || N % 8 || N / 8 > 256 || ldy % 8 || (((uintptr_t)dy) & 15) || M >= 0x7fffffffLL) return false;
  if (c->dry || M == 0) return true;
  const int64_t rpb = std::max<int64_t>(256, cdiv(M, 1024));
  const unsigned g = (unsigned)cdiv(M, rpb);
  switch (K) {
    case 1: rank_bwd_kernel<T, 1><<<g, 256, 0, c->stream>>>(x, dy, M, N, ldy, rgroup, rskip, gw, gb, rpb); break;
    case 2: rank_bwd_kernel<T, 2><<<g, 256, 0, c->stream>>>(x, dy, M, N, ldy, rgroup, rskip, gw, gb, rpb); break;
    case 3: rank_bwd_kernel<T, 3><<<g, 256, 0, c->stream>>>(x, dy, M, N, ldy, rgroup, rskip, gw, gb, rpb); break;
    default: rank_bwd_kernel<T, 4><<<g, 256, 0, c->stream>>>(x, dy, M, N, ldy, rgroup, rskip, gw, gb, rpb); break;
  }
  SPA_LAUNCH_CHECK(c);
  return true;
}

// ---------------------------------------------------------------------------------------------
// D1: clip, discretise, fixed noise, straight-through (track_autoencoder_3d.py:251-260)
// out = l - (l - q)  (same op order as the reference);  clipmask = 1[-1<=raw<=1] for the backward
// ---------------------------------------------------------------------------------------------
__global__ void discretize_kernel(const float* __restrict__ lat, const float* __restrict__ noise, int disc, float* __restrict__ out,
                                  float* __restrict__ clipmask, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float raw = lat[i];
    float l = raw != raw ? raw : fminf(fmaxf(raw, -1.f), 1.f);   // jnp.clip keeps a NaN a NaN (fmaxf / fminf would return the bound): a diverged latent must show
    if (clipmask) clipmask[i] = raw != raw ? raw : ((raw >= -1.f && raw <= 1.f) ? 1.f : 0.f);   // ... in the backward as well (jnp.clip's gradient of NaN is NaN)
    if (disc) {
      float q = rintf(__fmul_rn(l, 128.f)) / 128.f;
      q = __fsub_rn(__fadd_rn(q, noise[i] / 128.f), 1.0f / 256.0f);
      l = __fsub_rn(l, __fsub_rn(l, q));
    }
    out[i] = l;
  }
}
void k_discretize(spa3d_ctx* c, const float* lat, const float* noise, int discretize, float* out, float* clipmask, int64_t n) {
  if (c->dry || n == 0) return;
  discretize_kernel<<<GRID1D(n, 256), 256, 0, c->stream>>>(lat, noise, discretize, out, clipmask, n); SPA_LAUNCH_CHECK(c);
}
__global__ void mul_kernel(float* __restrict__ a, const float* __restrict__ b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a[i] *= b[i];
}
void k_mul(spa3d_ctx* c, float* a, const float* b, int64_t n) {
  if (c->dry || n == 0) return;
  mul_kernel<<<GRID1D(n, 256), 256, 0, c->stream>>>(a, b, n); SPA_LAUNCH_CHECK(c);
}

// ---------------------------------------------------------------------------------------------
// D4-D6: readout sequence assembly without materialising tile/eye (track_autoencoder_3d.py:235-246,276-284)
// seq[b][q][0][:] = qtok[b][q][:] ; seq[b][q][1+n][c<Cl] = lat[b][n][c] ; seq[b][q][1+n][Cl+dd] = lat[b][n][dd+5*t_q] or 0
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void assemble_kernel(const T* __restrict__ qtok, const T* __restrict__ lat, const int32_t* __restrict__ qframe, int64_t BQ, int Q,
                                int L, int Cl, int D, T* __restrict__ seq) {
  const int64_t tot = BQ * (L + 1) * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    int64_t row = i / D; int j = (int)(i - row * D);
    int64_t bq = row / (L + 1); int tkn = (int)(row - bq * (L + 1));
    T v;
    if (tkn == 0) v = qtok[bq * D + j];
    else {
      int64_t b = bq / Q; int n = tkn - 1;
      const T* lr = lat + (b * L + n) * Cl;
      if (j < Cl) v = lr[j];
      else {
        int64_t cc = (int64_t)(j - Cl) + 5 * (int64_t)qframe[bq];
        if (cc >= 0 && cc < Cl) v = lr[cc]; else { T z; st(&z, 0.f); v = z; }
      }
    }
    seq[i] = v;
  }
}
// 8 elements per thread (one 16-B store), 32-bit index math; only the window part (Cl <= j, source shifted by 5 t_q elements, hence
// unaligned) gathers element-wise.  Needs D % 8 == 0, Cl % 8 == 0 and < 2^31 chunks (else the scalar kernel above).
template <typename T>
__global__ void assemble_vec_kernel(const T* __restrict__ qtok, const T* __restrict__ lat, const int32_t* __restrict__ qframe, unsigned nchunk,
                                    int Q, int L, int Cl, int D, T* __restrict__ seq) {
  constexpr int NV = VecOf<T>::N;
  const unsigned cpr = (unsigned)D / NV;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < nchunk; i += gridDim.x * 256u) {
    const unsigned row = i / cpr; const int j = (int)(i - row * cpr) * NV;
    const unsigned bq = row / (unsigned)(L + 1); const int tkn = (int)(row - bq * (unsigned)(L + 1));
    float v[NV];
    if (tkn == 0) load_vec<T, NV>(qtok + (int64_t)bq * D + j, v);
    else {
      const unsigned b = bq / (unsigned)Q;
      const T* lr = lat + ((int64_t)b * L + (tkn - 1)) * Cl;
      if (j < Cl) load_vec<T, NV>(lr + j, v);
      else {
        const int base = j - Cl + 5 * qframe[bq];
#pragma unroll
        for (int e = 0; e < NV; ++e) { const int cc = base + e; v[e] = (cc >= 0 && cc < Cl) ? ld(lr + cc) : 0.f; }
      }
    }
    store_vec<T, NV>(seq + (int64_t)row * D + j, v);
  }
}
template <typename T>
void k_assemble_readout(spa3d_ctx* c, const T* qtok, const T* lat, const int32_t* qframe, int64_t B, int Q, int L, int Cl, int D, T* seq) {
  if (c->dry || B == 0) return;
  constexpr int NV = VecOf<T>::N;
  const int64_t nchunk = B * Q * (L + 1) * (D / NV);
  if (D % NV == 0 && Cl % NV == 0 && nchunk < 0x7fffffffLL && ((((uintptr_t)qtok) | ((uintptr_t)lat) | ((uintptr_t)seq)) & 15) == 0) {
    assemble_vec_kernel<T><<<GRID1D(nchunk, 256), 256, 0, c->stream>>>(qtok, lat, qframe, (unsigned)nchunk, Q, L, Cl, D, seq);
  } else {
    assemble_kernel<T><<<GRID1D(B * Q * (L + 1) * D, 256), 256, 0, c->stream>>>(qtok, lat, qframe, B * Q, Q, L, Cl, D, seq);
  }
  SPA_LAUNCH_CHECK(c);
}
// backward: dqtok = dseq[:, :, 0, :] ; dlat[b][n][c] = sum_q dseq[b][q][1+n][c] + sum_q dseq[b][q][1+n][Cl + c-5t_q] 1[0<=c-5t_q<D-Cl]
template <typename T>
__global__ void assemble_bwd_lat_kernel(const T* __restrict__ dseq, const int32_t* __restrict__ qframe, int64_t B, int Q, int L, int Cl,
                                        int D, float* __restrict__ dlat) {
  const int64_t tot = B * L * Cl;
  const int Wd = D - Cl;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    int64_t bn = i / Cl; int cc = (int)(i - bn * Cl);
    int64_t b = bn / L; int n = (int)(bn - b * L);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // independent chains: 512 dependent strided loads were pure latency
    auto term = [&](int q) {
      const T* r = dseq + (((b * Q + q) * (L + 1)) + 1 + n) * D;
      float t = ld(r + cc);
      const int dd = cc - 5 * qframe[b * Q + q];
      if (dd >= 0 && dd < Wd) t += ld(r + Cl + dd);
      return t;
    };
    int q = 0;
    for (; q + 3 < Q; q += 4) { s0 += term(q); s1 += term(q + 1); s2 += term(q + 2); s3 += term(q + 3); }
    for (; q < Q; ++q) s0 += term(q);
    dlat[i] = (s0 + s1) + (s2 + s3);
  }
}
template <typename T>
void k_assemble_readout_bwd(spa3d_ctx* c, const T* dseq, const int32_t* qframe, int64_t B, int Q, int L, int Cl, int D, T* dqtok,
                            float* dlat) {
  if (c->dry || B == 0) return;
  k_gather_rows<T>(c, dseq, L + 1, dqtok, B * Q, D);
  assemble_bwd_lat_kernel<T><<<GRID1D(B * L * Cl, 256), 256, 0, c->stream>>>(dseq, qframe, B, Q, L, Cl, D, dlat);
  SPA_LAUNCH_CHECK(c);
}

// ---------------------------------------------------------------------------------------------
// D8 head split + compute_loss_3d (track_autoencoder_3d.py:289-301, train.py:96-129)
// head[q][c*T+t], c<3 coords (coordinate-major), c==3 visibility logit
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float log_sigmoid_f(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
// The loss numerators and the visible count are summed over the whole batch by thousands of workgroups.  Float atomics make that sum depend
// on arrival order (the same batch gave 14089.21875 and 14089.216796875 in round 3), so two data-parallel replicas could log different
// losses and a test could not ask for bit-equal reruns.  Each workgroup reduces in a fixed order and adds its partial as a 64-bit FIXED-POINT
// integer (2^-24 units: exact, order-independent integer addition; quantisation 6e-8 per workgroup, far below fp32 resolution of the sums).
// Layout of the 10-float `sums` block: three 64-bit accumulators {position numerator, bce numerator, visible count} | denominator | loss scale | flag word.
// A partial that is not finite or beyond the fixed-point range sets a STICKY flag word (atomicOr) next to the accumulators and adds nothing; the sums then read as
// NaN.  (Round 4 added a marker value to the accumulator itself: k marked workgroups sum to k * 2^62 mod 2^64 = 0 for 4 | k, so a fully diverged forward reported loss 0.)
constexpr float LOSS_FIX = 16777216.f;                 // 2^24
__device__ __forceinline__ void loss_acc_add(unsigned long long* acc, unsigned* poison, float partial) {
  const float f = partial * LOSS_FIX;
  if (fabsf(f) < 1.0e15f) atomicAdd(acc, (unsigned long long)__float2ll_rn(f));   // NaN fails the comparison too
  else atomicOr(poison, 1u);
}
__device__ __forceinline__ float loss_acc_read(const unsigned long long* acc, const unsigned* poison) {
  if (*poison) return __int_as_float(0x7fc00000);
  return (float)((double)(long long)*acc * (1.0 / 16777216.0));
}
__global__ __launch_bounds__(256) void head_loss_fwd_kernel(const float* __restrict__ head, int64_t nq, int T_, const float* __restrict__ tgt,
                                                            const float* __restrict__ tvis, float* __restrict__ tracks,
                                                            float* __restrict__ vlog, float* __restrict__ clog, float* __restrict__ sums, unsigned* poison, int NC) {
  // head row: NC coordinate blocks of T, then the visibility logits; the 2-D model (NC == 2) has a 4th block: certainty logits
  __shared__ float red[3][4];
  float pn = 0.f, bn = 0.f;
  const int64_t n = nq * T_;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int64_t q = i / T_; int t = (int)(i - q * T_);
    const float* hr = head + q * 4 * T_;
    const float lg = hr[NC * T_ + t];
    float perr = 0.f;
    for (int cdx = 0; cdx < NC; ++cdx) {
      const float pv = hr[cdx * T_ + t];
      if (tracks) tracks[i * NC + cdx] = pv;
      if (tgt) perr += fabsf(pv - tgt[i * NC + cdx]);
    }
    if (vlog) vlog[i] = lg;
    if (clog) clog[i] = NC == 2 ? hr[3 * T_ + t] : 0.f;  // 3DSPA: certain_logits = zeros (3d:301); TRAJAN: real head (ta:344)
    if (tgt) {
      float y = tvis[i];
      pn += perr * y;
      bn += -y * log_sigmoid_f(lg) - (1.f - y) * log_sigmoid_f(-lg);
    }
  }
  if (!tgt) return;
  pn = wave_sum(pn); bn = wave_sum(bn);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = pn; red[1][w] = bn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    loss_acc_add((unsigned long long*)sums + 0, poison, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    loss_acc_add((unsigned long long*)sums + 1, poison, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}
void k_loss_fwd(spa3d_ctx* c, const float* head, int64_t nq, int T_, const float* tgt, const float* tvis, float* tracks, float* vlog,
                float* clog, float* sums, unsigned* poison, int NC) {
  if (c->dry || nq == 0) return;
  unsigned g = (unsigned)std::min<int64_t>(cdiv(nq * T_, 256), 4096);
  head_loss_fwd_kernel<<<g, 256, 0, c->stream>>>(head, nq, T_, tgt, tvis, tracks, vlog, clog, sums, poison, NC);
  SPA_LAUNCH_CHECK(c);
}
// same numerators from already-split predictions (spa3d_loss entry point)
__global__ __launch_bounds__(256) void loss_from_preds_kernel(const float* __restrict__ tracks, const float* __restrict__ vlog, int64_t n,
                                                              const float* __restrict__ tgt, const float* __restrict__ tvis, float* sums, unsigned* poison, int NC) {
  __shared__ float red[2][4];
  float pn = 0.f, bn = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float y = tvis[i], lg = vlog[i];
    float perr = 0.f;
    for (int cdx = 0; cdx < NC; ++cdx) perr += fabsf(tracks[i * NC + cdx] - tgt[i * NC + cdx]);
    pn += perr * y;
    bn += -y * log_sigmoid_f(lg) - (1.f - y) * log_sigmoid_f(-lg);
  }
  pn = wave_sum(pn); bn = wave_sum(bn);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = pn; red[1][w] = bn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    loss_acc_add((unsigned long long*)sums + 0, poison, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    loss_acc_add((unsigned long long*)sums + 1, poison, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}
void k_loss_from_preds(spa3d_ctx* c, const float* tracks, const float* vlog, int64_t n, const float* tgt, const float* tvis, float* sums,
                       unsigned* poison, int NC) {
  if (c->dry || n == 0) return;
  unsigned g = (unsigned)std::min<int64_t>(cdiv(n, 256), 4096);
  loss_from_preds_kernel<<<g, 256, 0, c->stream>>>(tracks, vlog, n, tgt, tvis, sums, poison, NC);
  SPA_LAUNCH_CHECK(c);
}
__global__ __launch_bounds__(256) void vis_count_kernel(const float* __restrict__ v, int64_t n, float* out, unsigned* poison) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += v[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss_acc_add((unsigned long long*)out, poison, red[0] + red[1] + red[2] + red[3]);
}
void k_vis_count(spa3d_ctx* c, const float* tvis, int64_t n, float* out, unsigned* poison) {
  if (c->dry || n == 0) return;
  unsigned g = (unsigned)std::min<int64_t>(cdiv(n, 256), 1024);
  vis_count_kernel<<<g, 256, 0, c->stream>>>(tvis, n, out, poison); SPA_LAUNCH_CHECK(c);
}
// sums = {pos_num, bce_num, vis_cnt} (fixed-point accumulators, see loss_acc_add); denom_dev = denom_host>0 ? denom_host : max(vis_cnt,1)
__global__ void set_denom_kernel(const float* sums, const unsigned* poison, float denom_host, float* denom_dev) {
  *denom_dev = denom_host > 0.f ? denom_host : fmaxf(loss_acc_read((const unsigned long long*)sums + 2, poison), 1.f);  // fmaxf(NaN, 1) = 1: the numerators carry the NaN
}
void k_set_denom(spa3d_ctx* c, const float* sums, const unsigned* poison, float denom_host, float* denom_dev) {
  if (c->dry) return;
  set_denom_kernel<<<1, 1, 0, c->stream>>>(sums, poison, denom_host, denom_dev); SPA_LAUNCH_CHECK(c);
}
__global__ void loss_finalize_kernel(const float* sums, const unsigned* poison, const float* denom_dev, float l1w, float bcew, float* loss3) {
  float d = *denom_dev;
  float pos = loss_acc_read((const unsigned long long*)sums + 0, poison) / d, vis = loss_acc_read((const unsigned long long*)sums + 1, poison) / d;
  loss3[0] = l1w * pos + bcew * vis; loss3[1] = pos; loss3[2] = vis;
}
void k_loss_finalize(spa3d_ctx* c, const float* sums, const unsigned* poison, const float* denom_dev, float l1w, float bcew, float* loss3) {
  if (c->dry) return;
  loss_finalize_kernel<<<1, 1, 0, c->stream>>>(sums, poison, denom_dev, l1w, bcew, loss3); SPA_LAUNCH_CHECK(c);
}
// d head: l1w*sign(pred-tgt)*vis/denom ; bcew*(sigmoid(l)-y)/denom ; sign(0)=0
template <typename T>
__global__ void loss_bwd_kernel(const float* __restrict__ head, int64_t nq, int T_, const float* __restrict__ tgt,
                                const float* __restrict__ tvis, const float* __restrict__ denom_dev, float l1w, float bcew, T* __restrict__ dhead,
                                int NC, const float* __restrict__ scale_dev) {
  const float inv = (scale_dev ? *scale_dev : 1.f) / *denom_dev;  // scale_dev: the fp16 mode's loss scale (a power of two)
  const int64_t n = nq * 4 * T_;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int64_t q = i / (4 * T_); int j = (int)(i - q * 4 * T_);
    int cc = j / T_, t = j - cc * T_;
    float y = tvis[q * T_ + t], g;
    if (cc < NC) {
      float df = head[i] - tgt[(q * T_ + t) * NC + cc];
      g = l1w * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * y * inv;
    } else if (cc == NC) {
      float l = head[i];
      g = bcew * (1.f / (1.f + expf(-l)) - y) * inv;
    } else {
      g = 0.f;  // TRAJAN's certainty head carries no loss term in compute_loss_2d (train.py:60-93)
    }
    st(dhead + i, g);
  }
}
template <typename T>
void k_loss_bwd(spa3d_ctx* c, const float* head, int64_t nq, int T_, const float* tgt, const float* tvis, const float* denom_dev, float l1w,
                float bcew, T* dhead, int NC, const float* scale_dev) {
  if (c->dry || nq == 0) return;
  loss_bwd_kernel<T><<<GRID1D(nq * 4 * T_, 256), 256, 0, c->stream>>>(head, nq, T_, tgt, tvis, denom_dev, l1w, bcew, dhead, NC, scale_dev);
  SPA_LAUNCH_CHECK(c);
}
// Loss scale of the 16-bit backward (fp16 mode).  setting > 0: that value.  setting < 0: automatic -- the largest power of two that keeps
// the head gradient's magnitude l1w / denom at or below |setting| (16): activations' gradients then sit mid-range in fp16 for any batch
// size (at BASELINE cfg#3, denom = 4.4 M: 8192; a 100-element toy batch: 1).
// `state` (may be null): the caller's dynamic-scale state, state[0] = a power-of-two multiplier in (0, 1] that spa3d_adamw_step halves after
// a step with a non-finite gradient norm and grows back after 200 finite ones (0 or garbage-free zero memory = 1).
__global__ void set_loss_scale_kernel(const float* __restrict__ denom_dev, float l1w, float setting, const float* __restrict__ state,
                                      float* __restrict__ scale_dev) {
  float s = setting;
  if (setting < 0.f) s = fminf(fmaxf(exp2f(floorf(log2f(*denom_dev * (-setting) / l1w))), 1.f), 16777216.f);
  if (state) { const float m = state[0]; if (m > 0.f && m < 1.f) s = fmaxf(s * m, 5.9604645e-8f); }
  *scale_dev = s;
}
void k_set_loss_scale(spa3d_ctx* c, const float* denom_dev, float l1w, float setting, float* scale_dev) {
  if (c->dry) return;
  set_loss_scale_kernel<<<1, 1, 0, c->stream>>>(denom_dev, l1w, setting, c->loss_scale_state, scale_dev); SPA_LAUNCH_CHECK(c);
}
__global__ void unscale_kernel(float* __restrict__ a, const float* __restrict__ scale_dev, int64_t n) {
  const float s = 1.f / *scale_dev;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a[i] *= s;
}
void k_unscale(spa3d_ctx* c, float* a, const float* scale_dev, int64_t n) {
  if (c->dry || n == 0) return;
  unscale_kernel<<<GRID1D(n, 256), 256, 0, c->stream>>>(a, scale_dev, n); SPA_LAUNCH_CHECK(c);
}

// ---------------------------------------------------------------------------------------------
// optimizer: clip_by_global_norm -> adamw -> apply_updates on flat buffers (train.py:239-242)
// ---------------------------------------------------------------------------------------------
// Global gradient norm, reproducible: every workgroup writes ITS partial sum of squares to scratch[SUMSQ_OFF + block] (fixed thread -> element
// map, fixed reduction tree) and adamw_guard_kernel adds the partials in index order.  A float atomicAdd here made the clip factor -- and so the
// whole AdamW update -- depend on arrival order: two data-parallel replicas holding bit-identical reduced gradients could drift apart.
constexpr int SUMSQ_MAXB = 512, SUMSQ_OFF = 256;  // partials live in floats [256, 768) of the >= 4 KiB scratch block
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float tstep, float clip, float b1,
                                                    float b2, float eps, float wd, float* scratch) {
  // bias correction at the number of updates actually APPLIED: calls so far (tstep = step + 1) minus the steps skipped before this one
  // (scratch[3]; this kernel returns early when this step itself is skipped).  -expm1(t log b) keeps 1 - b^t accurate for b -> 1.
  const float teff = fmaxf(tstep - scratch[3], 1.f);
  const float bc1 = -expm1f(teff * logf(b1)), bc2 = -expm1f(teff * logf(b2));
  const float gn = sqrtf(scratch[1]);
  const float sc = gn < clip ? 1.f : clip / gn;
  if (blockIdx.x == 0 && threadIdx.x == 0) scratch[0] = gn;
  if (!(gn <= 3.0e38f)) return;  // inf / NaN gradient (an fp16 overflow): parameters and both moments stay as they are (adamw_guard_kernel reports it)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float gi = g[i] * sc;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    float mh = mi / bc1, vh = vi / bc2;
    float pi = p[i];
    p[i] = pi - lr * (mh / (sqrtf(vh) + eps) + wd * pi);
  }
}
// scratch[2] := 1 if this step was skipped (non-finite gradient norm) else 0; scratch[3] += skipped steps; scratch[4] = dynamic loss-scale
// multiplier (0 = 1; halved on a skip, doubled up to 1 after 200 finite steps counted in scratch[5]) -- read by set_loss_scale_kernel when the
// caller registered it with spa3d_set_loss_scale_state.
__global__ __launch_bounds__(256) void adamw_guard_kernel(float* scratch, int nblocks) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += scratch[SUMSQ_OFF + i];  // fixed order: thread t takes partials t, t + 256
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x != 0) return;
  scratch[1] = red[0];
  const bool bad = !(sqrtf(scratch[1]) <= 3.0e38f);
  scratch[2] = bad ? 1.f : 0.f;
  // the dynamic loss-scale multiplier must be a power of two in [2^-24, 1]; anything else (uninitialised scratch of a caller written against
  // the old contract, a stray value in (0,1]) reads as 1
  float m = scratch[4]; { int e; if (!(m >= 5.9604645e-8f && m <= 1.f && frexpf(m, &e) == 0.5f)) m = 1.f; }
  if (bad) { scratch[3] += 1.f; m = fmaxf(m * 0.5f, 5.9604645e-8f); scratch[5] = 0.f; }
  else if (m < 1.f) { scratch[5] += 1.f; if (scratch[5] >= 200.f) { m = fminf(2.f * m, 1.f); scratch[5] = 0.f; } }
  scratch[4] = m;
}
void k_adamw(spa3d_ctx* c, float* p, const float* g, float* m, float* v, int64_t n, float lr, int64_t step, float clip, float b1, float b2,
             float eps, float wd, float* scratch) {
  (void)hipMemsetAsync(scratch, 0, 12, c->stream);
  unsigned gr = (unsigned)std::min<int64_t>(cdiv(n, 256), 4096);
  const unsigned gs = (unsigned)std::min<int64_t>(cdiv(n, 256), SUMSQ_MAXB);
  sumsq_kernel<<<gs, 256, 0, c->stream>>>(g, n, scratch + SUMSQ_OFF);
  adamw_guard_kernel<<<1, 256, 0, c->stream>>>(scratch, (int)gs);
  adamw_kernel<<<gr, 256, 0, c->stream>>>(p, g, m, v, n, lr, (float)(step + 1), clip, b1, b2, eps, wd, scratch);
  SPA_LAUNCH_CHECK(c);
}

// ---------------------------------------------------------------------------------------------
// jax.random.uniform(PRNGKey(k0,k1), [n]) legacy threefry layout
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  x0 += ks[0]; x1 += ks[1];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { x0 += x1; x1 = rotl32(x1, R[i & 1][j]); x1 ^= x0; }
    x0 += ks[(i + 1) % 3]; x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
  }
}
__global__ void uniform_noise_kernel(float* __restrict__ out, int64_t n, int64_t half, uint32_t k0, uint32_t k1) {
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < half; j += (int64_t)gridDim.x * 256) {
    uint32_t x0 = (uint32_t)j, x1 = (uint32_t)(half + j);
    threefry2x32(k0, k1, x0, x1);
    out[j] = __uint_as_float((x0 >> 9) | 0x3F800000u) - 1.0f;
    if (half + j < n) out[half + j] = __uint_as_float((x1 >> 9) | 0x3F800000u) - 1.0f;
  }
}
// deterministic mode: fold the fixed-point shadow of a range of the gradient buffer into it (and clear the shadow: a later flush of the same range adds nothing)
__global__ __launch_bounds__(256) void det_flush_kernel(float* __restrict__ g, long long* __restrict__ shadow, const unsigned* __restrict__ flag, int64_t n) {
  const bool bad = *flag != 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const long long q = shadow[i];
    if (q != 0 || bad) { g[i] = bad ? __int_as_float(0x7fc00000) : g[i] + (float)((double)q * (1.0 / 4294967296.0)); shadow[i] = 0; }
  }
}
void k_det_flush(spa3d_ctx* c, float* g, long long* shadow, const unsigned* flag, int64_t n) {
  if (c->dry || n <= 0) return;
  det_flush_kernel<<<(unsigned)std::min<int64_t>(cdiv(n, 256), 8192), 256, 0, c->stream>>>(g, shadow, flag, n); SPA_LAUNCH_CHECK(c);
}
void k_uniform_noise(spa3d_ctx* c, float* out, int64_t n, uint32_t k0, uint32_t k1) {
  if (c->dry || n == 0) return;
  int64_t half = (n + (n & 1)) / 2;
  uniform_noise_kernel<<<GRID1D(half, 256), 256, 0, c->stream>>>(out, n, half, k0, k1); SPA_LAUNCH_CHECK(c);
}

// ---------------------------------------------------------------------------------------------
// explicit instantiations
// ---------------------------------------------------------------------------------------------
#define INST(T)                                                                                                                        \
  template void k_layernorm<T>(spa3d_ctx*, const T*, const float*, T*, float*, int64_t, int);                                          \
  template void k_layernorm_bwd<T>(spa3d_ctx*, const T*, const float*, const float*, const T*, T*, float*, int64_t, int, const T*);    \
  template void k_rmsnorm_heads<T>(spa3d_ctx*, const T*, int64_t, const float*, T*, int64_t, int64_t, int, int);                       \
  template void k_rmsnorm_heads_bwd<T>(spa3d_ctx*, const T*, int64_t, const float*, const T*, int64_t, T*, int64_t, float*, int64_t,   \
                                       int, int);                                                                                     \
  template void k_softmax<T>(spa3d_ctx*, T*, const float*, int64_t, int, int, int);                                                    \
  template void k_softmax_bwd<T>(spa3d_ctx*, const T*, T*, int64_t, int, const float*, int64_t);                                                              \
  template void k_sin_embed<T>(spa3d_ctx*, const float*, int64_t, int, int, float, T*);                                                \
  template void k_embed_tokens<T>(spa3d_ctx*, const float*, int64_t, int, int, float, T*, int);                                           \
  template void k_colsum<T>(spa3d_ctx*, const T*, int64_t, int, int64_t, float*, int, int);                                                    \
  template void k_gelu<T>(spa3d_ctx*, const T*, T*, int64_t);                                                                          \
  template void k_add<T>(spa3d_ctx*, T*, const T*, int64_t);                                                                           \
  template void k_cast_from_f32<T>(spa3d_ctx*, const float*, T*, int64_t);                                                             \
  template void k_cast_to_f32<T>(spa3d_ctx*, const T*, float*, int64_t);                                                               \
  template void k_pack<T>(spa3d_ctx*, const float*, int64_t, int, int, T*, int64_t, T*, int64_t);                                      \
  template void k_transpose<T>(spa3d_ctx*, const T*, int, int, T*);                                                                    \
  template void k_set_readout_rows<T>(spa3d_ctx*, T*, const float*, int64_t, int, int);                                                \
  template void k_embed_maps<T>(spa3d_ctx*, const int32_t*, int64_t, int, int, int32_t*, int32_t*, T*, const float*, int);            \
  template void k_gather_rows<T>(spa3d_ctx*, const T*, int64_t, T*, int64_t, int);                                                     \
  template void k_rows_idx<T>(spa3d_ctx*, int, const T*, const int32_t*, T*, int64_t, int);                                                     \
  template void k_scatter_rows<T>(spa3d_ctx*, const T*, T*, int64_t, int64_t, int);                                                    \
  template void k_compact_tokens<T>(spa3d_ctx*, const T*, T*, int64_t, int, int);                                                      \
  template void k_broadcast_rows<T>(spa3d_ctx*, const float*, int, int, T*, int64_t);                                                  \
  template void k_bcast_grad<T>(spa3d_ctx*, const T*, int64_t, int64_t, int64_t, float*);                                              \
  template bool k_rank_fwd<T>(spa3d_ctx*, const T*, const T*, const float*, T*, int64_t, int, int, int64_t, int, int);                \
  template bool k_rank_bwd<T>(spa3d_ctx*, const T*, const T*, int64_t, int, int, int64_t, int, int, float*, float*);                                              \
  template void k_assemble_readout<T>(spa3d_ctx*, const T*, const T*, const int32_t*, int64_t, int, int, int, int, T*);                \
  template void k_share_assemble<T>(spa3d_ctx*, const T*, const T*, const int32_t*, const int32_t*, int64_t, int64_t, int, int, int, T*);   \
  template void k_share_expand<T>(spa3d_ctx*, const T*, const int32_t*, const int32_t*, int64_t, int64_t, int, int, const T*, T*);    \
  template void k_share_reduce<T>(spa3d_ctx*, const T*, const int32_t*, const int32_t*, int64_t, int64_t, int, int, int, T*);         \
  template void k_assemble_readout_bwd<T>(spa3d_ctx*, const T*, const int32_t*, int64_t, int, int, int, int, T*, float*);              \
  template void k_loss_bwd<T>(spa3d_ctx*, const float*, int64_t, int, const float*, const float*, const float*, float, float, T*, int, const float*);
INST(float)
INST(bf16_t)

// ---------------------------------------------------------------------------------------------
// Single-query attention (last block of the track encoder / readout stack: only token 0 leaves the stack,
// track_autoencoder_3d.py:187-188,286, so only its query row is needed; K/V still come from every token).
// One wave per (sequence, head).  lane = 4*kgrp + part: 16 keys per pass, each key row split over 4 lanes.
// RMSNorm of q/k, 1/sqrt(Dh), key mask, softmax and PV fused; the probabilities are kept (fp32, tiny) for the
// backward.  HBM-bound (K and V are read once): scores live in LDS and the key loop is a runtime loop so the kernels
// stay near 100 VGPRs (the first version unrolled 20 passes over 32-wide arrays: 256 VGPRs, one wave per SIMD, spills).
// CC = channels per lane (Dh/4) at compile time; vec: 16-byte accesses with 16-byte chunk i of a lane = chunk 4*i+part
// of the row, so the 4 lanes of a key touch 64 contiguous bytes per instruction.
// ---------------------------------------------------------------------------------------------
#define Q1_MAXS 320
template <typename T, int CC>
__device__ __forceinline__ int q1_chan(int j, int part, bool vec) {
  constexpr int NV = VecOf<T>::N;
  return vec ? (j / NV) * (4 * NV) + part * NV + (j % NV) : part * CC + j;
}
template <typename T, int CC>
__device__ __forceinline__ void q1_load(const T* p, bool vec, float (&f)[CC], int part) {
  constexpr int NV = VecOf<T>::N;
  if (vec) {
#pragma unroll
    for (int i = 0; i < CC / NV; ++i) {
      float t[NV]; load_vec<T, NV>(p + (4 * i + part) * NV, t);
#pragma unroll
      for (int j = 0; j < NV; ++j) f[i * NV + j] = t[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < CC; ++j) f[j] = ld(p + part * CC + j);
  }
}
template <typename T, int CC>
__device__ __forceinline__ void q1_store(T* p, bool vec, const float (&f)[CC], int part) {
  constexpr int NV = VecOf<T>::N;
  if (vec) {
#pragma unroll
    for (int i = 0; i < CC / NV; ++i) {
      float t[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) t[j] = f[i * NV + j];
      store_vec<T, NV>(p + (4 * i + part) * NV, t);
    }
  } else {
#pragma unroll
    for (int j = 0; j < CC; ++j) st(p + part * CC + j, f[j]);
  }
}
__device__ __forceinline__ float quad_sum(float v) { v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); return v; }
__device__ __forceinline__ float kgrp_sum(float v) {
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int CC>
__global__ __launch_bounds__(256) void attn_q1_fwd_kernel(const T* __restrict__ q0, int64_t ldq0, const T* __restrict__ k, const T* __restrict__ v,
                                                          int64_t ldk, int64_t ldv, const float* __restrict__ sq, const float* __restrict__ sk,
                                                          const float* __restrict__ km, int64_t nprob, int Smax, int H, T* __restrict__ o0,
                                                          float* __restrict__ p0, int vec_, const int32_t* __restrict__ seq_off) {
  __shared__ float scs[4][Q1_MAXS];
  constexpr int Dh = CC * 4;
  const int lane = threadIdx.x & 63, part = lane & 3, kg = lane >> 2, wv = threadIdx.x >> 6;
  const bool vec = vec_ != 0;
  float* sc = scs[wv];
  const float alpha = rsqrtf((float)Dh);
  float sqv[CC], skv[CC];
#pragma unroll
  for (int j = 0; j < CC; ++j) { const int ch = q1_chan<T, CC>(j, part, vec); sqv[j] = sq[ch]; skv[j] = sk[ch]; }
  for (int64_t prob = (int64_t)blockIdx.x * 4 + wv; prob < nprob; prob += (int64_t)gridDim.x * 4) {
    const int64_t seq = prob / H; const int h = (int)(prob - seq * H);
    // ragged sequences (token pruning): rows [seq_off[seq], seq_off[seq+1]) of the compact tensors; dense otherwise
    const int64_t rowbase = seq_off ? (int64_t)seq_off[seq] : seq * Smax;
    const int S = seq_off ? seq_off[seq + 1] - (int)rowbase : Smax;
    const int nit = (S + 15) / 16;
    float qh[CC];
    q1_load<T, CC>(q0 + seq * ldq0 + h * Dh, vec, qh, part);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < CC; ++j) ss += qh[j] * qh[j];
    const float rq = rsqrtf(quad_sum(ss) / Dh + 1e-6f);
#pragma unroll
    for (int j = 0; j < CC; ++j) qh[j] *= rq * sqv[j];
    float m = -3.4028234663852886e38f;
#pragma unroll 2
    for (int it = 0; it < nit; ++it) {
      const int key = it * 16 + kg;
      const int kr_ = key < S ? key : S - 1;  // absent keys re-read the last row (keeps the quad shuffles convergent), result unused
      float kv_[CC];
      q1_load<T, CC>(k + (rowbase + kr_) * ldk + h * Dh, vec, kv_, part);
      float ks = 0.f, d = 0.f;
#pragma unroll
      for (int j = 0; j < CC; ++j) { ks += kv_[j] * kv_[j]; d += qh[j] * kv_[j] * skv[j]; }
      ks = quad_sum(ks); d = quad_sum(d);
      float lg = d * rsqrtf(ks / Dh + 1e-6f) * alpha;
      if (km && km[rowbase + kr_] == 0.f) lg = -3.4028234663852886e38f;
      if (key < S) { m = fmaxf(m, lg); if (part == 0) sc[key] = lg; }
    }
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float l = 0.f;
    for (int key = lane; key < S; key += 64) { const float e = expf(sc[key] - m); sc[key] = e; l += e; }  // own-wave LDS, program order
    l = wave_sum(l);
    const float inv = 1.f / l;
    float acc[CC];
#pragma unroll
    for (int j = 0; j < CC; ++j) acc[j] = 0.f;
#pragma unroll 2
    for (int it = 0; it < nit; ++it) {
      const int key = it * 16 + kg;
      if (key < S) {
        const float p = sc[key] * inv;
        if (part == 0) p0[prob * Smax + key] = p;
        float vv[CC];
        q1_load<T, CC>(v + (rowbase + key) * ldv + h * Dh, vec, vv, part);
#pragma unroll
        for (int j = 0; j < CC; ++j) acc[j] += p * vv[j];
      }
    }
#pragma unroll
    for (int j = 0; j < CC; ++j) acc[j] = kgrp_sum(acc[j]);
    if (kg == 0) q1_store<T, CC>(o0 + seq * (int64_t)H * Dh + h * Dh, vec, acc, part);
  }
}
template <typename T>
void k_attn_q1_fwd(spa3d_ctx* c, const T* q0, int64_t ldq0, const T* k, const T* v, int64_t ldk, int64_t ldv, const float* sq, const float* sk,
                   const float* km, int64_t nseq, int S, int H, int Dh, T* o0, float* p0, const int32_t* seq_off) {
  if (c->dry || nseq == 0) return;
  if (S > Q1_MAXS || Dh % 4 || Dh > 128) { if (!c->hip_err) { c->hip_err = -3; c->err = "attn_q1: S <= 320 and Dh % 4 == 0, Dh <= 128 required"; } return; }
  const int64_t nprob = nseq * H;
  ProfScope ps(c, PROF_ATTN_Q1, 4.0 * (double)nprob * S * Dh, (double)nprob * S * Dh * 2.0 * sizeof(T) + (double)nprob * S * 4.0);  // K, V once (+ p0)
  ps.tag(nseq, S, H, 0);
  unsigned g = (unsigned)std::min<int64_t>(cdiv(nprob, 4), 8192);
  constexpr int NV = VecOf<T>::N;
  const int vec = (Dh % (4 * NV) == 0 && ldk % NV == 0 && ldv % NV == 0 && ldq0 % NV == 0 &&
                   ((((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)q0) | ((uintptr_t)o0)) & 15) == 0) ? 1 : 0;
#define Q1F(CCv) attn_q1_fwd_kernel<T, CCv><<<g, 256, 0, c->stream>>>(q0, ldq0, k, v, ldk, ldv, sq, sk, km, nprob, S, H, o0, p0, vec, seq_off)
  switch (Dh / 4) { case 24: Q1F(24); break; case 16: Q1F(16); break; case 32: Q1F(32); break; case 8: Q1F(8); break; case 4: Q1F(4); break;
    case 2: Q1F(2); break; default: if (!c->hip_err) { c->hip_err = -3; c->err = "attn_q1: unsupported head width"; } return; }
#undef Q1F
  SPA_LAUNCH_CHECK(c);
}

// backward of the above: dq0 [nseq, H*Dh]; dk, dv for EVERY key row (overwritten); scale gradients accumulated.
// With x^ the RMS-normalised rows, q^ = x^_q*s_q, k^ = x^_k*s_k and ds_k = p_k (dp_k - sum p dp) / sqrt(Dh):
//   u = sum_k ds_k x^_k  gives both  dq^ = u*s_k  and  ds_k(scale) = q^*u;   dk^_k = ds_k q^;  dv_k = p_k dO.
// Registers: q^, dO, u (+ one key row); the scales and the scale-gradient accumulators live in LDS.
template <typename T, int CC, bool DET = false>  // DET: deterministic-gradient mode (common.hpp): the scale gradients accumulate as 64-bit fixed point, in LDS and in the shadow
__global__ __launch_bounds__(256) void attn_q1_bwd_kernel(const T* __restrict__ q0, int64_t ldq0, const T* __restrict__ k, const T* __restrict__ v,
                                                          int64_t ldk, int64_t ldv, const float* __restrict__ sq, const float* __restrict__ sk,
                                                          const float* __restrict__ km, int64_t nprob, int Smax, int H,
                                                          const float* __restrict__ p0, const T* __restrict__ d_o0, T* __restrict__ dq0,
                                                          T* __restrict__ dk, T* __restrict__ dv, float* __restrict__ dsq, float* __restrict__ dsk,
                                                          int vec_, const int32_t* __restrict__ seq_off) {
  __shared__ float dps[4][Q1_MAXS];
  __shared__ float scl[2][4 * CC];  // s_q, s_k in lane-channel order [part][j]
  __shared__ float red[2][4 * CC];  // block accumulators of d s_q, d s_k
  __shared__ unsigned long long redq[DET ? 2 : 1][DET ? 4 * CC : 1];  // DET: the same in 2^-32 fixed point (integer LDS atomics do not depend on arrival order)
  constexpr int Dh = CC * 4;
  constexpr int NV = VecOf<T>::N;
  constexpr int G = (CC % NV == 0) ? NV : 1;  // output chunk; vec implies G == NV
  const int lane = threadIdx.x & 63, part = lane & 3, kg = lane >> 2, wv = threadIdx.x >> 6;
  const bool vec = vec_ != 0;
  float* dp = dps[wv];
  const float* sql = scl[0] + part * CC;
  const float* skl = scl[1] + part * CC;
  const float alpha = rsqrtf((float)Dh);
  for (int t = threadIdx.x; t < 4 * CC; t += 256) {
    const int pt = t / CC, j = t - pt * CC;
    const int ch = q1_chan<T, CC>(j, pt, vec);
    scl[0][t] = sq[ch]; scl[1][t] = sk[ch]; red[0][t] = 0.f; red[1][t] = 0.f;
    if constexpr (DET) { redq[0][t] = 0ull; redq[1][t] = 0ull; }
  }
  __syncthreads();
  for (int64_t prob = (int64_t)blockIdx.x * 4 + wv; prob < nprob; prob += (int64_t)gridDim.x * 4) {
    const int64_t seq = prob / H; const int h = (int)(prob - seq * H);
    // ragged sequences (token pruning): rows [seq_off[seq], seq_off[seq+1]) of the compact tensors; dense otherwise
    const int64_t rowbase = seq_off ? (int64_t)seq_off[seq] : seq * Smax;
    const int S = seq_off ? seq_off[seq + 1] - (int)rowbase : Smax;
    const int nit = (S + 15) / 16;
    float qs[CC], dout[CC];
    q1_load<T, CC>(q0 + seq * ldq0 + h * Dh, vec, qs, part);
    q1_load<T, CC>(d_o0 + seq * (int64_t)H * Dh + h * Dh, vec, dout, part);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < CC; ++j) ss += qs[j] * qs[j];
    const float rq = rsqrtf(quad_sum(ss) / Dh + 1e-6f);
#pragma unroll
    for (int j = 0; j < CC; ++j) qs[j] *= rq * sql[j];  // q^
    // pass 1: dp_k = dO . V_k (to LDS) and sum_k p_k dp_k
    float pd = 0.f;
#pragma unroll 2
    for (int it = 0; it < nit; ++it) {
      const int key = it * 16 + kg;
      const int kr_ = key < S ? key : S - 1;
      float vv[CC];
      q1_load<T, CC>(v + (rowbase + kr_) * ldv + h * Dh, vec, vv, part);
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < CC; ++j) d += dout[j] * vv[j];
      d = quad_sum(d);
      if (key < S && part == 0) { dp[key] = d; pd += p0[prob * Smax + key] * d; }
    }
    pd = wave_sum(pd);
    // pass 2: per key dv, dk (through the RMSNorm) and u
    float u[CC];
#pragma unroll
    for (int j = 0; j < CC; ++j) u[j] = 0.f;
    for (int it = 0; it < nit; ++it) {
      const int key = it * 16 + kg;
      const bool valid = key < S;
      const int kr_ = valid ? key : S - 1;
      const int64_t roff = rowbase + kr_;
      float xk[CC];
      q1_load<T, CC>(k + roff * ldk + h * Dh, vec, xk, part);
      float ks = 0.f;
#pragma unroll
      for (int j = 0; j < CC; ++j) ks += xk[j] * xk[j];
      const float rk = rsqrtf(quad_sum(ks) / Dh + 1e-6f);
      const float p = valid ? p0[prob * Smax + kr_] : 0.f;
      const bool keep = !(km && km[rowbase + kr_] == 0.f);
      const float ds = (valid && keep) ? p * (dp[kr_] - pd) * alpha : 0.f;  // where() passes no gradient to masked logits
      float gx = 0.f;
#pragma unroll
      for (int j = 0; j < CC; ++j) {
        xk[j] *= rk;                          // x^ of the key row
        u[j] += ds * xk[j];
        gx += qs[j] * skl[j] * xk[j];         // (dk^ * s_k) . x^ / ds
      }
      gx = quad_sum(gx) * ds / Dh;
      if (valid) {
        T* dkr = dk + roff * ldk + h * Dh;
        T* dvr = dv + roff * ldv + h * Dh;
#pragma unroll
        for (int i = 0; i < CC / G; ++i) {
          float ok[G], ov[G];
#pragma unroll
          for (int jj = 0; jj < G; ++jj) {
            const int j = i * G + jj;
            ok[jj] = rk * (ds * qs[j] * skl[j] - xk[j] * gx);
            ov[jj] = p * dout[j];
          }
          if (vec) {
            store_vec<T, G>(dkr + (4 * i + part) * G, ok);
            store_vec<T, G>(dvr + (4 * i + part) * G, ov);
          } else {
#pragma unroll
            for (int jj = 0; jj < G; ++jj) { st(dkr + part * CC + i * G + jj, ok[jj]); st(dvr + part * CC + i * G + jj, ov[jj]); }
          }
        }
      }
    }
    // query side: dq^ = u*s_k; d s_k += q^*u; d s_q += dq^ * x^_q; dq through the RMSNorm
    float xq[CC];
    q1_load<T, CC>(q0 + seq * ldq0 + h * Dh, vec, xq, part);
    float gq = 0.f;
#pragma unroll
    for (int j = 0; j < CC; ++j) {
      u[j] = kgrp_sum(u[j]);
      xq[j] *= rq;
      gq += u[j] * skl[j] * sql[j] * xq[j];
    }
    gq = quad_sum(gq) / Dh;
    if (kg == 0) {
#pragma unroll
      for (int j = 0; j < CC; ++j) {
        const float dqh = u[j] * skl[j];
        // (deterministic mode: LDS float atomics depend on arrival order too -- every contribution goes straight to the fixed-point shadow)
        if constexpr (DET) {
          atomicAdd(&redq[1][part * CC + j], (unsigned long long)__float2ll_rn(qs[j] * u[j] * 4294967296.f));
          atomicAdd(&redq[0][part * CC + j], (unsigned long long)__float2ll_rn(dqh * xq[j] * 4294967296.f));
        } else { atomicAdd(&red[1][part * CC + j], qs[j] * u[j]); atomicAdd(&red[0][part * CC + j], dqh * xq[j]); }
        xq[j] = rq * (dqh * sql[j] - xq[j] * gq);
      }
      q1_store<T, CC>(dq0 + seq * (int64_t)H * Dh + h * Dh, vec, xq, part);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 4 * CC; t += 256) {
    const int pt = t / CC, j = t - pt * CC;
    const int ch = q1_chan<T, CC>(j, pt, vec);
    if constexpr (DET) {  // (a non-finite partial saturates the conversion: the 16-bit dq / dk / dv beside it are NaN already and reach every upstream leaf)
      grad_add(dsq + ch, (float)((double)(long long)redq[0][t] * (1.0 / 4294967296.0))); grad_add(dsk + ch, (float)((double)(long long)redq[1][t] * (1.0 / 4294967296.0)));
    } else { atomicAdd(dsq + ch, red[0][t]); atomicAdd(dsk + ch, red[1][t]); }
  }
}
template <typename T>
void k_attn_q1_bwd(spa3d_ctx* c, const T* q0, int64_t ldq0, const T* k, const T* v, int64_t ldk, int64_t ldv, const float* sq, const float* sk,
                   const float* km, int64_t nseq, int S, int H, int Dh, const float* p0, const T* d_o0, T* dq0, T* dk, T* dv, float* dsq,
                   float* dsk, const int32_t* seq_off) {
  if (c->dry || nseq == 0) return;
  if (S > Q1_MAXS || Dh % 4 || Dh > 128) { if (!c->hip_err) { c->hip_err = -3; c->err = "attn_q1: S <= 320 and Dh % 4 == 0, Dh <= 128 required"; } return; }
  const int64_t nprob = nseq * H;
  ProfScope ps(c, PROF_ATTN_Q1, 8.0 * (double)nprob * S * Dh, (double)nprob * S * Dh * 4.0 * sizeof(T) + (double)nprob * S * 4.0);  // K, V read; dK, dV written
  ps.tag(nseq, S, H, 1);
  unsigned g = (unsigned)std::min<int64_t>(cdiv(nprob, 4), 4096);
  constexpr int NV = VecOf<T>::N;
  const int vec = (Dh % (4 * NV) == 0 && ldk % NV == 0 && ldv % NV == 0 && ldq0 % NV == 0 &&
                   ((((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)dk) | ((uintptr_t)dv) | ((uintptr_t)q0) | ((uintptr_t)d_o0) |
                     ((uintptr_t)dq0)) & 15) == 0) ? 1 : 0;
#define Q1B(CCv) do { if (c->det_grads) attn_q1_bwd_kernel<T, CCv, true><<<g, 256, 0, c->stream>>>(q0, ldq0, k, v, ldk, ldv, sq, sk, km, nprob, S, H, p0, d_o0, dq0, dk, dv, dsq, dsk, vec, seq_off); \
                      else attn_q1_bwd_kernel<T, CCv, false><<<g, 256, 0, c->stream>>>(q0, ldq0, k, v, ldk, ldv, sq, sk, km, nprob, S, H, p0, d_o0, dq0, dk, dv, dsq, dsk, vec, seq_off); } while (0)
  switch (Dh / 4) { case 24: Q1B(24); break; case 16: Q1B(16); break; case 32: Q1B(32); break; case 8: Q1B(8); break; case 4: Q1B(4); break;
    case 2: Q1B(2); break; default: if (!c->hip_err) { c->hip_err = -3; c->err = "attn_q1: unsupported head width"; } return; }
#undef Q1B
  SPA_LAUNCH_CHECK(c);
}
// dst[i*stride_rows][:] += src[i][:]
template <typename T>
__global__ void add_rows_strided_kernel(T* __restrict__ dst, const T* __restrict__ src, int64_t drows, int64_t n, int d) {
  const int64_t tot = n * d;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    int64_t r = i / d; int j = (int)(i - r * d);
    T* p = dst + r * drows * d + j;
    st(p, ld(p) + ld(src + i));
  }
}
template <typename T> void k_add_rows_strided(spa3d_ctx* c, T* dst, const T* src, int64_t drows, int64_t n, int d) {
  if (c->dry || n == 0) return;
  add_rows_strided_kernel<T><<<GRID1D(n * d, 256), 256, 0, c->stream>>>(dst, src, drows, n, d); SPA_LAUNCH_CHECK(c);
}
#define INST_Q1(T)                                                                                                                     \
  template void k_attn_q1_fwd<T>(spa3d_ctx*, const T*, int64_t, const T*, const T*, int64_t, int64_t, const float*, const float*,       \
                                 const float*, int64_t, int, int, int, T*, float*, const int32_t*);                                     \
  template void k_attn_q1_bwd<T>(spa3d_ctx*, const T*, int64_t, const T*, const T*, int64_t, int64_t, const float*, const float*,       \
                                 const float*, int64_t, int, int, int, const float*, const T*, T*, T*, T*, float*, float*, const int32_t*); \
  template void k_add_rows_strided<T>(spa3d_ctx*, T*, const T*, int64_t, int64_t, int);
INST_Q1(float)
INST_Q1(bf16_t)

// ---------------------------------------------------------------------------------------------
// TRAJAN track pooling (track_autoencoder.py:230-232): mean of the frame tokens over VISIBLE frames
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void vis_mean_pool_kernel(const T* __restrict__ tok, const float* __restrict__ vis, int64_t nseq, int T_, int d, T* __restrict__ out) {
  const int64_t tot = nseq * d;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    int64_t s_ = i / d; int j = (int)(i - s_ * d);
    float a = 0.f, cnt = 0.f;
    for (int t = 0; t < T_; ++t) { const float v = vis[s_ * T_ + t] != 0.f ? 1.f : 0.f; a += ld(tok + (s_ * T_ + t) * d + j) * v; cnt += v; }
    st(out + i, a / fmaxf(1.f, cnt));
  }
}
template <typename T> void k_vis_mean_pool(spa3d_ctx* c, const T* tok, const float* vis, int64_t nseq, int T_, int d, T* out) {
  if (c->dry || nseq == 0) return;
  vis_mean_pool_kernel<T><<<GRID1D(nseq * d, 256), 256, 0, c->stream>>>(tok, vis, nseq, T_, d, out); SPA_LAUNCH_CHECK(c);
}
template <typename T>
__global__ void vis_mean_pool_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ vis, int64_t nseq, int T_, int d, T* __restrict__ dtok) {
  const int64_t tot = nseq * T_ * d;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    int64_t r = i / d; int j = (int)(i - r * d);
    int64_t s_ = r / T_;
    float cnt = 0.f;
    for (int t = 0; t < T_; ++t) cnt += vis[s_ * T_ + t] != 0.f ? 1.f : 0.f;
    const float v = vis[r] != 0.f ? 1.f : 0.f;
    st(dtok + i, ld(dout + s_ * d + j) * v / fmaxf(1.f, cnt));
  }
}
template <typename T> void k_vis_mean_pool_bwd(spa3d_ctx* c, const T* dout, const float* vis, int64_t nseq, int T_, int d, T* dtok) {
  if (c->dry || nseq == 0) return;
  vis_mean_pool_bwd_kernel<T><<<GRID1D(nseq * T_ * d, 256), 256, 0, c->stream>>>(dout, vis, nseq, T_, d, dtok); SPA_LAUNCH_CHECK(c);
}
// key mask of the 2-D model (ta:217-223): km[seq][t] = visible & (t < boundary), no readout key
__global__ void keymask2d_kernel(const float* __restrict__ vis, const int32_t* __restrict__ boundary, int64_t nseq, int N, int T_, float* km) {
  const int64_t n = nseq * T_;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int64_t s_ = i / T_; int t = (int)(i - s_ * T_);
    km[i] = (vis[i] != 0.f && t < boundary[s_ / N]) ? 1.f : 0.f;
  }
}
void k_keymask2d(spa3d_ctx* c, const float* visible, const int32_t* boundary, int64_t nseq, int N, int T_, float* km) {
  if (c->dry || nseq == 0) return;
  keymask2d_kernel<<<GRID1D(nseq * T_, 256), 256, 0, c->stream>>>(visible, boundary, nseq, N, T_, km); SPA_LAUNCH_CHECK(c);
}
template void k_vis_mean_pool<float>(spa3d_ctx*, const float*, const float*, int64_t, int, int, float*);
template void k_vis_mean_pool<bf16_t>(spa3d_ctx*, const bf16_t*, const float*, int64_t, int, int, bf16_t*);
template void k_vis_mean_pool_bwd<float>(spa3d_ctx*, const float*, const float*, int64_t, int, int, float*);
template void k_vis_mean_pool_bwd<bf16_t>(spa3d_ctx*, const bf16_t*, const float*, int64_t, int, int, bf16_t*);
SPA_DET_UPLOAD_DEF(det_upload_kernels)
}  // namespace SPA_NS
