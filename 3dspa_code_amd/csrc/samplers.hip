// samplers.hip -- the track-feature producers that feed the hot path: bilinear sampling of
// DINOv2 patch features / depth at 2-D track positions and the 2D->3D lift (inference.py:287-447, Python double loops
// over N*T there).  HBM/L2-bound gathers: one wave per (track, frame), 16-byte lane-contiguous accesses along the
// channel axis.  Arithmetic is float32 in the reference's exact operation order (no contraction, IEEE division), so the
// results are bit-identical to the reference functions run under NumPy >= 2 (tests/golden/sampler_golden.npz).
#include "common.hpp"

struct Corner { int x0, y0, x1, y1; float wx, wy; };
// floor / +1 / weights / clamp (inference.py:305-316, :369-380, :415-425): weights BEFORE clamping
__device__ __forceinline__ Corner corners(float px, float py, int Wm, int Hm) {
  Corner c;
  const float fx0 = floorf(px), fy0 = floorf(py);
  c.wx = __fsub_rn(px, fx0); c.wy = __fsub_rn(py, fy0);
  const long long ix = (long long)fx0, iy = (long long)fy0;
  c.x0 = (int)min(max(ix, 0LL), (long long)Wm - 1); c.x1 = (int)min(max(ix + 1, 0LL), (long long)Wm - 1);
  c.y0 = (int)min(max(iy, 0LL), (long long)Hm - 1); c.y1 = (int)min(max(iy + 1, 0LL), (long long)Hm - 1);
  return c;
}
__device__ __forceinline__ float blend(float f00, float f01, float f10, float f11, float wx, float wy) {
  const float ax = __fsub_rn(1.f, wx), ay = __fsub_rn(1.f, wy);
  float r = __fmul_rn(__fmul_rn(f00, ax), ay);
  r = __fadd_rn(r, __fmul_rn(__fmul_rn(f01, wx), ay));
  r = __fadd_rn(r, __fmul_rn(__fmul_rn(f10, ax), wy));
  r = __fadd_rn(r, __fmul_rn(__fmul_rn(f11, wx), wy));
  return r;
}
__device__ __forceinline__ float depth_at(const float* __restrict__ depth, int t, int H, int W, float x, float y) {
  const Corner c = corners(x, y, W, H);
  const float* d = depth + (int64_t)t * H * W;
  return blend(d[(int64_t)c.y0 * W + c.x0], d[(int64_t)c.y0 * W + c.x1], d[(int64_t)c.y1 * W + c.x0], d[(int64_t)c.y1 * W + c.x1], c.wx, c.wy);
}

// sample_dino_features_for_tracks (inference.py:339-395): out[n][t][:] = bilinear(feat[t], tracks[n][t] * scale)
template <typename TOUT>
__global__ __launch_bounds__(256) void sample_dino_kernel(const float* __restrict__ feat, const float* __restrict__ tracks, int64_t NT, int T,
                                                          int Hp, int Wp, int D, float scale_w, float scale_h, TOUT* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < NT; p += (int64_t)gridDim.x * 4) {
    const int t = (int)(p % T);
    const float px = __fmul_rn(tracks[p * 2], scale_w), py = __fmul_rn(tracks[p * 2 + 1], scale_h);
    const Corner c = corners(px, py, Wp, Hp);
    const float* base = feat + (int64_t)t * Hp * Wp * D;
    const float* r00 = base + ((int64_t)c.y0 * Wp + c.x0) * D; const float* r01 = base + ((int64_t)c.y0 * Wp + c.x1) * D;
    const float* r10 = base + ((int64_t)c.y1 * Wp + c.x0) * D; const float* r11 = base + ((int64_t)c.y1 * Wp + c.x1) * D;
    TOUT* o = out + p * D;
    if ((D & 3) == 0) {
      for (int ch = lane * 4; ch < D; ch += 256) {
        const float4 a = *(const float4*)(r00 + ch), b = *(const float4*)(r01 + ch), cc = *(const float4*)(r10 + ch), d = *(const float4*)(r11 + ch);
        const float v0 = blend(a.x, b.x, cc.x, d.x, c.wx, c.wy), v1 = blend(a.y, b.y, cc.y, d.y, c.wx, c.wy);
        const float v2 = blend(a.z, b.z, cc.z, d.z, c.wx, c.wy), v3 = blend(a.w, b.w, cc.w, d.w, c.wx, c.wy);
        if constexpr (sizeof(TOUT) == 4) *(float4*)(o + ch) = make_float4(v0, v1, v2, v3);
        else { uint2 u; u.x = f2bf_pack2(v0, v1); u.y = f2bf_pack2(v2, v3); *(uint2*)(o + ch) = u; }
      }
    } else {
      for (int ch = lane; ch < D; ch += 64) st(o + ch, blend(r00[ch], r01[ch], r10[ch], r11[ch], c.wx, c.wy));
    }
  }
}

// sample_depth_features_for_tracks (inference.py:398-447): 256 channels, [0]=d, [1]=d/10, [2]=d - d_prev (t>0), rest 0
__global__ __launch_bounds__(256) void sample_depth_kernel(const float* __restrict__ depth, const float* __restrict__ tracks, int64_t NT, int T,
                                                           int H, int W, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < NT; p += (int64_t)gridDim.x * 4) {
    const int t = (int)(p % T);
    const float d = depth_at(depth, t, H, W, tracks[p * 2], tracks[p * 2 + 1]);
    float g = 0.f;
    if (t > 0) g = __fsub_rn(d, depth_at(depth, t - 1, H, W, tracks[(p - 1) * 2], tracks[(p - 1) * 2 + 1]));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane == 0) v = make_float4(d, __fdiv_rn(d, 10.0f), g, 0.f);
    *(float4*)(out + p * 256 + lane * 4) = v;
  }
}

// lift_2d_to_3d (inference.py:287-336)
__global__ void lift_kernel(const float* __restrict__ tracks, const float* __restrict__ depth, int64_t NT, int T, int H, int W, float fx,
                            float fy, float cx, float cy, float* __restrict__ out) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < NT; p += (int64_t)gridDim.x * 256) {
    const int t = (int)(p % T);
    const float x = tracks[p * 2], y = tracks[p * 2 + 1];
    const float z = depth_at(depth, t, H, W, x, y);
    out[p * 3] = __fdiv_rn(__fmul_rn(__fsub_rn(x, cx), z), fx);
    out[p * 3 + 1] = __fdiv_rn(__fmul_rn(__fsub_rn(y, cy), z), fy);
    out[p * 3 + 2] = z;
  }
}

extern "C" {

int spa3d_op_sample_dino(const float* feat, const float* tracks_2d, int32_t N, int32_t T, int32_t Hp, int32_t Wp, int32_t D, int32_t H,
                         int32_t W, void* out, int32_t out_dtype, void* stream) {
  if (!feat || !tracks_2d || !out || N <= 0 || T <= 0 || Hp <= 0 || Wp <= 0 || D <= 0 || H <= 0 || W <= 0) return SPA3D_ERR_ARG;
  if ((((uintptr_t)feat) | ((uintptr_t)out)) & 15) return SPA3D_ERR_ARG;
  spa3d_ctx c; c.stream = (hipStream_t)stream;
  const int64_t NT = (int64_t)N * T;
  const float sw = (float)((double)Wp / (double)W), sh = (float)((double)Hp / (double)H);  // python floats, weak against f32 (:359-360)
  const unsigned g = (unsigned)std::min<int64_t>((NT + 3) / 4, 16384);
  if (out_dtype == SPA3D_F32) sample_dino_kernel<float><<<g, 256, 0, c.stream>>>(feat, tracks_2d, NT, T, Hp, Wp, D, sw, sh, (float*)out);
  else sample_dino_kernel<bf16_t><<<g, 256, 0, c.stream>>>(feat, tracks_2d, NT, T, Hp, Wp, D, sw, sh, (bf16_t*)out);
  SPA_LAUNCH_CHECK(&c);
  return c.hip_err ? SPA3D_ERR_HIP : SPA3D_OK;
}

int spa3d_op_sample_depth_features(const float* depth, const float* tracks_2d, int32_t N, int32_t T, int32_t H, int32_t W, float* out,
                                   void* stream) {
  if (!depth || !tracks_2d || !out || N <= 0 || T <= 0 || H <= 0 || W <= 0 || (((uintptr_t)out) & 15)) return SPA3D_ERR_ARG;
  spa3d_ctx c; c.stream = (hipStream_t)stream;
  const int64_t NT = (int64_t)N * T;
  sample_depth_kernel<<<(unsigned)std::min<int64_t>((NT + 3) / 4, 16384), 256, 0, c.stream>>>(depth, tracks_2d, NT, T, H, W, out);
  SPA_LAUNCH_CHECK(&c);
  return c.hip_err ? SPA3D_ERR_HIP : SPA3D_OK;
}

int spa3d_op_lift_2d_to_3d(const float* tracks_2d, const float* depth, int32_t N, int32_t T, int32_t H, int32_t W, const double* intrinsics,
                           float* out, void* stream) {
  if (!depth || !tracks_2d || !out || N <= 0 || T <= 0 || H <= 0 || W <= 0) return SPA3D_ERR_ARG;
  spa3d_ctx c; c.stream = (hipStream_t)stream;
  double fx, fy, cx, cy;
  if (intrinsics) { fx = intrinsics[0]; fy = intrinsics[1]; cx = intrinsics[2]; cy = intrinsics[3]; }
  else { fx = fy = (double)std::max(H, W); cx = W / 2.0; cy = H / 2.0; }  // inference.py:297-300
  const int64_t NT = (int64_t)N * T;
  lift_kernel<<<(unsigned)std::min<int64_t>((NT + 255) / 256, 4096), 256, 0, c.stream>>>(tracks_2d, depth, NT, T, H, W, (float)fx, (float)fy,
                                                                                             (float)cx, (float)cy, out);
  SPA_LAUNCH_CHECK(&c);
  return c.hip_err ? SPA3D_ERR_HIP : SPA3D_OK;
}

}  // extern "C"
