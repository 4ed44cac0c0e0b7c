"""CPU oracle for the track-feature producers either side of the hot path.  TEST
INFRASTRUCTURE ONLY -- the product never imports this.

Vectorised NumPy restatement of /root/reference/inference.py:
  lift_2d_to_3d                      :287-336
  sample_dino_features_for_tracks    :339-395
  sample_depth_features_for_tracks   :398-447
PINNED: tests/golden/sampler_golden.npz holds outputs of the reference's own functions executed in this container
(tests/golden/make_sampler_golden.py); tests/test_sampler.py requires bit-equality of this restatement with them.
Arithmetic: float32 end to end for float32 inputs (NumPy >= 2 promotion, which is what ran); the reference's op order
`f00*(1-wx)*(1-wy) + f01*wx*(1-wy) + f10*(1-wx)*wy + f11*wx*wy` is kept, weights are taken BEFORE the indices are
clamped (points outside the image extrapolate against clamped texels, exactly as the reference does).
"""
import numpy as np

F = np.float32


def _corners(px, py, Wm, Hm):
  """floor / +1 / weights / clamp of inference.py:305-316 (and :369-380, :415-425) for float32 coordinate arrays."""
  fx0, fy0 = np.floor(px), np.floor(py)
  wx, wy = (px - fx0).astype(F), (py - fy0).astype(F)
  x0, y0 = fx0.astype(np.int64), fy0.astype(np.int64)
  x1, y1 = x0 + 1, y0 + 1
  x0, x1 = np.clip(x0, 0, Wm - 1), np.clip(x1, 0, Wm - 1)
  y0, y1 = np.clip(y0, 0, Hm - 1), np.clip(y1, 0, Hm - 1)
  return x0, y0, x1, y1, wx, wy


def _blend(f00, f01, f10, f11, wx, wy):
  one = F(1)
  return (f00 * (one - wx) * (one - wy) + f01 * wx * (one - wy) + f10 * (one - wx) * wy + f11 * wx * wy).astype(F)


def _depth_at(depth, tracks_2d):
  T = depth.shape[0]
  x, y = tracks_2d[..., 0].astype(F), tracks_2d[..., 1].astype(F)  # [N,T]
  x0, y0, x1, y1, wx, wy = _corners(x, y, depth.shape[2], depth.shape[1])
  t = np.arange(T)[None, :]
  d = depth[..., 0]
  return _blend(d[t, y0, x0], d[t, y0, x1], d[t, y1, x0], d[t, y1, x1], wx, wy), x, y


def lift_2d_to_3d(tracks_2d, depth, intrinsics=None):
  """inference.py:287-336 -> [N,T,3] float32."""
  if intrinsics is None:  # :297-300
    H, W = depth.shape[1:3]
    fx = fy = max(H, W)
    cx, cy = W / 2, H / 2
  else:
    fx, fy, cx, cy = intrinsics
  z, x, y = _depth_at(depth.astype(F), tracks_2d)
  out = np.empty(tracks_2d.shape[:2] + (3,), dtype=F)
  out[..., 0] = ((x - F(cx)) * z / F(fx)).astype(F)  # :331
  out[..., 1] = ((y - F(cy)) * z / F(fy)).astype(F)
  out[..., 2] = z
  return out


def sample_dino_features_for_tracks(dino_features, tracks_2d, video_shape):
  """inference.py:339-395 -> [N,T,D] float32."""
  if dino_features is None:
    return None
  T, Hp, Wp, D = dino_features.shape
  _, H, W, _ = video_shape
  scale_h, scale_w = Hp / H, Wp / W  # python floats (:359-360); weak against the float32 coordinates
  px = (tracks_2d[..., 0].astype(F) * F(scale_w)).astype(F)
  py = (tracks_2d[..., 1].astype(F) * F(scale_h)).astype(F)
  x0, y0, x1, y1, wx, wy = _corners(px, py, Wp, Hp)
  t = np.arange(T)[None, :]
  f = dino_features.astype(F)
  return _blend(f[t, y0, x0], f[t, y0, x1], f[t, y1, x0], f[t, y1, x1], wx[..., None], wy[..., None])


def sample_depth_features_for_tracks(depth, tracks_2d):
  """inference.py:398-447 -> [N,T,256] float32: channel 0 depth, 1 depth/10, 2 temporal difference (0 at t=0), rest 0."""
  if depth is None:
    return None
  d, _, _ = _depth_at(depth.astype(F), tracks_2d)
  out = np.zeros(tracks_2d.shape[:2] + (256,), dtype=F)
  out[..., 0] = d
  out[..., 1] = (d / F(10.0)).astype(F)
  out[:, 1:, 2] = (d[:, 1:].astype(np.float64) - d[:, :-1].astype(np.float64)).astype(F)  # :441-443 (float64 array read-back)
  return out
