"""Track-feature producers with the reference's names and signatures (inference.py:287-447), on the GPU.

The reference runs Python double loops over N*T with NumPy scalars; here each call is one HIP kernel
(csrc/samplers.hip).  Inputs may be NumPy arrays or torch tensors; outputs are float32 torch tensors on the GPU,
bit-identical to the reference functions run under NumPy >= 2 (tests/test_sampler.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .model import _stream


def _dev(x, device):
  t = torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x)
  return t.to(device=device, dtype=torch.float32).contiguous()


def lift_2d_to_3d(tracks_2d, depth, intrinsics=None, device='cuda'):
  """tracks_2d [N,T,2], depth [T,H,W,1] -> tracks_3d [N,T,3] in camera coordinates (inference.py:287-336)."""
  tr, dp = _dev(tracks_2d, device), _dev(depth, device)
  N, T = tr.shape[:2]
  if dp.dim() != 4 or dp.shape[0] != T or dp.shape[-1] != 1:
    raise ValueError(f'depth must be [T,H,W,1] with T={T}, got {tuple(dp.shape)}')
  out = torch.empty(N, T, 3, dtype=torch.float32, device=tr.device)
  intr = None
  if intrinsics is not None:
    fx, fy, cx, cy = (float(v) for v in intrinsics)
    intr = (C.c_double * 4)(fx, fy, cx, cy)
  _lib.check(_lib.load().spa3d_op_lift_2d_to_3d(tr.data_ptr(), dp.data_ptr(), N, T, dp.shape[1], dp.shape[2], intr, out.data_ptr(),
                                                _stream(tr)), what='spa3d_op_lift_2d_to_3d')
  return out


def sample_dino_features_for_tracks(dino_features, tracks_2d, video_shape, device='cuda', out_dtype=torch.float32):
  """dino_features [T,Hp,Wp,D], tracks_2d [N,T,2] (pixels), video_shape (T,H,W,3) -> [N,T,D] (inference.py:339-395).
  out_dtype=torch.bfloat16 writes the hot path's input dtype directly (same values, rounded once)."""
  if dino_features is None:
    return None
  ft, tr = _dev(dino_features, device), _dev(tracks_2d, device)
  T, Hp, Wp, D = ft.shape
  _, H, W, _ = (int(v) for v in video_shape)
  N = tr.shape[0]
  if tr.shape[1] != T:
    raise ValueError('tracks_2d and dino_features disagree on T')
  out = torch.empty(N, T, D, dtype=out_dtype, device=ft.device)
  _lib.check(_lib.load().spa3d_op_sample_dino(ft.data_ptr(), tr.data_ptr(), N, T, Hp, Wp, D, H, W, out.data_ptr(),
                                              _lib.F32 if out_dtype == torch.float32 else _lib.BF16, _stream(ft)),
             what='spa3d_op_sample_dino')
  return out


def sample_depth_features_for_tracks(depth, tracks_2d, device='cuda'):
  """depth [T,H,W,1], tracks_2d [N,T,2] -> [N,T,256]: depth, depth/10, temporal difference, zeros (inference.py:398-447)."""
  if depth is None:
    return None
  dp, tr = _dev(depth, device), _dev(tracks_2d, device)
  N, T = tr.shape[:2]
  if dp.dim() != 4 or dp.shape[0] != T or dp.shape[-1] != 1:
    raise ValueError(f'depth must be [T,H,W,1] with T={T}, got {tuple(dp.shape)}')
  out = torch.empty(N, T, 256, dtype=torch.float32, device=dp.device)
  _lib.check(_lib.load().spa3d_op_sample_depth_features(dp.data_ptr(), tr.data_ptr(), N, T, dp.shape[1], dp.shape[2], out.data_ptr(),
                                                        _stream(dp)), what='spa3d_op_sample_depth_features')
  return out
