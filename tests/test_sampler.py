"""Feature producers (inference.py:287-447).  PINNED: tests/golden/sampler_golden.npz holds the outputs
of the reference's OWN NumPy functions executed in the build container (tests/golden/make_sampler_golden.py).
CPU: the oracle restatement is bit-identical to them.  GPU: so are the HIP kernels (float32, same operation order)."""
import os

import numpy as np
import pytest
import torch

from oracle import sampler_oracle as S

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sampler_golden.npz'), allow_pickle=False)


def test_oracle_is_bit_identical_to_reference_outputs():
  tr, dp, dn = Z['tracks_2d'], Z['depth'], Z['dino']
  assert np.array_equal(S.lift_2d_to_3d(tr, dp), Z['lift_default'])
  assert np.array_equal(S.lift_2d_to_3d(tr, dp, tuple(Z['intrinsics'])), Z['lift_intr'])
  assert np.array_equal(S.sample_dino_features_for_tracks(dn, tr, tuple(Z['video_shape'])), Z['dino_tracks'])
  assert np.array_equal(S.sample_depth_features_for_tracks(dp, tr), Z['depth_tracks'])
  assert S.sample_dino_features_for_tracks(None, tr, tuple(Z['video_shape'])) is None  # inference.py:352-353


def test_oracle_edge_cases():
  # integer coordinates -> exact texel; far outside -> clamped texels with extrapolating weights (reference behaviour)
  depth = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4, 1)
  tr = np.array([[[1.0, 2.0], [3.0, 0.0]]], dtype=np.float32)
  out = S.sample_depth_features_for_tracks(depth, tr)
  assert out.shape == (1, 2, 256) and out[0, 0, 0] == depth[0, 2, 1, 0] and out[0, 1, 0] == depth[1, 0, 3, 0]
  assert out[0, 0, 2] == 0 and out[0, 1, 2] == out[0, 1, 0] - out[0, 0, 0] and np.all(out[..., 3:] == 0)
  assert out[0, 0, 1] == np.float32(out[0, 0, 0] / np.float32(10.0))


@pytest.mark.gpu
def test_hip_kernels_bit_identical_to_reference_outputs():
  import spa3d
  tr, dp, dn = Z['tracks_2d'], Z['depth'], Z['dino']
  assert np.array_equal(spa3d.lift_2d_to_3d(tr, dp).cpu().numpy(), Z['lift_default'])
  assert np.array_equal(spa3d.lift_2d_to_3d(tr, dp, tuple(Z['intrinsics'])).cpu().numpy(), Z['lift_intr'])
  assert np.array_equal(spa3d.sample_dino_features_for_tracks(dn, tr, tuple(Z['video_shape'])).cpu().numpy(), Z['dino_tracks'])
  assert np.array_equal(spa3d.sample_depth_features_for_tracks(dp, tr).cpu().numpy(), Z['depth_tracks'])
  b = spa3d.sample_dino_features_for_tracks(dn, tr, tuple(Z['video_shape']), out_dtype=torch.bfloat16)
  assert torch.equal(b.cpu(), torch.from_numpy(Z['dino_tracks']).bfloat16())


@pytest.mark.gpu
def test_hip_kernels_at_production_shape_match_oracle():
  """2048 tracks x 150 frames, DINOv2-base 37x37x768 patches of a 518x518 video, odd channel count variant"""
  import spa3d
  rng = np.random.default_rng(3)
  T, H, W, N = 12, 518, 518, 2048
  for D in (768, 30):
    dn = rng.standard_normal((T, 37, 37, D)).astype(np.float32)
    tr = np.stack([rng.random((N, T)) * (W + 20) - 10, rng.random((N, T)) * (H + 20) - 10], -1).astype(np.float32)
    got = spa3d.sample_dino_features_for_tracks(dn, tr, (T, H, W, 3)).cpu().numpy()
    assert np.array_equal(got, S.sample_dino_features_for_tracks(dn, tr, (T, H, W, 3)))
  dp = (rng.random((T, H, W, 1)) * 9 + 0.1).astype(np.float32)
  assert np.array_equal(spa3d.sample_depth_features_for_tracks(dp, tr).cpu().numpy(), S.sample_depth_features_for_tracks(dp, tr))
  assert np.array_equal(spa3d.lift_2d_to_3d(tr, dp).cpu().numpy(), S.lift_2d_to_3d(tr, dp))
