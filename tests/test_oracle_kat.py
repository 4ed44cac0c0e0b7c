"""Closed-form known-answer tests that pin the CPU oracle.  The reference holds no
tests or fixtures for this path, so these KATs -- derived from the reference source lines cited in
each test and from the documented Flax/JAX/Optax semantics -- are what the oracle is anchored on."""
import math

import numpy as np
import pytest
import torch

from util import O
from oracle import np_blocks as NB


def test_sin_embed_zero_and_layout():
  # track_autoencoder.py:28-37: x=0 -> sin(0)=0 for the first 32, sin(fl32(pi/2)) for the next 32, per coordinate
  e = O.sinusoidal_embedding(torch.zeros(2, 3))
  assert e.shape == (2, 192)
  v = e.view(2, 3, 64)
  assert torch.all(v[..., :32] == 0)
  assert torch.allclose(v[..., 32:], torch.ones(2, 3, 32), atol=1e-7)
  # "(coords d)" layout: coordinate outermost; channel c*64+f = sin(x_c * 2^(f/3))
  x = torch.tensor([[0.25, -0.5, 0.75]])
  e = O.sinusoidal_embedding(x.double())[0]
  for c in range(3):
    for f in (0, 1, 7, 31):
      s = np.float32(2 ** (f / 3))
      arg = np.float32(np.float32(x[0, c]) * s)
      assert abs(float(e[c * 64 + f]) - math.sin(float(arg))) < 1e-12
      arg2 = np.float32(arg + np.float32(0.5 * np.pi))  # cos is sin(v + pi/2) in fp32 -- never cos()
      assert abs(float(e[c * 64 + 32 + f]) - math.sin(float(arg2))) < 1e-12


def test_sin_scales_are_float32_of_python_doubles():
  s = O.sin_scales(32)
  assert s.dtype == np.float32 and s[0] == 1.0 and s[3] == 2.0 and abs(s[31] - 2 ** (31 / 3)) < 1e-3 * 2 ** (31 / 3)


@pytest.mark.parametrize('t', [0, 7, 149, 204, 205, 299])
def test_append_time_feat_eye_einsum_is_a_window_gather(t):
  # track_autoencoder_3d.py:235-246: eye(128, C, 5*idx) einsum == lat[n, 5t : 5t+128] with zero fill past C
  cfg = O.Config()
  m = O.TrackAutoEncoder3D(cfg)
  g = torch.Generator().manual_seed(0)
  lat = torch.randn(1, 2, 4, 1152, generator=g, dtype=torch.float64)
  qf = torch.tensor([[t, 3]], dtype=torch.int32)
  out = m.append_time_feat(lat, qf)
  assert out.shape == (1, 2, 4, 1280)
  assert torch.equal(out[..., :1152], lat)
  for q, tt in enumerate((t, 3)):
    want = torch.zeros(4, 128, dtype=torch.float64)
    lo, hi = 5 * tt, min(5 * tt + 128, 1152)
    if hi > lo:
      want[:, :hi - lo] = lat[0, q, :, lo:hi]
    assert torch.equal(out[0, q, :, 1152:], want)


@pytest.mark.parametrize('t,want', [(0, 0.0), (149, 0.0), (150, 1.0), (299, 1.0)])
def test_query_time_channel_is_a_floor_division(t, want):
  # track_autoencoder_3d.py:268-269: query_frame // 150.0
  assert float(torch.floor(torch.tensor(float(t)) / 150.0)) == want


def test_round_is_half_to_even_like_jnp_round():
  assert torch.round(torch.tensor([0.5, 1.5, 2.5, -0.5])).tolist() == [0.0, 2.0, 2.0, -0.0]


def test_layernorm_rmsnorm_kats():
  one = torch.ones(8, dtype=torch.float64)
  # constant vector: variance 0 -> output 0 (fast variance clamps at 0)
  assert torch.all(O.layer_norm(3.0 * one, one) == 0)
  # one-hot e0 * a (d=4): mean a/4, var 3a^2/16
  a = 2.0
  x = torch.tensor([a, 0, 0, 0], dtype=torch.float64)
  y = O.layer_norm(x, torch.ones(4, dtype=torch.float64))
  r = 1 / math.sqrt(3 * a * a / 16 + 1e-6)
  assert torch.allclose(y, torch.tensor([0.75 * a * r, -0.25 * a * r, -0.25 * a * r, -0.25 * a * r], dtype=torch.float64), atol=1e-14)
  # RMSNorm of a constant vector c: c / sqrt(c^2 + eps) * scale
  y = O.rms_norm(3.0 * one, 2.0 * one)
  assert torch.allclose(y, 2.0 * 3.0 / math.sqrt(9 + 1e-6) * one, atol=1e-14)


def test_gelu_tanh_kats():
  def g(x):
    return 0.5 * x * (1 + math.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))
  xs = torch.tensor([0.0, 1.0, -1.0, 3.0, -3.0], dtype=torch.float64)
  assert torch.allclose(O.gelu_tanh(xs), torch.tensor([g(float(x)) for x in xs], dtype=torch.float64), atol=1e-15)
  assert abs(float(O.gelu_tanh(torch.tensor(1.0, dtype=torch.float64))) - 0.8411919906082768) < 1e-12


def test_softmax_all_masked_row_is_uniform_and_masked_keys_get_zero():
  q = torch.randn(1, 3, 1, 4, dtype=torch.float64)
  k = torch.randn(1, 5, 1, 4, dtype=torch.float64)
  v = torch.eye(5, dtype=torch.float64)[None, :, None, :4]
  all_masked = torch.zeros(1, 1, 3, 5)
  out = O.dot_product_attention(q, k, v, all_masked)  # finfo.min fill -> uniform weights
  assert torch.allclose(out, v.mean(1, keepdim=True).expand(1, 3, 1, 4), atol=1e-12)
  only0 = torch.zeros(1, 1, 3, 5)
  only0[..., 0] = 1
  out = O.dot_product_attention(q, k, v, only0)
  assert torch.allclose(out, v[:, 0:1].expand(1, 3, 1, 4), atol=1e-12)


def test_bce_and_loss_normalisers():
  # optax.sigmoid_binary_cross_entropy at logits {0, +-20}
  l = torch.tensor([0.0, 20.0, -20.0], dtype=torch.float64)
  assert torch.allclose(O.sigmoid_binary_cross_entropy(l, torch.ones(3, dtype=torch.float64)),
                        torch.tensor([math.log(2), math.log1p(math.exp(-20)), 20 + math.log1p(math.exp(-20))], dtype=torch.float64))
  # train.py:96-129: both terms are divided by max(sum(visible),1) of the WHOLE batch; BCE sums over all elements
  B, Q, T = 2, 3, 4
  pred = O.Results(torch.zeros(B, Q, T, 3, dtype=torch.float64), torch.zeros(B, Q, T, 1, dtype=torch.float64),
                   torch.zeros(B, Q, T, 1, dtype=torch.float64))
  tgt = {'query_tracks': torch.ones(B, Q, T, 3, dtype=torch.float64), 'query_tracks_visible': torch.zeros(B, Q, T, 1, dtype=torch.float64)}
  tgt['query_tracks_visible'][0, 0, :2] = 1  # 2 visible points
  ld = O.compute_loss_3d(pred, tgt)
  assert abs(float(ld['position_loss']) - (2 * 3) / 2) < 1e-12
  assert abs(float(ld['visible_loss']) - (B * Q * T) * math.log(2) / 2) < 1e-12
  assert abs(float(ld['total_loss']) - (5000 * 3.0 + 1e-8 * B * Q * T * math.log(2) / 2)) < 1e-9
  tgt['query_tracks_visible'].zero_()  # denominator clamps at 1
  assert float(O.compute_loss_3d(pred, tgt)['position_loss']) == 0.0


def test_threefry2x32_random123_vectors():
  # public Random123 known-answer vectors: the block function under jax.random.uniform(PRNGKey(0))
  f = lambda k, c: tuple(int(x) for x in NB.threefry2x32(k, c))
  assert f((0, 0), (0, 0)) == (0x6b200159, 0x99ba4efe)
  assert f((0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff)) == (0x1cb996fc, 0xbb002be7)
  assert f((0x13198a2e, 0x03707344), (0x243f6a88, 0x85a308d3)) == (0xc4923a9c, 0x483df7a0)
  u = NB.jax_uniform_legacy((7, 5))
  assert u.shape == (7, 5) and u.dtype == np.float32 and float(u.min()) >= 0.0 and float(u.max()) < 1.0


def test_param_tree_matches_survey_counts():
  # 108 744 312 parameters xyz-only at T_out=150; +296 064 with repaired dino(768)/depth(1) projections
  n0 = sum(v.numel() for v in O.tree_flatten(O.init_params(O.Config(), with_dino=False, with_depth=False)).values())
  assert n0 == 108744312
  cfg = O.Config(**{k: v for k, v in dict(num_output_frames=150).items()})
  p = O.tree_flatten(O.init_params(cfg, depth_dim=1))
  assert sum(v.numel() for v in p.values()) == 108744312 + 296064
  assert tuple(p['query_encoder/kernel'].shape) == (12352, 1280)
  assert tuple(p['tracks_to_latents/layer_0/cross_att/dense_key/kernel'].shape) == (384, 8, 96)
  assert tuple(p['track_readout_attn/layer_3/self_att/dense_out/kernel'].shape) == (8, 96, 1280)


def test_lr_schedule_and_adamw_semantics():
  # train.py:41-57: linear 0->lr over warmup, then cosine to 0; first update uses lr(0) = 0
  assert O.lr_schedule(0, 1e-4, 10, 110) == 0.0
  assert abs(O.lr_schedule(5, 1e-4, 10, 110) - 5e-5) < 1e-18
  assert abs(O.lr_schedule(10, 1e-4, 10, 110) - 1e-4) < 1e-18
  assert abs(O.lr_schedule(60, 1e-4, 10, 110) - 5e-5) < 1e-12
  assert O.lr_schedule(110, 1e-4, 10, 110) < 1e-20
  # one AdamW step from zero moments: update = -lr*(sign-like g/(|g|+eps) + wd*p); clip only above norm 1
  P, G = {'a': torch.tensor([1.0, -2.0], dtype=torch.float64)}, {'a': torch.tensor([0.3, -0.4], dtype=torch.float64)}
  M, V = {'a': torch.zeros(2, dtype=torch.float64)}, {'a': torch.zeros(2, dtype=torch.float64)}
  gn = O.adamw_step(P, G, M, V, step=0, lr=0.1)
  assert abs(gn - 0.5) < 1e-12
  want = torch.tensor([1.0 - 0.1 * (0.3 / (0.3 + 1e-8) + 0.01 * 1.0), -2.0 - 0.1 * (-0.4 / (0.4 + 1e-8) + 0.01 * -2.0)], dtype=torch.float64)
  assert torch.allclose(P['a'], want, atol=1e-12)
