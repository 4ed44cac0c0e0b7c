"""The two independent CPU restatements must agree, and the torch oracle's autograd gradients
must agree with central finite differences (item 4) -- what stands in for reference-held fixtures (PARITY UNPINNED)."""
import numpy as np
import pytest
import torch

from util import MINI, O
from oracle import np_blocks as NB


def _np_tree(p):
  return {k: (_np_tree(v) if isinstance(v, dict) else v.double().numpy()) for k, v in p.items()}


def test_torch_and_numpy_restatements_agree_on_blocks():
  cfg = O.Config(**MINI, use_dino=False, use_depth=False)
  p = O.init_params(cfg, seed=3, dtype=torch.float64, with_dino=False, with_depth=False, perturb=0.2)
  g = torch.Generator().manual_seed(0)
  # self-attention transformer with a key mask (track encoder shape)
  x = torch.randn(9, cfg.track_token_dim, generator=g, dtype=torch.float64)
  km = torch.tensor([1, 1, 0, 1, 0, 1, 1, 0, 1], dtype=torch.float64)
  ref = O.transformer(p['input_track_transformer'], x[None], qq_mask=km[None, None, :].expand(1, 9, 9))[0]
  got = NB.transformer(_np_tree(p['input_track_transformer']), x.numpy(), None, km.numpy())
  assert np.abs(ref.numpy() - got).max() < 1e-12
  # cross-attention transformer (tracks_to_latents shape): K/V from un-normalised kv
  lat = torch.randn(cfg.num_latent_tokens, cfg.encoder_latent_dim, generator=g, dtype=torch.float64)
  kv = torch.randn(7, cfg.track_token_dim, generator=g, dtype=torch.float64)
  ref = O.transformer(p['tracks_to_latents'], lat[None], kv[None])[0]
  got = NB.transformer(_np_tree(p['tracks_to_latents']), lat.numpy(), kv.numpy(), None)
  assert np.abs(ref.numpy() - got).max() < 1e-12
  # sinusoidal embedding
  xs = torch.rand(5, 4, generator=g)
  assert np.abs(O.sinusoidal_embedding(xs.double()).numpy() - NB.sin_embed(xs.numpy())).max() < 1e-15


def test_oracle_gradients_match_finite_differences():
  cfg = O.Config(**MINI, use_dino=True, use_depth=True, dino_feature_dim=6, depth_feature_dim=2)
  B, N, Q, T = 2, 4, 3, 8
  batch = O.synthetic_batch(B, N, Q, T, dino_dim=6, depth_dim=2, dtype=torch.float64)
  batch['boundary_frame'] = torch.tensor([8, 5], dtype=torch.int32)
  p = O.init_params(cfg, seed=1, dtype=torch.float64, depth_dim=2, perturb=0.1)
  # keep every latent strictly inside (-1,1) and off the rounding grid so clip / round are locally smooth
  noise = torch.rand(B, cfg.num_latent_tokens, cfg.latent_token_dim, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
  m = O.TrackAutoEncoder3D(cfg)
  _, _, grads = O.loss_and_grads(m, p, batch, discretize=False)
  flat = O.tree_flatten(p)
  rng = np.random.default_rng(0)
  checked = 0
  for name in ['track_token_projection/kernel', 'dino_projection/kernel', 'depth_projection/bias', 'input_readout_token/state_init',
               'input_track_transformer/layer_1/self_att/norm_key/scale', 'input_track_transformer/layer_0/MLP_in/kernel',
               'tracks_to_latents/layer_1/cross_att/dense_value/kernel', 'initializer/state_init', 'compressor/kernel',
               'decompress_attn/layer_0/norm_attn/scale', 'query_encoder/kernel', 'track_readout_attn/layer_1/self_att/dense_out/bias',
               'track_readout_attn/norm_encoder/scale', 'track_predictor/kernel']:
    t = flat[name]
    for _ in range(2):
      idx = tuple(int(rng.integers(0, s)) for s in t.shape)
      old = float(t[idx])
      h = 1e-5 * max(1.0, abs(old))
      t[idx] = old + h
      lp = float(O.compute_loss_3d(m(O.tree_unflatten(flat), batch, discretize=False), batch)['total_loss'])
      t[idx] = old - h
      lm = float(O.compute_loss_3d(m(O.tree_unflatten(flat), batch, discretize=False), batch)['total_loss'])
      t[idx] = old
      fd = (lp - lm) / (2 * h)
      an = float(grads[name][idx])
      assert abs(fd - an) <= 2e-4 * max(1.0, abs(fd), abs(an)), (name, idx, fd, an)
      checked += 1
  assert checked == 28


def test_straight_through_and_clip_gradient():
  # 3d:251-260: d/dl [l - stop_grad(l - q)] = 1 inside the clip range, 0 outside; value = discretised + noise
  l = torch.tensor([0.3, 1.7, -2.0, -0.999], dtype=torch.float64, requires_grad=True)
  lc = torch.clamp(l, -1.0, 1.0)
  q = torch.round(lc * 128.0) / 128.0 + 0.25 / 128.0 - 1.0 / 256.0
  out = lc - (lc - q).detach()
  out.sum().backward()
  assert l.grad.tolist() == [1.0, 0.0, 0.0, 1.0]
  assert torch.allclose(out.detach(), q)


def test_scan_chunked_decode_is_identical():
  cfg = O.Config(**MINI, use_dino=False, use_depth=False)
  batch = O.synthetic_batch(2, 4, 6, 8, dtype=torch.float64)
  p = O.init_params(cfg, seed=2, dtype=torch.float64, with_dino=False, with_depth=False)
  a = O.TrackAutoEncoder3D(cfg)(p, batch, discretize=False)
  cfg2 = O.Config(**MINI, use_dino=False, use_depth=False, decoder_scan_chunk_size=2)
  b = O.TrackAutoEncoder3D(cfg2)(p, batch, discretize=False)
  assert torch.allclose(a.tracks, b.tracks, atol=1e-13) and torch.allclose(a.visible_logits, b.visible_logits, atol=1e-13)
