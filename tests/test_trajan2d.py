"""The 2-D TRAJAN twin (track_autoencoder.py:117-390) behind the same kernels:
(x,y) tracks, no readout token, key mask from visibility & boundary only, visible-mean pooling, certainty head.
It validates every shared block against reference code that needs no repair.  PARITY UNPINNED (oracle = this repo's
restatement; the reference needs JAX/Flax to run)."""
import pytest
import torch

from util import O, batch_to, max_abs, rel_err

MINI2D = dict(num_output_frames=8, num_latent_tokens=8, latent_token_dim=16, num_frequencies=4, track_token_dim=32, encoder_latent_dim=48,
              decoder_num_channels=192, num_heads=2, qkv_size=32, enc_mlp=64, enc_layers=2, t2l_mlp=64, t2l_layers=2, dec_mlp=64,
              dec_layers=1, ro_mlp=64, ro_layers=2)


@pytest.fixture(scope='module')
def spa3d():
  import spa3d as s
  return s


def _model(spa3d, cfg, precision):
  m = spa3d.TrackAutoEncoder(num_output_frames=cfg.num_output_frames, num_latent_tokens=cfg.num_latent_tokens,
                             latent_token_dim=cfg.latent_token_dim, num_frequencies=cfg.num_frequencies,
                             track_token_dim=cfg.track_token_dim, encoder_latent_dim=cfg.encoder_latent_dim,
                             decoder_num_channels=cfg.decoder_num_channels, precision=precision)
  m.num_heads, m.qkv_size = cfg.num_heads, cfg.qkv_size
  m.enc_mlp, m.enc_layers, m.t2l_mlp, m.t2l_layers = cfg.enc_mlp, cfg.enc_layers, cfg.t2l_mlp, cfg.t2l_layers
  m.dec_mlp, m.dec_layers, m.ro_mlp, m.ro_layers = cfg.dec_mlp, cfg.dec_layers, cfg.ro_mlp, cfg.ro_layers
  return m


def test_2d_parameter_tree_matches_flax_names_and_shapes(spa3d):
  m = spa3d.TrackAutoEncoder(precision='bf16')
  _, leaves, n = m._handle(0, 0)
  ref = O.tree_flatten(O.init_params_2d(O.config_2d()))
  got = {name: shape for name, shape, _ in leaves}
  assert set(got) == set(ref) and 'input_readout_token/state_init' not in got
  for k, v in ref.items():
    assert tuple(v.shape) == tuple(got[k]), k
  assert tuple(got['track_token_projection/kernel']) == (192, 256) and tuple(got['query_encoder/kernel']) == (8256, 1024)
  assert tuple(got['input_track_transformer/layer_0/self_att/dense_query/kernel']) == (256, 8, 64)


@pytest.mark.gpu
def test_2d_forward_loss_grads_fp32(spa3d):
  cfg = O.config_2d(**MINI2D)
  B, N, Q, T = 3, 7, 5, 8
  batch = O.synthetic_batch_2d(B, N, Q, T, seed=5)
  batch['boundary_frame'] = torch.tensor([8, 6, 2], dtype=torch.int32)
  batch['support_tracks_visible'][1, 2] = 0  # a track that is never visible: all-masked attention rows + empty pooling
  gb = batch_to(batch, 'cuda')
  model = _model(spa3d, cfg, 'fp32')
  params = model.init(3, gb)['params']
  g = torch.Generator().manual_seed(0)
  for k, v in O.tree_flatten(params).items():
    if k.endswith('bias') or k.endswith('scale'):
      v.add_((0.1 * torch.randn(v.shape, generator=g)).to(v.device))
  noise = torch.rand(B, cfg.num_latent_tokens, cfg.latent_token_dim, generator=g)
  p64 = O.tree_unflatten({k: v.detach().cpu().double() for k, v in O.tree_flatten(params).items()})
  b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
  om = O.TrackAutoEncoder2D(cfg)
  ld_ref, preds_ref, grads_ref = O.loss_and_grads(om, p64, b64, noise=noise.double())
  ld, grads, preds = model.loss_and_grads({'params': params}, gb, noise=noise.cuda(), return_predictions=True)
  assert preds.tracks.shape == (B, Q, T, 2)
  assert max_abs(preds.tracks, preds_ref.tracks) < 1e-4
  assert max_abs(preds.visible_logits, preds_ref.visible_logits) < 1e-4
  assert max_abs(preds.certain_logits, preds_ref.certain_logits) < 1e-4 and float(preds.certain_logits.abs().max()) > 0
  for k in ('total_loss', 'position_loss', 'visible_loss'):
    assert abs(float(ld[k]) - float(ld_ref[k])) <= 1e-5 * abs(float(ld_ref[k])) + 1e-7, k
  l2 = spa3d.compute_loss_2d(preds, gb)
  assert abs(float(l2['total_loss']) - float(ld_ref['total_loss'])) <= 1e-5 * abs(float(ld_ref['total_loss']))
  gf = O.tree_flatten(grads)
  assert set(gf) == set(grads_ref)
  worst = 0.0
  for k, gref in grads_ref.items():
    e = rel_err(gf[k], gref) if float(gref.norm()) > 1e-12 else float(gf[k].abs().max())
    worst = max(worst, e)
    assert e < 2e-3, (k, e)
  print('2d mini fp32 worst grad rel err', worst)
  # default grid (ta:256-266) and encode/decode split
  nb = {k: v for k, v in gb.items() if k != 'query_points'}
  pg = model.apply({'params': params}, nb, discretize=False)
  rg = om(p64, {k: v for k, v in b64.items() if k != 'query_points'}, discretize=False)
  assert pg.tracks.shape == (B, 1024, T, 2) and max_abs(pg.tracks, rg.tracks) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize('Q', [16, 64])
def test_2d_default_size_bf16_runs_and_tracks_fp32(spa3d, Q):
  """default TRAJAN sizes (d=256, heads of 64, 2/6/3/4 layers): bf16 path vs the fp32 path of the same library.  Q = 64 over 24 frames:
  the bf16 run takes the shared-latent-row path of the readout stack's first block (several queries per frame), the fp32 run the dense one."""
  cfg = O.config_2d(num_output_frames=24)
  B, N, T = 2, 48, 24
  batch = O.synthetic_batch_2d(B, N, Q, T)
  gb = batch_to(batch, 'cuda')
  noise = torch.rand(B, 128, 64).cuda()
  mb = spa3d.TrackAutoEncoder(num_output_frames=24, precision='bf16')
  params = mb.init(0, gb)['params']
  ld_b, g_b, p_b = mb.loss_and_grads({'params': params}, gb, noise=noise, return_predictions=True)
  gbf = g_b.flat.clone()
  mf = spa3d.TrackAutoEncoder(num_output_frames=24, precision='fp32')
  ld_f, g_f, p_f = mf.loss_and_grads({'params': params}, gb, noise=noise, return_predictions=True)
  assert rel_err(p_b.tracks, p_f.tracks) < 5e-2
  cos = float((gbf.double() @ g_f.flat.double()) / (gbf.double().norm() * g_f.flat.double().norm()))
  print('2d bf16 vs fp32: tracks rel', rel_err(p_b.tracks, p_f.tracks), 'grad cosine', cos)
  assert cos > 0.97
