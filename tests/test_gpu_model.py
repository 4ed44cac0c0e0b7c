"""GPU parity of the whole hot path behind the reference API (init / apply / encode / decode / compute_loss_3d /
value_and_grad / train step) against the CPU oracle, through the C-ABI.

Tolerances (north_star: "within 1e-4 fp32"):
  fp32 path : outputs / latents max-abs <= 1e-4 vs the oracle run in fp32 (same op order as the reference);
              losses relative 1e-5; every gradient leaf relative Frobenius error <= 2e-3 vs the fp64 oracle
              (fp32 activations through 11 blocks; measured values are printed).
  bf16 path : outputs rel-Frobenius <= 5e-2, losses relative 5e-2 (bf16 activations end to end) -- a throughput
              mode, stated separately as SURVEY hard-part (4) requires.
PARITY UNPINNED: the oracle is this repo's restatement (see oracle/spa3d_oracle.py header)."""
import numpy as np
import pytest
import torch

from util import MINI, O, batch_to, max_abs, product_model, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def spa3d():
  import spa3d as s
  return s


def _params_to_oracle(params, dtype):
  flat = O.tree_flatten(params)
  return O.tree_unflatten({k: v.detach().cpu().to(dtype) for k, v in flat.items()})


def _perturb(params, seed=0, amt=0.1):
  """zero-initialised biases / unit scales would hide bugs in their paths"""
  g = torch.Generator().manual_seed(seed)
  for k, v in O.tree_flatten(params).items():
    if k.endswith('bias') or k.endswith('scale'):
      v.add_((amt * torch.randn(v.shape, generator=g)).to(v.device))


def _setup(spa3d, cfg, B, N, Q, T, precision, dino=0, depth=0, seed=0):
  batch = O.synthetic_batch(B, N, Q, T, seed=1234, dino_dim=dino, depth_dim=depth)
  if cfg.num_output_frames != T:
    raise AssertionError('tests use T_out == T')
  model = product_model(spa3d, cfg, precision)
  gb = batch_to(batch, 'cuda')
  params = model.init(seed, gb)['params']
  _perturb(params)
  return model, params, batch, gb


def _noise(B, cfg, seed=3):
  return torch.rand(B, cfg.num_latent_tokens, cfg.latent_token_dim, generator=torch.Generator().manual_seed(seed))


# ---------------------------------------------------------------------------------------------- mini config, all features
@pytest.mark.parametrize('dino,depth', [(0, 0), (24, 1), (16, 5)])
def test_mini_forward_loss_grads_fp32(spa3d, dino, depth):
  cfg = O.Config(**MINI, use_dino=dino > 0, use_depth=depth > 0, dino_feature_dim=max(dino, 1), depth_feature_dim=max(depth, 1))
  B, N, Q, T = 3, 10, 6, 8
  model, params, batch, gb = _setup(spa3d, cfg, B, N, Q, T, 'fp32', dino, depth)
  batch['boundary_frame'] = torch.tensor([8, 5, 3], dtype=torch.int32)
  gb['boundary_frame'] = batch['boundary_frame'].cuda()
  noise = _noise(B, cfg)
  om = O.TrackAutoEncoder3D(cfg)
  p64 = _params_to_oracle(params, torch.float64)
  b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
  ld_ref, preds_ref, grads_ref = O.loss_and_grads(om, p64, b64, discretize=True, noise=noise.double())
  preds = model.apply({'params': params}, gb, discretize=True, noise=noise.cuda())
  assert max_abs(preds.tracks, preds_ref.tracks) < 1e-4
  assert max_abs(preds.visible_logits, preds_ref.visible_logits) < 1e-4
  assert float(preds.certain_logits.abs().max()) == 0.0
  ld = spa3d.compute_loss_3d(preds, gb)
  for k in ('total_loss', 'position_loss', 'visible_loss'):
    assert abs(float(ld[k]) - float(ld_ref[k])) <= 1e-5 * abs(float(ld_ref[k])) + 1e-7, k
  ld2, grads, preds2 = model.loss_and_grads({'params': params}, gb, discretize=True, noise=noise.cuda(), return_predictions=True)
  assert max_abs(preds2.tracks, preds.tracks) == 0.0
  for k in ('total_loss', 'position_loss', 'visible_loss'):
    assert abs(float(ld2[k]) - float(ld_ref[k])) <= 1e-5 * abs(float(ld_ref[k])) + 1e-7, k
  gflat = O.tree_flatten(grads)
  assert set(gflat) == set(grads_ref)
  worst = 0.0
  for k, gref in grads_ref.items():
    e = rel_err(gflat[k], gref) if float(gref.norm()) > 1e-12 else float(gflat[k].abs().max())
    worst = max(worst, e)
    assert e < 2e-3, (k, e)
  print('mini fp32 worst grad rel err', worst)


def test_mini_chunking_and_api_equivalences(spa3d, monkeypatch):
  cfg = O.Config(**MINI, use_dino=False, use_depth=False)
  B, N, Q, T = 4, 6, 5, 8
  model, params, batch, gb = _setup(spa3d, cfg, B, N, Q, T, 'fp32')
  noise = _noise(B, cfg).cuda()
  ld, grads, preds = model.loss_and_grads({'params': params}, gb, noise=noise, return_predictions=True)
  g_all = grads.flat.clone()
  monkeypatch.setenv('SPA3D_CHUNK', '1')
  ld1, grads1, preds1 = model.loss_and_grads({'params': params}, gb, noise=noise, return_predictions=True)
  monkeypatch.delenv('SPA3D_CHUNK')
  assert max_abs(preds1.tracks, preds.tracks) < 1e-6  # same arithmetic per sample
  assert rel_err(grads1.flat, g_all) < 1e-5  # only the accumulation order differs
  assert abs(float(ld1['total_loss']) - float(ld['total_loss'])) < 1e-5 * abs(float(ld['total_loss']))
  # encode -> decode == __call__ ; scan-chunked decode (3d:312-349) is the same numbers
  lat = model.apply({'params': params}, gb, method=model.encode)
  ctx = model.apply({'params': params}, gb, method=model.get_decoder_context)
  dec = model.apply({'params': params}, lat, ctx, method=model.decode, noise=noise)
  assert max_abs(dec.tracks, preds.tracks) < 1e-6
  ref_lat = O.TrackAutoEncoder3D(cfg).encode(_params_to_oracle(params, torch.float64),
                                              {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()})
  assert max_abs(lat, ref_lat) < 1e-4
  # discretize=False path
  p0 = model.apply({'params': params}, gb, discretize=False)
  r0 = O.TrackAutoEncoder3D(cfg)(_params_to_oracle(params, torch.float64),
                                 {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}, discretize=False)
  assert max_abs(p0.tracks, r0.tracks) < 1e-4
  # default 32x32 query grid when query_points is absent (3d:215-226)
  nb = {k: v for k, v in gb.items() if k != 'query_points'}
  pg = model.apply({'params': params}, nb, discretize=False)
  cb = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items() if k != 'query_points'}
  rg = O.TrackAutoEncoder3D(cfg)(_params_to_oracle(params, torch.float64), cb, discretize=False)
  assert pg.tracks.shape == (B, 1024, T, 3) and max_abs(pg.tracks, rg.tracks) < 1e-4


def test_mini_bf16_mode(spa3d):
  cfg = O.Config(**MINI, use_dino=True, use_depth=True, dino_feature_dim=24, depth_feature_dim=1)
  B, N, Q, T = 2, 12, 6, 8
  model, params, batch, gb = _setup(spa3d, cfg, B, N, Q, T, 'bf16', 24, 1)
  noise = _noise(B, cfg)
  om = O.TrackAutoEncoder3D(cfg)
  p64 = _params_to_oracle(params, torch.float64)
  b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
  b64['dino_features'] = batch['dino_features'].bfloat16().double()
  b64['depth_features'] = batch['depth_features'].bfloat16().double()
  ld_ref, preds_ref, grads_ref = O.loss_and_grads(om, p64, b64, discretize=True, noise=noise.double())
  ld, grads, preds = model.loss_and_grads({'params': params}, gb, noise=noise.cuda(), return_predictions=True)
  assert rel_err(preds.tracks, preds_ref.tracks) < 5e-2
  assert abs(float(ld['total_loss']) - float(ld_ref['total_loss'])) < 5e-2 * abs(float(ld_ref['total_loss']))
  # gradient direction: cosine similarity of the whole flat gradient (L1-loss sign flips make leaf-wise checks brittle in bf16)
  gf = O.tree_flatten(grads)
  a = torch.cat([gf[k].double().cpu().reshape(-1) for k in sorted(grads_ref)])
  b = torch.cat([grads_ref[k].reshape(-1) for k in sorted(grads_ref)])
  cos = float((a @ b) / (a.norm() * b.norm()))
  print('bf16 mini grad cosine', cos)
  assert cos > 0.98


# ---------------------------------------------------------------------------------------------- BASELINE cfg#1 (full-size model)
def test_cfg1_fp32_parity_1e4(spa3d):
  """BASELINE.json configs[0]: B=2, 64 support + 16 query, T=24, xyz-only, full-size model, 1 fwd+bwd step."""
  cfg = O.Config(num_output_frames=24, use_dino=False, use_depth=False)
  B, N, Q, T = 2, 64, 16, 24
  model, params, batch, gb = _setup(spa3d, cfg, B, N, Q, T, 'fp32')
  noise = _noise(B, cfg)
  om = O.TrackAutoEncoder3D(cfg)
  p32 = _params_to_oracle(params, torch.float32)
  ld_ref, preds_ref, grads_ref = O.loss_and_grads(om, p32, batch, discretize=True, noise=noise)
  ld, grads, preds = model.loss_and_grads({'params': params}, gb, noise=noise.cuda(), return_predictions=True)
  e_t, e_v = max_abs(preds.tracks, preds_ref.tracks), max_abs(preds.visible_logits, preds_ref.visible_logits)
  print('cfg1 fp32 max abs err tracks', e_t, 'logits', e_v)
  assert e_t < 1e-4 and e_v < 1e-4
  for k in ('total_loss', 'position_loss', 'visible_loss'):
    assert abs(float(ld[k]) - float(ld_ref[k])) <= 2e-5 * abs(float(ld_ref[k])), k
  gf = O.tree_flatten(grads)
  worst = ('', 0.0)
  for k, gref in grads_ref.items():
    e = rel_err(gf[k], gref)
    if e > worst[1]:
      worst = (k, e)
  print('cfg1 fp32 worst grad leaf', worst)
  # fp32-vs-fp32: two different summation orders through 11 blocks; L1 sign(pred-tgt) is discontinuous
  assert worst[1] < 5e-3


def test_train_step_matches_oracle_adamw(spa3d):
  cfg = O.Config(**MINI, use_dino=False, use_depth=False)
  B, N, Q, T = 2, 6, 4, 8
  model, params, batch, gb = _setup(spa3d, cfg, B, N, Q, T, 'fp32')
  noise = _noise(B, cfg)
  st = spa3d.TrainState(model, params, learning_rate=1e-2, warmup_steps=2, total_steps=10)
  om = O.TrackAutoEncoder3D(cfg)
  P = _params_to_oracle(params, torch.float64)
  b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
  flatP = O.tree_flatten(P)
  M = {k: torch.zeros_like(v) for k, v in flatP.items()}
  V = {k: torch.zeros_like(v) for k, v in flatP.items()}
  for step in range(3):
    mt = st.train_step(gb, noise=noise.cuda())
    ld, _, g = O.loss_and_grads(om, O.tree_unflatten(flatP), b64, noise=noise.double())
    lr = O.lr_schedule(step, 1e-2, 2, 10)
    gn = O.adamw_step(flatP, g, M, V, step, lr)
    assert abs(mt['train/learning_rate'] - lr) < 1e-12
    assert abs(float(mt['train/loss']) - float(ld['total_loss'])) < 1e-4 * abs(float(ld['total_loss']))
    assert abs(float(mt['train/grad_norm']) - gn) < 1e-3 * gn
  got = O.tree_flatten(st.params)
  for k, v in flatP.items():
    assert max_abs(got[k], v) < 2e-4, k  # three Adam steps at lr 1e-2: sign-like updates amplify tiny grad differences


def test_eval_step_matches_oracle_and_leaves_the_state_alone(spa3d):
  """TrainState.eval_step (train.py:189-213): forward + compute_loss_3d on the current parameters, the reference's 'eval/...' metric
  keys, predictions returned, nothing updated."""
  cfg = O.Config(**MINI, use_dino=False, use_depth=False)
  B, N, Q, T = 2, 6, 4, 8
  model, params, batch, gb = _setup(spa3d, cfg, B, N, Q, T, 'fp32')
  noise = _noise(B, cfg)
  st = spa3d.TrainState(model, params, learning_rate=1e-2, warmup_steps=2, total_steps=10)
  before = st.flat.clone()
  metrics, preds = st.eval_step(gb, noise=noise.cuda())
  assert sorted(metrics) == ['eval/loss', 'eval/position_loss', 'eval/visible_loss']
  om = O.TrackAutoEncoder3D(cfg)
  P = _params_to_oracle(params, torch.float64)
  b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
  ref = om(P, b64, noise=noise.double())
  ld = O.compute_loss_3d(ref, b64)
  for k, r in (('eval/loss', 'total_loss'), ('eval/position_loss', 'position_loss'), ('eval/visible_loss', 'visible_loss')):
    assert abs(float(metrics[k]) - float(ld[r])) < 1e-4 * abs(float(ld[r])), k
  assert max_abs(preds.tracks, ref.tracks) < 1e-4 and max_abs(preds.visible_logits, ref.visible_logits) < 1e-4
  assert torch.equal(st.flat, before) and st.step == 0
  mt = st.train_step(gb, noise=noise.cuda())   # the same loss before the first update
  assert abs(float(mt['train/loss']) - float(metrics['eval/loss'])) < 1e-5 * abs(float(metrics['eval/loss']))


def test_cfg1_bf16_tiled_vs_generic_and_fp32(spa3d, monkeypatch):
  """Full-size model at cfg#1 shapes in bf16: the tiled MFMA GEMMs / fused kernels (default) against the generic
  kernels (SPA3D_GEMM_IMPL=1, SPA3D_ATTN_IMPL=1: same bf16 arithmetic, different accumulation order) and against
  the fp32 path that test_cfg1_fp32_parity_1e4 pins to the oracle."""
  cfg = O.Config(num_output_frames=24, use_dino=True, use_depth=True, dino_feature_dim=768, depth_feature_dim=1)
  B, N, Q, T = 2, 64, 16, 24
  batch = O.synthetic_batch(B, N, Q, T, seed=1234, dino_dim=768, depth_dim=1)
  gb = batch_to(batch, 'cuda')
  gb['dino_features'] = gb['dino_features'].bfloat16()
  gb['depth_features'] = gb['depth_features'].bfloat16()
  noise = _noise(B, cfg).cuda()
  m_fast = product_model(spa3d, cfg, 'bf16')
  params = m_fast.init(0, gb)['params']
  _perturb(params)
  ld_f, g_f, p_f = m_fast.loss_and_grads({'params': params}, gb, noise=noise, return_predictions=True)
  g_fast = g_f.flat.clone()
  monkeypatch.setenv('SPA3D_GEMM_IMPL', '1')
  monkeypatch.setenv('SPA3D_ATTN_IMPL', '1')
  m_gen = product_model(spa3d, cfg, 'bf16')
  ld_g, g_g, p_g = m_gen.loss_and_grads({'params': params}, gb, noise=noise, return_predictions=True)
  monkeypatch.delenv('SPA3D_GEMM_IMPL')
  monkeypatch.delenv('SPA3D_ATTN_IMPL')
  e = rel_err(p_f.tracks, p_g.tracks)
  cos = float((g_fast.double() @ g_g.flat.double()) / (g_fast.double().norm() * g_g.flat.double().norm()))
  print('bf16 tiled vs generic: tracks rel', e, 'grad cosine', cos)
  assert e < 2e-2 and cos > 0.995
  fa, fb = O.tree_flatten(g_f), O.tree_flatten(g_g)
  worst = max((rel_err(fa[k].float(), fb[k].float()), k) for k in fb if float(fb[k].float().norm()) > 1e-12)
  print('bf16 tiled vs generic: worst gradient leaf', worst)
  assert worst[0] < 0.25
  m32 = product_model(spa3d, cfg, 'fp32')
  gb32 = dict(gb)
  gb32['dino_features'] = gb['dino_features'].float()
  gb32['depth_features'] = gb['depth_features'].float()
  ld_32, g_32, p_32 = m32.loss_and_grads({'params': params}, gb32, noise=noise, return_predictions=True)
  e32 = rel_err(p_f.tracks, p_32.tracks)
  cos32 = float((g_fast.double() @ g_32.flat.double()) / (g_fast.double().norm() * g_32.flat.double().norm()))
  print('bf16 tiled vs fp32: tracks rel', e32, 'loss', float(ld_f['total_loss']), float(ld_32['total_loss']), 'grad cosine', cos32)
  assert e32 < 5e-2 and cos32 > 0.97
  assert abs(float(ld_f['total_loss']) - float(ld_32['total_loss'])) < 5e-2 * abs(float(ld_32['total_loss']))


@pytest.mark.gpu
def test_cfg1_bf16_forced_8phase_kernels_vs_generic(spa3d, monkeypatch):
  """Same graph with the 8-phase NT / TN kernels forced on for every eligible GEMM (SPA3D_GEMM_IMPL=4: small and
  ragged M, the row remaps of the pruned blocks, all tile configurations) against the generic kernels."""
  cfg = O.Config(num_output_frames=24, use_dino=True, use_depth=True, dino_feature_dim=768, depth_feature_dim=1)
  B, N, Q, T = 2, 64, 16, 24
  batch = O.synthetic_batch(B, N, Q, T, seed=4321, dino_dim=768, depth_dim=1)
  gb = batch_to(batch, 'cuda')
  gb['dino_features'] = gb['dino_features'].bfloat16()
  gb['depth_features'] = gb['depth_features'].bfloat16()
  noise = _noise(B, cfg).cuda()
  monkeypatch.setenv('SPA3D_GEMM_IMPL', '4')
  m_8p = product_model(spa3d, cfg, 'bf16')
  params = m_8p.init(0, gb)['params']
  _perturb(params)
  ld_f, g_f, p_f = m_8p.loss_and_grads({'params': params}, gb, noise=noise, return_predictions=True)
  g_fast = g_f.flat.clone()
  monkeypatch.setenv('SPA3D_GEMM_IMPL', '1')
  monkeypatch.setenv('SPA3D_ATTN_IMPL', '1')
  m_gen = product_model(spa3d, cfg, 'bf16')
  ld_g, g_g, p_g = m_gen.loss_and_grads({'params': params}, gb, noise=noise, return_predictions=True)
  e = rel_err(p_f.tracks, p_g.tracks)
  cos = float((g_fast.double() @ g_g.flat.double()) / (g_fast.double().norm() * g_g.flat.double().norm()))
  print('bf16 forced 8-phase vs generic: tracks rel', e, 'grad cosine', cos)
  assert e < 2e-2 and cos > 0.995
  assert torch.isfinite(g_fast).all()
  # every gradient leaf, not only the whole-gradient direction: a wrong small leaf (a norm scale, the readout token) cannot hide
  fa, fb = O.tree_flatten(g_f), O.tree_flatten(g_g)
  worst = max((rel_err(fa[k].float(), fb[k].float()), k) for k in fb if float(fb[k].float().norm()) > 1e-12)
  print('bf16 forced 8-phase vs generic: worst gradient leaf', worst)
  assert worst[0] < 0.25  # two bf16 paths with different summation orders through 11 blocks (T=150 tests: 8 % vs the fp32 path)

  # the bias gradients ride along in the 8-phase dW kernel (column sums of dY): leaf by leaf against the generic path's colsum kernel
  def walk(a, b, path=''):
    for k in a:
      if isinstance(a[k], dict):
        yield from walk(a[k], b[k], path + '/' + k)
      else:
        yield path + '/' + k, a[k], b[k]
  nb = 0
  for name, ga, gb_ in walk(g_f, g_g):
    if name.endswith('/bias') and float(gb_.abs().max()) > 0:
      nb += 1
      assert rel_err(ga.float(), gb_.float()) < 3e-2, name   # bf16 dY differs slightly between the two paths; fp32 sums of identical inputs agree to 1e-6
  assert nb > 10


def test_bf16_and_fp32_training_trajectories_agree(spa3d):
  """bf16 activations end to end TRAIN like fp32: the full-size model (109 M parameters) at BASELINE cfg#1's shape, 30 AdamW
  steps from the same initialisation on the same batch in both modes (same injected noise).  Two trainings that differ only in rounding
  separate chaotically step by step (L1 loss, Adam's sign-like updates), so the statement tested is the one that matters: both losses
  fall by >= 5x, the curves track each other (mean |log ratio| <= 0.12, never more than 35 % apart at a step) and end within 20 % of each
  other (measured: 15 829 -> 1 555 in fp32, 15 832 -> 1 432 in bf16; largest pointwise gap 21 %)."""
  cfg = O.Config(num_output_frames=24, use_dino=True, use_depth=True, dino_feature_dim=768, depth_feature_dim=1)
  B, N, Q, T = 2, 64, 16, 24
  batch = O.synthetic_batch(B, N, Q, T, seed=1234, dino_dim=768, depth_dim=1)
  gb = batch_to(batch, 'cuda')
  noise = _noise(B, cfg).cuda()
  curves = {}
  for precision in ('fp32', 'bf16'):
    b = dict(gb)
    if precision == 'bf16':
      b['dino_features'] = gb['dino_features'].bfloat16(); b['depth_features'] = gb['depth_features'].bfloat16()
    else:
      b['dino_features'] = gb['dino_features'].bfloat16().float(); b['depth_features'] = gb['depth_features'].bfloat16().float()
    model = product_model(spa3d, cfg, precision)
    st = spa3d.TrainState(model, model.init(0, gb)['params'], learning_rate=3e-4, warmup_steps=5, total_steps=60)
    curves[precision] = [float(st.train_step(b, noise=noise)['train/loss']) for _ in range(30)]
  f, h = np.array(curves['fp32']), np.array(curves['bf16'])
  print('loss fp32', np.round(f[::5], 2).tolist(), 'bf16', np.round(h[::5], 2).tolist(), 'max rel diff', float(np.abs(f - h).max() / f.max()))
  lr_ = np.abs(np.log(f) - np.log(h))
  print('mean |log ratio|', float(lr_.mean()), 'max pointwise rel gap', float((np.abs(f - h) / f).max()))
  assert f[-1] < 0.2 * f[0] and h[-1] < 0.2 * h[0]
  assert lr_.mean() <= 0.12 and np.all(np.abs(f - h) <= 0.35 * f) and abs(f[-1] - h[-1]) <= 0.2 * f[-1]
