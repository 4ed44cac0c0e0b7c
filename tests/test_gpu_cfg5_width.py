"""BASELINE.json configs[4] ("8192 support + 2048 query, T=300, C=772, fp16") at its OWN width, under a comparison instead of a finite-loss assertion.

What only this width exercises: the chunked cross attention of `tracks_to_latents` over thousands of keys (/root/reference/track_autoencoder_3d.py:95-100,201:
128 latent queries x N track tokens, here 32 / 64 key chunks of 128 per head), the shared-row plan of the first readout block at Q = 2048 queries over 300 frames
(~300 slots per sample), the S = 301 split-pass attention backward (attention.py:175 at T = 300) over a multi-round persistent grid, `floor(t / 150)` in {0, 1}
and the time-feature window running off the 1152 latent channels (3d:235-246) -- all in the fp16 default dispatch with the workspace poisoned (NaN-filled before
the chunk), so a read of an unwritten row fails the gates.

  * N = 4096: fp16 default dispatch against the library's fp32 parity mode (which tests/test_gpu_t150.py pins to the oracle at T = 300, case c772_t300), same gate
    table as tests/test_gpu_round3.py::test_full_width_sample_...: tracks, logits, losses, every gradient leaf, whole-gradient cosine, plan fractions.
    Why not N = 8192 here: the fp32 mode keeps every activation of a sample as 4-byte values -- 2 x the ~168 GB fp16 stash of one 8192-track sample at T = 300 --
    which does not fit 288 GB; one sample is the smallest chunk.  Every kernel is per-track except the cross attention, which runs 32 instead of 64 key chunks.
  * N = 8192 (the full width): fp16 against bf16 -- two independent 16-bit roundings of the same graph through the same dispatch; the gates are those of the
    N = 4096 case widened by bf16's own distance to fp32 (8x fp16's).  This is the one run that streams 8192 keys through the chunked cross attention.
"""
import pytest
import torch

from util import Gates, O, rel_err

pytestmark = pytest.mark.gpu
Q, T = 2048, 300


def _run(spa3d, precision, batch, noise, poison=1):
  model = spa3d.TrackAutoEncoder3D(num_output_frames=T, dino_feature_dim=768, depth_feature_dim=1, precision=precision)
  b = dict(batch)
  cast = {'fp32': torch.float32, 'fp16': torch.float16, 'bf16': torch.bfloat16}[precision]
  b['dino_features'] = batch['dino_features'].to(cast); b['depth_features'] = batch['depth_features'].to(cast)
  params = model.init(0, b)['params']
  lib = spa3d._lib.load()
  h = model._handle(768, 1)[0]
  spa3d._lib.check(lib.spa3d_set_option(h, b'poison', float(poison)), h)
  ld, grads, preds = model.loss_and_grads({'params': params}, b, noise=noise, return_predictions=True)
  torch.cuda.synchronize()
  spa3d._lib.check(lib.spa3d_set_option(h, b'poison', 0.0), h)
  o = (spa3d._lib.C.c_double * 4)()
  spa3d._lib.check(lib.spa3d_plan_stats(h, o))
  out = ([float(ld[k]) for k in ('total_loss', 'position_loss', 'visible_loss')], {k: v.clone() for k, v in O.tree_flatten(grads).items()},
         preds.tracks.clone(), preds.visible_logits.clone(), list(o))
  del model, params, grads, preds, b
  torch.cuda.empty_cache()
  return out


def _compare(title, lo, ref, bounds, measured):
  (l16, g16, t16, v16, _), (l32, g32, t32, v32, _) = lo, ref
  assert all(bool(torch.isfinite(x).all()) for x in (t16, v16)) and all(bool(torch.isfinite(g16[k]).all()) for k in g16)
  names = sorted(g32)
  a = torch.cat([g16[k].double().reshape(-1) for k in names]); b_ = torch.cat([g32[k].double().reshape(-1) for k in names])
  cos = float((a @ b_) / (a.norm() * b_.norm()))
  tot = float(b_.norm())
  worst = max((rel_err(g16[k], g32[k]), k) for k in names if float(g32[k].double().norm()) > 1e-3 * tot)
  print(f'{title}: losses {l16} vs {l32}; worst significant gradient leaf {worst}; 1 - cos {1.0 - cos:.3e}')
  gt = Gates(title)
  gt.le('tracks, relative Frobenius', rel_err(t16, t32), bounds[0], measured[0])
  gt.le('visible logits, relative Frobenius', rel_err(v16, v32), bounds[1], measured[1])
  gt.le('total loss, relative', abs(l16[0] - l32[0]) / abs(l32[0]), bounds[2], measured[2])
  gt.le('worst significant gradient leaf, relative', worst[0], bounds[3], measured[3])
  gt.le('1 - cosine(whole gradient)', 1.0 - cos, bounds[4], measured[4])
  gt.check()


def _plan(stats):
  kept, slots = stats[0] / stats[1], stats[2] / stats[3]
  print(f'plan stats: encoder rows kept {kept:.3f}, readout slots per query {slots:.3f}')
  assert 0.85 < kept < 0.95, kept        # Bernoulli(0.9) visibility: the pruned encoder path ran
  assert 0.10 < slots < 0.20, slots      # 2048 queries over 300 frames -> ~300 distinct frames per sample: the shared-row path ran (<= 0.45)


def test_cfg5_width_fp16_vs_fp32_parity_mode_n4096():
  import spa3d
  import bench  # synthetic-input recipe of the benchmark; nothing is timed here
  dev = torch.device('cuda', 0)
  batch = bench.synth_batch(1, 4096, Q, T, 768, 1, dev, seed=314, feat_dtype=torch.float16)
  noise = torch.rand(1, 128, 96, generator=torch.Generator().manual_seed(5)).to(dev)
  lo = _run(spa3d, 'fp16', batch, noise)
  _plan(lo[4])
  ref = _run(spa3d, 'fp32', batch, noise, poison=0)
  _compare('cfg#5 width (N = 4096, Q = 2048, T = 300): fp16 default dispatch vs the fp32 parity mode', lo, ref,
           (1.55e-3, 1.55e-3, 3.0e-5, 4.4e-2, 5.6e-6), ('1.03e-3 (round 5)', '1.03e-3', '2.0e-5', '2.9e-2: tracks_to_latents/layer_0/cross_att/dense_query/kernel', '3.7e-6'))


def test_cfg5_full_width_fp16_vs_bf16_n8192():
  import spa3d
  import bench
  dev = torch.device('cuda', 0)
  batch = bench.synth_batch(1, 8192, Q, T, 768, 1, dev, seed=315, feat_dtype=torch.float16)
  noise = torch.rand(1, 128, 96, generator=torch.Generator().manual_seed(6)).to(dev)
  lo = _run(spa3d, 'fp16', batch, noise)
  _plan(lo[4])
  # bf16 reads bf16-rounded feature planes: part of the difference below is that input rounding (2^-9 relative), as in every bf16-vs-fp16 comparison of the suite
  ref = _run(spa3d, 'bf16', batch, noise)
  _compare('cfg#5 FULL width (N = 8192, Q = 2048, T = 300): fp16 vs bf16, default dispatch both', lo, ref,
           (1.3e-2, 1.3e-2, 1.4e-4, 0.11, 1.0e-4), ('8.5e-3 (round 5)', '8.5e-3', '9.3e-5', '7.3e-2: decompress_attn/layer_2/self_att/dense_key/kernel', '6.6e-5'))
