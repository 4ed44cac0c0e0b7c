"""Generates tests/golden/t150_golden.npz: the full-size model (109 M parameters) at the BENCHMARK's sequence
lengths (T = T_out = 150: S = 151 track sequences, 129-token readout sequences) on small B/N/Q, from the CPU ORACLE of
this repo (oracle/spa3d_oracle.py, fp64).

These are NOT reference outputs: the reference (JAX/Flax) cannot be imported or executed here and holds no fixtures
 -- PARITY UNPINNED.  The fixture freezes the oracle's fp64 outputs / latents / losses /
per-leaf gradient norms (plus a few whole gradient leaves) so that the GPU parity tests at T=150
(tests/test_gpu_t150.py) do not depend on re-running a 109 M-parameter fp64 model on the GPU box's host cores.

Parameters and batches are NOT stored (436 MB): both sides regenerate them from fixed seeds with the same
torch CPU generators (O.init_params / O.synthetic_batch); checksums of the regenerated inputs are stored and
checked first, so a generator drift shows up as "inputs differ", not as a parity failure.

Cases (BASELINE.json configs[1] and configs[2] channel sets):
  c772 : xyz + depth(1) + DINOv2-768, B=2, N=6, Q=4, boundary_frame = (150, 97)
  c4   : xyz + depth(1) only (depth-only parametrisation, cfg#2), same sizes
  c772_t300 : T = T_out = 300, B=1, N=4, Q=8 (BASELINE configs[4] sequence length; see the case table)
  c772_q320 : B=1, N=6, Q=320: several queries per frame (the shared-latent-row path of the readout stack's first block)
  c772_tiles : B=2, N=64, Q=64 -> M = 19 328 track-token rows and 16 512 readout rows, i.e. above the 16 384-row
         threshold at which the default dispatch takes the 8-phase / persistent MFMA kernels the benchmark runs on.
bf16 inputs: DINO / depth features are rounded to bf16 before the oracle sees them (the product's bf16 mode stores
them so); the fp32 test feeds the same rounded values.

    python tests/golden/make_t150_golden.py [case ...]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import spa3d_oracle as O  # noqa: E402

CASES = {
    'c772': dict(B=2, N=6, Q=4, dino=768, depth=1, boundary=(150, 97), pseed=5, bseed=77),
    'c4': dict(B=2, N=6, Q=4, dino=0, depth=1, boundary=(150, 97), pseed=6, bseed=78),
    'c772_tiles': dict(B=2, N=64, Q=64, dino=768, depth=1, boundary=(150, 120), pseed=5, bseed=79),
    # BASELINE configs[4] sequence length: T = T_out = 300 (S = 301).  Query frames are set by hand so that floor(t / 150) takes
    # both values (track_autoencoder_3d.py:268-269) and the 128-wide window lat[:, 5t : 5t+128] runs partly (t = 205..230) and
    # wholly (t >= 231) off the end of the 1152 latent channels (:239-245).
    # 320 queries over 150 frames (about 130 distinct frames: below the 45 % threshold at which the product computes readout block 1's
    # LayerNorm / QKV once per distinct (sample, query frame), DESIGN.md "Shared latent rows"): the benchmark's regime, many queries per frame
    'c772_q320': dict(B=1, N=6, Q=320, dino=768, depth=1, boundary=(150,), pseed=5, bseed=81),
    'c772_t300': dict(B=1, N=4, Q=8, T=300, dino=768, depth=1, boundary=(260,), pseed=8, bseed=80, qframes=(0, 149, 150, 204, 205, 230, 231, 299)),
}
KEEP_GRADS = ('input_readout_token/state_init', 'depth_projection/kernel', 'track_token_projection/bias',
              'input_track_transformer/layer_0/norm_q/scale', 'input_track_transformer/layer_2/self_att/norm_key/scale',
              'input_track_transformer/layer_2/self_att/dense_out/bias', 'input_track_transformer/norm_encoder/scale',
              'track_readout_attn/layer_3/norm_attn/scale', 'track_readout_attn/layer_0/self_att/norm_query/scale',
              'track_readout_attn/norm_encoder/scale', 'track_predictor/bias', 'compressor/bias')


def make_inputs(case):
  """(cfg, params fp32, batch fp32 with bf16-representable features, noise) of one case; shared with the tests."""
  c = CASES[case]
  T = c.get('T', 150)
  cfg = O.Config(num_output_frames=T, use_dino=c['dino'] > 0, use_depth=c['depth'] > 0, dino_feature_dim=max(c['dino'], 1),
                 depth_feature_dim=max(c['depth'], 1))
  p = O.init_params(cfg, seed=c['pseed'], dtype=torch.float32, with_dino=c['dino'] > 0, with_depth=c['depth'] > 0,
                    depth_dim=c['depth'], perturb=0.1)
  batch = O.synthetic_batch(c['B'], c['N'], c['Q'], T, seed=c['bseed'], dino_dim=c['dino'], depth_dim=c['depth'])
  batch['boundary_frame'] = torch.tensor(c['boundary'], dtype=torch.int32)
  if 'qframes' in c:  # query point = (t, position of the query track at frame t), data_loader.py:77-85
    tq = torch.tensor(c['qframes'])[None].expand(c['B'], -1)
    xyz = torch.gather(batch['query_tracks'], 2, tq[:, :, None, None].expand(c['B'], c['Q'], 1, 3))[:, :, 0]
    batch['query_points'] = torch.cat([tq[..., None].float(), xyz], dim=-1)
  for k in ('dino_features', 'depth_features'):
    if k in batch:
      batch[k] = batch[k].bfloat16().float()
  noise = torch.rand(c['B'], cfg.num_latent_tokens, cfg.latent_token_dim, generator=torch.Generator().manual_seed(c['bseed'] + 1))
  return cfg, p, batch, noise


def checksums(p, batch, noise):
  flat = O.tree_flatten(p)
  names = sorted(flat)
  ps = np.array([float(flat[k].double().sum()) for k in names] + [float((flat[k].double() ** 2).sum()) for k in names])
  bs = np.array([float(batch[k].double().sum()) for k in sorted(batch)] + [float(noise.double().sum())])
  return ps, bs


def run_case(case):
  cfg, p, batch, noise = make_inputs(case)
  p64 = O.tree_map(lambda t: t.double(), p)
  b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
  m = O.TrackAutoEncoder3D(cfg)
  t0 = time.time()
  ld, preds, grads = O.loss_and_grads(m, p64, b64, discretize=True, noise=noise.double())
  with torch.no_grad():
    lat = m.encode(p64, b64)
  print(f'{case}: oracle fp64 fwd+bwd {time.time() - t0:.1f} s, loss {float(ld["total_loss"]):.6f}', flush=True)
  out = {}
  ps, bs = checksums(p, batch, noise)
  out[f'{case}/param_checksums'] = ps
  out[f'{case}/batch_checksums'] = bs
  out[f'{case}/tracks'] = preds.tracks.detach().numpy().astype(np.float32)  # fp32 storage: 1e-7 relative, tolerances are >= 1e-5
  out[f'{case}/visible_logits'] = preds.visible_logits.detach().numpy().astype(np.float32)
  out[f'{case}/latents'] = lat.numpy().astype(np.float32)
  out[f'{case}/losses'] = np.array([float(ld['total_loss']), float(ld['position_loss']), float(ld['visible_loss'])])
  names = sorted(grads)
  out[f'{case}/grad_names'] = np.array(names)
  out[f'{case}/grad_norms'] = np.array([float(grads[k].norm()) for k in names])
  for k in KEEP_GRADS:
    if k in grads:
      out[f'{case}/grad/{k}'] = grads[k].numpy().astype(np.float32)
  return out


def main():
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 't150_golden.npz')
  out = {}
  if os.path.exists(path):
    z = np.load(path, allow_pickle=False)
    out = {k: z[k] for k in z.files}
  for case in (sys.argv[1:] or list(CASES)):
    out = {k: v for k, v in out.items() if not k.startswith(case + '/')}
    out.update(run_case(case))
  np.savez_compressed(path, **out)
  print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
  torch.set_num_threads(8)
  main()
