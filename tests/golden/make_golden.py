"""Generates tests/golden/mini_golden.npz from the CPU ORACLE of this repo (oracle/spa3d_oracle.py, fp64).

These are NOT reference outputs: the reference (JAX/Flax) cannot be imported or executed here and holds no
fixtures of its own -- PARITY UNPINNED.  The fixture freezes the oracle's numbers so that
(a) the oracle cannot drift silently (tests/test_golden.py, CPU) and (b) the HIP path is checked against data that
travels to the GPU box without the oracle having to be re-run in a particular precision (tests/test_golden.py, GPU).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import spa3d_oracle as O  # noqa: E402
from util import MINI  # noqa: E402

DINO, DEPTH = 24, 1
B, N, Q, T = 3, 10, 6, 8


def main():
  cfg = O.Config(**MINI, use_dino=True, use_depth=True, dino_feature_dim=DINO, depth_feature_dim=DEPTH)
  p = O.init_params(cfg, seed=7, dtype=torch.float64, depth_dim=DEPTH, perturb=0.1)
  p = O.tree_map(lambda t: t.float().double(), p)  # values exactly representable in the fp32 parameter buffer
  batch = O.synthetic_batch(B, N, Q, T, seed=99, dino_dim=DINO, depth_dim=DEPTH)
  batch['boundary_frame'] = torch.tensor([8, 5, 3], dtype=torch.int32)
  noise = torch.rand(B, cfg.num_latent_tokens, cfg.latent_token_dim, generator=torch.Generator().manual_seed(11))
  b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
  m = O.TrackAutoEncoder3D(cfg)
  ld, preds, grads = O.loss_and_grads(m, p, b64, discretize=True, noise=noise.double())
  lat = m.encode(p, b64)
  out = {f'param/{k}': v.float().numpy() for k, v in O.tree_flatten(p).items()}
  out.update({f'batch/{k}': v.numpy() for k, v in batch.items()})
  out['noise'] = noise.numpy()
  out['expect/tracks'] = preds.tracks.detach().numpy()
  out['expect/visible_logits'] = preds.visible_logits.detach().numpy()
  out['expect/latents'] = lat.detach().numpy()
  out['expect/losses'] = np.array([float(ld['total_loss']), float(ld['position_loss']), float(ld['visible_loss'])])
  names = sorted(grads)
  out['expect/grad_names'] = np.array(names)
  out['expect/grad_norms'] = np.array([float(grads[k].norm()) for k in names])
  for k in ('track_token_projection/kernel', 'dino_projection/kernel', 'input_track_transformer/layer_1/self_att/norm_query/scale',
            'tracks_to_latents/layer_0/cross_att/dense_key/kernel', 'track_readout_attn/layer_1/MLP_out/bias', 'initializer/state_init'):
    out[f'expect/grad/{k}'] = grads[k].numpy()
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mini_golden.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
  main()
