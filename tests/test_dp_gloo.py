"""Data-parallel semantics on CPU with 2 gloo ranks (the N>1 path of bench.py / TrainState):
the batch shards over ranks, the loss denominator is the GLOBAL visible count (one scalar all-reduce before the
backward, train.py:111-113), gradients are SUM-all-reduced in buckets, and the result equals the single-process
gradient of the whole batch.  The oracle plays the compute kernel here (no GPU in this container); the collectives
and their host logic are the product's (3dspa_code_amd/train.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import MINI, O


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, q):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    import spa3d
    torch.set_num_threads(2)
    cfg = O.Config(**MINI, use_dino=False, use_depth=False)
    B = 4
    batch = O.synthetic_batch(B, 5, 4, 8, seed=17, dtype=torch.float64)
    batch['query_tracks_visible'][0] = 0  # make the shards' visible counts differ a lot
    params = O.init_params(cfg, seed=5, dtype=torch.float64, with_dino=False, with_depth=False, perturb=0.1)
    noise = torch.rand(B, cfg.num_latent_tokens, cfg.latent_token_dim, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    m = O.TrackAutoEncoder3D(cfg)
    lo, hi = rank * B // world, (rank + 1) * B // world
    shard = {k: v[lo:hi] for k, v in batch.items()}
    denom = spa3d.global_visible_count(shard['query_tracks_visible'])
    ld, _, g = O.loss_and_grads(m, params, shard, noise=noise[lo:hi], denom=denom)
    names = sorted(g)
    flat = torch.cat([g[k].reshape(-1) for k in names])
    l3 = torch.stack([ld['total_loss'], ld['position_loss'], ld['visible_loss']])
    spa3d.allreduce_flat_(flat, bucket_elems=1000, extra=(l3,))  # several buckets + the loss scalars
    if rank == 0:
      ld_ref, _, g_ref = O.loss_and_grads(m, params, batch, noise=noise)
      ref = torch.cat([g_ref[k].reshape(-1) for k in names])
      q.put((float((flat - ref).abs().max()), float(ref.abs().max()), float(l3[0]), float(ld_ref['total_loss']), denom,
             float(batch['query_tracks_visible'].sum())))
  finally:
    dist.destroy_process_group()


def test_dp2_gradients_equal_full_batch():
  ctx = mp.get_context('spawn')
  q = ctx.SimpleQueue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  for p in procs:
    p.join(300)
    assert p.exitcode == 0
  err, scale, loss, loss_ref, denom, vis = q.get()
  assert denom == vis  # global count, identical on every rank
  assert err < 1e-9 * max(1.0, scale)
  assert abs(loss - loss_ref) < 1e-9 * abs(loss_ref)


def test_local_denominator_would_be_wrong():
  """Why the scalar pre-reduce exists: averaging per-shard losses (local denominators) is NOT the reference loss."""
  cfg = O.Config(**MINI, use_dino=False, use_depth=False)
  batch = O.synthetic_batch(4, 5, 4, 8, seed=17, dtype=torch.float64)
  batch['query_tracks_visible'][0] = 0
  params = O.init_params(cfg, seed=5, dtype=torch.float64, with_dino=False, with_depth=False)
  m = O.TrackAutoEncoder3D(cfg)
  full = float(O.compute_loss_3d(m(params, batch, discretize=False), batch)['total_loss'])
  halves = [float(O.compute_loss_3d(m(params, {k: v[i:i + 2] for k, v in batch.items()}, discretize=False),
                                    {k: v[i:i + 2] for k, v in batch.items()})['total_loss']) for i in (0, 2)]
  assert abs(sum(halves) / 2 - full) > 1e-3 * abs(full)
