"""Data-parallel semantics on CPU with 2 gloo ranks (the N>1 path of bench.py / TrainState):
the batch shards over ranks, the loss denominator is the GLOBAL visible count (one scalar all-reduce before the
backward, train.py:111-113), gradients are SUM-all-reduced in buckets, and the result equals the single-process
gradient of the whole batch.  The oracle plays the compute kernel here (no GPU in this container); the collectives
and their host logic are the product's (3dspa_code_amd/train.py).  test_trainstate_step_world2_equals_full_batch drives
TrainState.train_step ITSELF (world > 1 branch: denominator pre-reduce, rank slice of the global discretisation noise,
bucketed SUM all-reduce, clip + AdamW, rank-0 broadcast at construction) with the oracle injected as compute / AdamW."""
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
    if p.is_alive():
      p.kill()
    assert p.exitcode == 0, f'worker exit code {p.exitcode}'
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


# ---------------------------------------------------------------------------------------------------------------------
# TrainState.train_step at world size 2 (the product's host logic end to end, compute injected)
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_hooks(cfg, names_shapes):
  """compute / adamw / noise_fn stand-ins with the signatures TrainState documents (oracle = checker-side code only)."""
  import numpy as np
  from oracle import np_blocks as NB
  m = O.TrackAutoEncoder3D(cfg)

  def to_tree(flat_tree):
    return O.tree_unflatten({k: v.double() for k, v in O.tree_flatten(flat_tree).items()})

  def compute(params, batch, grads_flat, denom, discretize, noise):
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    ld, _, g = O.loss_and_grads(m, to_tree(params), b64, discretize=discretize, noise=None if noise is None else noise.double(),
                                denom=denom if denom > 0 else None)
    for name, shape, off in names_shapes:
      n = int(np.prod(shape))
      grads_flat[off:off + n] = g[name].reshape(-1).float()
    return ld

  def adamw(flat, grads, mm, vv, lr, step, clip, b1, b2, eps, wd, scratch):
    gn = O.adamw_step({'p': flat}, {'p': grads}, {'p': mm}, {'p': vv}, step, lr, clip=clip, wd=wd, b1=b1, b2=b2, eps=eps)
    scratch[0] = gn

  def noise_fn(n, device):
    return torch.from_numpy(NB.jax_uniform_legacy((n,), (0, 0)).astype(np.float32)).to(device)

  def evaluate(params, batch, denom, discretize, noise):
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    preds = m(to_tree(params), b64, discretize=discretize, noise=None if noise is None else noise.double())
    return O.compute_loss_3d(preds, b64, denom=denom if denom > 0 else None), preds

  compute.evaluate = evaluate
  return compute, adamw, noise_fn


def _ts_worker(rank, world, port, q):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    import spa3d
    from util import product_model
    torch.set_num_threads(2)
    cfg = O.Config(**MINI, use_dino=False, use_depth=False)
    B = 4
    batch = O.synthetic_batch(B, 5, 4, 8, seed=17)
    batch['query_tracks_visible'][0] = 0
    model = product_model(spa3d, cfg, 'fp32')  # host object only: handle creation + leaf table run without a GPU
    # every rank starts from DIFFERENT parameters: construction must broadcast rank 0's
    params = O.init_params(cfg, seed=5 + rank, dtype=torch.float32, with_dino=False, with_depth=False, perturb=0.1)
    _, leaves, n = model._handle(0, 0)
    compute, adamw, noise_fn = _oracle_hooks(cfg, leaves)
    st = spa3d.TrainState(model, params, learning_rate=1e-2, warmup_steps=1, total_steps=10, grad_bucket_bytes=4000, compute=compute,
                          adamw=adamw, noise_fn=noise_fn, evaluate=compute.evaluate)
    ref0 = model.flat_from_tree(O.init_params(cfg, seed=5, dtype=torch.float32, with_dino=False, with_depth=False, perturb=0.1))
    bcast_ok = bool(torch.equal(st.flat, ref0))
    lo, hi = rank * B // world, (rank + 1) * B // world
    shard = {k: v[lo:hi] for k, v in batch.items()}
    ev, _ = st.eval_step(shard)                      # before any update: must equal the full-batch loss of the first train step
    outs = [st.train_step(shard) for _ in range(2)]  # noise=None: the rank slice of the global PRNGKey(0) draw
    if rank == 0:
      # single-process reference: the same TrainState code at world 1 semantics, full batch, global noise
      c1, a1, n1 = _oracle_hooks(cfg, leaves)
      flat = ref0.clone(); mm = torch.zeros_like(flat); vv = torch.zeros_like(flat); g = torch.zeros_like(flat); sc = torch.zeros(4)
      sched = spa3d.create_learning_rate_schedule(1e-2, 1, 10)
      full_noise = n1(B * cfg.num_latent_tokens * cfg.latent_token_dim, 'cpu').view(B, cfg.num_latent_tokens, cfg.latent_token_dim)
      losses = []
      for step in range(2):
        ld = c1(model.tree_from_flat(flat, 0, 0), batch, g, 0.0, True, full_noise)
        losses.append(float(ld['total_loss']))
        a1(flat, g, mm, vv, sched(step), step, 1.0, 0.9, 0.999, 1e-8, 0.01, sc)
      q.put((bcast_ok, float((st.flat - flat).abs().max()), [float(o['train/loss']) for o in outs], losses,
             float(outs[-1]['train/grad_norm']), float(sc[0]), float(ev['eval/loss'])))
    else:
      q.put((bcast_ok,))
  finally:
    dist.destroy_process_group()


def test_trainstate_step_world2_equals_full_batch():
  ctx = mp.get_context('spawn')
  q = ctx.SimpleQueue()
  port = _free_port()
  procs = [ctx.Process(target=_ts_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  for p in procs:  # join with a timeout and check the exit codes BEFORE draining the queue: a crashed worker must fail, not hang, the suite
    p.join(300)
    if p.is_alive():
      p.kill()
    assert p.exitcode == 0, f'worker exit code {p.exitcode}'
  res = [q.get(), q.get()]
  full = next(r for r in res if len(r) > 1)
  assert all(r[0] for r in res), 'rank-0 parameter broadcast at construction'
  _, perr, losses, ref_losses, gn, gn_ref, ev_loss = full
  assert abs(ev_loss - ref_losses[0]) < 1e-5 * abs(ref_losses[0])  # eval_step: global denominator, losses summed over ranks
  assert perr < 2e-6, perr  # two AdamW steps on fp32 buffers; gradients differ only by summation order
  for a, b in zip(losses, ref_losses):
    assert abs(a - b) < 1e-5 * abs(b)
  assert abs(gn - gn_ref) < 1e-4 * gn_ref


# ---------------------------------------------------------------------------------------------------------------------
# resume under data parallelism: the optimizer counters travel with the state
# ---------------------------------------------------------------------------------------------------------------------
def _resume_worker(rank, world, port, q, path):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    import spa3d
    from util import product_model
    torch.set_num_threads(2)
    cfg = O.Config(**MINI, use_dino=False, use_depth=False)
    B = 4
    batch = O.synthetic_batch(B, 5, 4, 8, seed=23)
    model = product_model(spa3d, cfg, 'fp32')
    params = O.init_params(cfg, seed=9 + rank, dtype=torch.float32, with_dino=False, with_depth=False, perturb=0.1)
    _, leaves, n = model._handle(0, 0)
    compute, _, noise_fn = _oracle_hooks(cfg, leaves)

    def adamw(flat, grads, mm, vv, lr, step, clip, b1, b2, eps, wd, scratch):  # the HIP kernel's rule: bias correction counts APPLIED updates (spa3d.h)
      gn = O.adamw_step({'p': flat}, {'p': grads}, {'p': mm}, {'p': vv}, step - int(scratch[3]), lr, clip=clip, wd=wd, b1=b1, b2=b2, eps=eps)
      scratch[0] = gn

    mk = lambda p: spa3d.TrainState(model, p, learning_rate=1e-2, warmup_steps=1, total_steps=10, grad_bucket_bytes=4000, compute=compute, adamw=adamw,
                                    noise_fn=noise_fn)
    s0 = mk(params)  # (every rank constructs it: the constructor broadcasts rank 0's buffers, a collective)
    if rank == 0:  # a state that has made 5 calls of which 2 were skipped, with a halved loss-scale multiplier
      s0.step = 5; s0.m.normal_(generator=torch.Generator().manual_seed(1)).mul_(1e-3); s0.v.uniform_(1e-6, 1e-5, generator=torch.Generator().manual_seed(2))
      s0.scratch[3:6] = torch.tensor([2.0, 0.5, 7.0])
      spa3d.save_checkpoint(path, s0.params, s0)
    dist.barrier()
    st = mk(O.init_params(cfg, seed=40 + rank, dtype=torch.float32, with_dino=False, with_depth=False, perturb=0.1))
    spa3d.load_train_state(path, st, rank0_only=True)   # rank 1 never opens the file
    counters = [float(x) for x in st.scratch[3:6]]
    lo, hi = rank * B // world, (rank + 1) * B // world
    st.train_step({k: v[lo:hi] for k, v in batch.items()})
    import hashlib  # (digests, not arrays: the parent joins before it drains the queue, and a pipe-sized payload would block the put)
    dg = lambda t: hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()
    q.put((rank, st.step, counters, dg(st.flat), dg(st.m), float(st.flat.double().abs().sum())))
  finally:
    dist.destroy_process_group()


def test_dp_resume_carries_the_optimizer_counters_to_every_rank(tmp_path):
  ctx = mp.get_context('spawn')
  q = ctx.SimpleQueue()
  port = _free_port()
  path = str(tmp_path / 'state.npz')
  procs = [ctx.Process(target=_resume_worker, args=(r, 2, port, q, path)) for r in range(2)]
  for p in procs:
    p.start()
  res = []
  for p in procs:
    p.join(300)
    if p.is_alive():
      p.kill()
    assert p.exitcode == 0, f'worker exit code {p.exitcode}'
  res = sorted([q.get(), q.get()], key=lambda r: r[0])
  for r in res:
    assert r[1] == 6 and r[2] == [2.0, 0.5, 7.0], r[1:3]   # step and scratch[3:6] on BOTH ranks (rank 1 got them by broadcast)
  assert res[0][3] == res[1][3] and res[0][4] == res[1][4], 'replicas diverged after one step from a resumed state'
  assert res[0][5] > 0.0
