"""RCCL on the one GPU of the test box: a 1-rank `nccl` (= RCCL on ROCm) process group, with TrainState's collectives FORCED on
(`force_collectives=True`): rank-0 broadcast of parameters + Adam moments, the scalar denominator all-reduce, the rank slice of the
global noise and the bucketed asynchronous gradient all-reduce all go through the RCCL communicator on device buffers.  At world size 1
every reduction is the identity, so the step must equal the plain single-process step (up to the run-to-run last-bit differences of
the float-atomic gradient accumulation: two plain runs differ by as much).  (More than one rank per GPU is not
possible with RCCL; the N > 1 semantics are covered at world size 2 over gloo in tests/test_dp_gloo.py, and the driver's scaling run
exercises 2/4/8 GPUs.)  Runs in a child process so the process group does not leak into the other tests."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
  import os, sys
  sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
  import torch, torch.distributed as dist
  from util import MINI, O, product_model
  import spa3d
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', RANK='0', WORLD_SIZE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
  torch.cuda.set_device(0)
  dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
  cfg = O.Config(**MINI, use_dino=True, use_depth=True, dino_feature_dim=24, depth_feature_dim=1)
  batch = {k: v.cuda() for k, v in O.synthetic_batch(3, 10, 6, 8, seed=5, dino_dim=24, depth_dim=1).items()}
  outs = []
  for force in (False, True):
    model = product_model(spa3d, cfg, 'bf16')
    st = spa3d.TrainState(model, model.init(0, batch)['params'], learning_rate=1e-2, warmup_steps=1, total_steps=10,
                          grad_bucket_bytes=4096, force_collectives=force)
    for _ in range(3):
      m = st.train_step(batch)
    torch.cuda.synchronize()
    outs.append((st.flat.clone(), float(m['train/loss']), float(m['train/grad_norm'])))
  assert outs[1][1] == outs[1][1] and abs(outs[1][1]) < 1e30
  diff = float((outs[0][0] - outs[1][0]).abs().max())
  same = diff < 2e-3 and abs(outs[0][1] - outs[1][1]) <= 1e-5 * abs(outs[0][1]) and abs(outs[0][2] - outs[1][2]) <= 1e-4 * abs(outs[0][2])
  # the plain step draws its noise inside the library, the forced one passes the rank slice of the global draw: same values
  print('RCCL_WORLD1', same, diff, outs[0][1], outs[1][1], outs[0][2], outs[1][2], dist.get_backend())
  dist.destroy_process_group()
  assert same
''') % (ROOT, ROOT)


def test_rccl_world1_forced_collectives_equal_the_plain_step():
  r = subprocess.run([sys.executable, '-c', CHILD], capture_output=True, text=True, timeout=600)
  print(r.stdout[-2000:], r.stderr[-3000:])
  assert r.returncode == 0 and 'RCCL_WORLD1 True' in r.stdout


CHILD2 = textwrap.dedent('''
  import ctypes as C, gc, os, sys
  sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
  import torch, torch.distributed as dist
  from util import MINI, O, product_model
  import spa3d
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29578', RANK='0', WORLD_SIZE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
  torch.cuda.set_device(0)
  dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
  cfg = O.Config(**MINI, use_dino=True, use_depth=True, dino_feature_dim=24, depth_feature_dim=1)
  batch = {k: v.cuda() for k, v in O.synthetic_batch(3, 10, 6, 8, seed=5, dino_dim=24, depth_dim=1).items()}
  model = product_model(spa3d, cfg, 'bf16')
  mk = lambda: spa3d.TrainState(model, model.init(0, batch)['params'], learning_rate=1e-2, warmup_steps=1, total_steps=10,
                                grad_bucket_bytes=4096, force_collectives=True)
  lib = spa3d._lib.load()
  h = model._handle(24, 1)[0]
  def registered():
    g4 = (C.c_int64 * 4)(); assert lib.spa3d_grad_events_recorded(h, g4) == 0; return int(g4[2]), int(g4[3])
  st = mk()
  old = st
  st = mk()                     # the rebinding of a resume: the NEW state registers its events ...
  assert old._overlap is not None and st._overlap is not None
  mine = (int(st._overlap[1][0].cuda_event), int(st._overlap[1][1].cuda_event))
  assert registered() == mine
  old.close(); del old; gc.collect()   # ... and the old state's teardown (its __del__ runs AFTER the new constructor) must not detach them
  assert registered() == mine, 'the old state detached the new state events'
  ref = mk(); ref._overlap = None      # reference: same steps with the stream-ordered all-reduce (this state holds no registration: `st` re-registers below)
  lib.spa3d_set_grad_events(h, C.c_void_p(mine[0]), C.c_void_p(mine[1]))
  st._ev_gen = st._events_recorded()
  used = []
  for _ in range(3):
    g0 = st._events_recorded(); m = st.train_step(batch); g1 = st._events_recorded()
    used.append(g1[2] and g1[0] == g0[0] + 1 and g1[1] == g0[1] + 1)
  torch.cuda.synchronize()
  a = (st.flat.clone(), float(m['train/loss']))
  # a state whose events are NOT the registered ones must fall back to stream order instead of waiting on stale events
  stale = mk()                          # registers ITS events ...
  lib.spa3d_set_grad_events(h, C.c_void_p(mine[0]), C.c_void_p(mine[1]))   # ... which are then replaced behind its back
  for _ in range(3):
    m2 = stale.train_step(batch)
    assert stale._events_recorded()[2] is False
  for _ in range(3):
    m3 = ref.train_step(batch)
  torch.cuda.synchronize()
  d1 = float((stale.flat - ref.flat).abs().max()); d2 = float((a[0] - ref.flat).abs().max())
  print('TWO_STATES', all(used), d1, d2, a[1], float(m2['train/loss']), float(m3['train/loss']))
  dist.destroy_process_group()
  assert all(used) and d1 < 2e-3 and d2 < 2e-3
''') % (ROOT, ROOT)


def test_two_train_states_on_one_model_keep_the_newer_registration():
  """rebinding `state = TrainState(model, ...)` runs the old state's __del__ after the new one has registered its gradient-segment
  events on the shared handle.  The teardown is owner-aware (spa3d_detach), the overlapped all-reduce is used only when THIS call recorded
  THIS state's events (spa3d_grad_events_recorded), and a state whose events were replaced falls back to the stream-ordered reduction."""
  r = subprocess.run([sys.executable, '-c', CHILD2], capture_output=True, text=True, timeout=600)
  print(r.stdout[-2000:], r.stderr[-3000:])
  assert r.returncode == 0 and 'TWO_STATES True' in r.stdout
