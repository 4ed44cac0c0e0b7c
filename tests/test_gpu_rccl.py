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
