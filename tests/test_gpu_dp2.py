"""Data parallelism with the REAL HIP step at world size 2 on the one GPU of the test box: two processes share cuda:0, the collectives go over
gloo (RCCL refuses two ranks on one device; the collective library is not what is under test here -- TrainState's data-parallel logic around
the library's own forward / backward / AdamW is: rank-0 broadcast, the global loss denominator, the rank slice of the global
discretisation noise, the bucketed gradient SUM, clip + AdamW on reduced gradients).  Two steps on a 2-sample batch split 1 + 1 must equal
the single-process steps on the whole batch (fp32 parity mode: only the summation order differs).  tests/test_dp_gloo.py covers the same
logic on CPU with the oracle injected; tests/test_gpu_rccl.py runs RCCL itself at world size 1; bench.py --gpus N is the N-GPU run.
BASELINE.json configs[3] (no reference counterpart: SURVEY 2)."""
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
  rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
  torch.cuda.set_device(0)
  if world > 1:
    dist.init_process_group('gloo', rank=rank, world_size=world)
  cfg = O.Config(**MINI, use_dino=True, use_depth=True, dino_feature_dim=24, depth_feature_dim=1)
  BT = int(os.environ.get('BATCH', '2')); bl = BT // world
  full = {k: v.cuda() for k, v in O.synthetic_batch(BT, 10, 6, 8, seed=5, dino_dim=24, depth_dim=1).items()}
  full['query_tracks_visible'][0, :3] = 0   # unequal visible counts per shard: local denominators would be wrong
  batch = {k: v[rank * bl:(rank + 1) * bl].contiguous() for k, v in full.items()} if world > 1 else full
  prec = os.environ.get('PREC', 'fp32')
  if prec != 'fp32':
    for kk in ('dino_features', 'depth_features'):
      batch[kk] = batch[kk].half() if prec == 'fp16' else batch[kk].bfloat16(); full[kk] = full[kk].to(batch[kk].dtype)
  model = product_model(spa3d, cfg, prec)
  # deliberately different initial parameters per rank: the construction-time broadcast must make them rank 0's
  st = spa3d.TrainState(model, model.init(rank, full)['params'], learning_rate=1e-2, warmup_steps=1, total_steps=10, grad_bucket_bytes=4096)
  losses = []
  for _ in range(3):
    m = st.train_step(batch)
    losses.append(float(m['train/loss']))
  torch.cuda.synchronize()
  # EVERY rank saves its replica: the test compares them bit for bit
  torch.save({'flat': st.flat.cpu(), 'm': st.m.cpu(), 'losses': losses, 'gn': float(m['train/grad_norm']), 'overlap': st._overlap is not None},
             os.environ['OUT'] + '.rank%%d' %% rank)
  if world > 1:
    dist.barrier(); dist.destroy_process_group()
''') % (ROOT, ROOT)


def _launch(world, out, port, **extra):
  procs = []
  for r in range(world):
    env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), OUT=out, **extra)
    procs.append(subprocess.Popen([sys.executable, '-c', CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
  outs = []
  for p in procs:
    try:
      o, _ = p.communicate(timeout=600)
    except subprocess.TimeoutExpired:
      p.kill(); o = 'timeout'
    outs.append((p.returncode, o))
  return outs


def _port():
  import socket
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  return port


def test_hip_trainstate_world2_equals_the_full_batch_step(tmp_path):
  import torch
  o1, o2 = str(tmp_path / 'w1.pt'), str(tmp_path / 'w2.pt')
  r1 = _launch(1, o1, _port())
  assert r1[0][0] == 0, r1[0][1][-3000:]
  r2 = _launch(2, o2, _port())
  assert all(rc == 0 for rc, _ in r2), '\n'.join(o[-2000:] for _, o in r2)
  a = torch.load(o1 + '.rank0', weights_only=True)
  b, b1 = torch.load(o2 + '.rank0', weights_only=True), torch.load(o2 + '.rank1', weights_only=True)
  # replicas are IDENTICAL after three steps: same reduced gradients, a fixed-order gradient norm (no float atomics) -> same clip, same update
  assert torch.equal(b['flat'], b1['flat']) and torch.equal(b['m'], b1['m']) and b['losses'] == b1['losses'] and b['gn'] == b1['gn']
  diff = float((a['flat'] - b['flat']).abs().max())
  print('world 1 vs world 2: max |param diff|', diff, 'losses', a['losses'], b['losses'], 'grad norms', a['gn'], b['gn'])
  assert diff < 3e-5   # three AdamW steps at lr 1e-2 on fp32 buffers; gradients differ by summation order only
  for x, y in zip(a['losses'], b['losses']):
    assert abs(x - y) <= 1e-5 * abs(x)
  assert abs(a['gn'] - b['gn']) <= 1e-4 * abs(a['gn'])


@pytest.mark.parametrize('prec,batch,chunk,det', [('fp32', 4, 1, 0), ('fp32', 6, 2, 0), ('fp16', 6, 2, 0), ('fp32', 6, 2, 1), ('fp16', 6, 2, 1)])
def test_hip_trainstate_world2_several_chunks_overlap_on_and_off(tmp_path, prec, batch, chunk, det):
  """Several sample chunks per rank, so parameter gradients ACCUMULATE across chunks and the segment events of the overlapped all-reduce (include/spa3d.h,
  spa3d_set_grad_events) must fire in the last chunk only.  (4, 1): B_local = 2 as 1 + 1.  (6, 2): B_local = 3 as 1 + 2 -- the RAGGED chunk runs first, the
  last chunk is a full one (csrc/model.hip run_body).  fp16: the loss-scaled buffer is unscaled per segment right before the segment's event, so BASELINE
  configs[4] overlaps too.  Within a run both replicas are IDENTICAL; overlap on and off are two runs, whose parameter gradients differ in the last bits (dW /
  broadcast-gradient sums use float atomics), and both equal the single-process step on the whole batch."""
  import torch
  outs = {}
  for tag, world, env in (('full', 1, {}), ('ov1', 2, {'SPA3D_DP_OVERLAP': '1'}), ('ov0', 2, {'SPA3D_DP_OVERLAP': '0'})):
    o = str(tmp_path / (tag + '.pt'))
    r = _launch(world, o, _port(), BATCH=str(batch), SPA3D_CHUNK=str(chunk), PREC=prec, SPA3D_DET_GRADS=str(det), **env)
    assert all(rc == 0 for rc, _ in r), '\n'.join(x[-2000:] for _, x in r)
    outs[tag] = [torch.load(o + '.rank%d' % k, weights_only=True) for k in range(world)]
  assert outs['ov1'][0]['overlap'] and not outs['ov0'][0]['overlap']
  for tag in ('ov1', 'ov0'):
    assert torch.equal(outs[tag][0]['flat'], outs[tag][1]['flat']) and torch.equal(outs[tag][0]['m'], outs[tag][1]['m'])
  d01 = float((outs['ov1'][0]['flat'] - outs['ov0'][0]['flat']).abs().max())
  diff = float((outs['full'][0]['flat'] - outs['ov1'][0]['flat']).abs().max())
  print(f'{prec}, {batch} samples as 2 ranks x chunks of {chunk} vs one process: overlap on vs off max |param diff| {d01:.3e}; vs the full batch {diff:.3e}')
  if det:  # det_grads (include/spa3d.h): order-independent gradient sums -> the overlapped and the stream-ordered all-reduce give the SAME bits (round 4 could only ask for 1e-5)
    assert torch.equal(outs['ov1'][0]['flat'], outs['ov0'][0]['flat']) and outs['ov1'][0]['losses'] == outs['ov0'][0]['losses'], d01
  if prec == 'fp32':
    assert d01 < 1e-5 and all(abs(x - y) <= 1e-6 * abs(x) for x, y in zip(outs['ov1'][0]['losses'], outs['ov0'][0]['losses']))
    assert diff < 3e-5
  else:  # 16-bit activations: an AdamW step at lr 1e-2 moves a parameter by <= 1e-2 whatever the gradient's size, so last-bit gradient noise of tiny leaves shows
    assert d01 < 2e-2 and all(abs(x - y) <= 1e-3 * abs(x) for x, y in zip(outs['ov1'][0]['losses'], outs['ov0'][0]['losses']))
    assert diff < 3e-2
