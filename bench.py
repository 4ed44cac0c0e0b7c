#!/usr/bin/env python3
"""bench.py -- train-step tracks/sec of the 3DSPA hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = forward + compute_loss_3d + backward + (RCCL gradient all-reduce) + clip/AdamW over one synthetic
batch resident in HBM.  Workload at every N: BASELINE.json configs[2] per GPU (B=64, 2048 support + 512 query,
T=150, xyz+depth+DINOv2-768, bf16) => weak scaling; configs[3] is exactly this at N=8 (global B=512).
Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

# algorithmic work of the reference graph per sample at N=2048,Q=512,T=150,C=772 (SURVEY 0.4 / BASELINE.md 3):
F_REF_FWD_PER_STEP_B64 = 598.6e12  # forward FLOPs at B=64
PEAK_BF16_FLOPS = 2.5e15  # dense bf16 MFMA, MI355X_MICROARCH.md chip table
PEAK_HBM = 8.0e12


def synth_batch(B, N, Q, T, dino_dim, depth_dim, device, seed):
  """random-walk tracks in [0,1]^3, Bernoulli(0.9) visibility, boundary_frame=T, depth=z,
  DINO ~ N(0,1) bf16, query point = (random frame, position of the query track at that frame)."""
  g = torch.Generator(device=device).manual_seed(seed)

  def walk(n):
    x0 = torch.rand(B, n, 1, 3, generator=g, device=device)
    steps = 0.01 * torch.randn(B, n, T, 3, generator=g, device=device)
    return torch.clamp(x0 + torch.cumsum(steps, dim=2), 0.0, 1.0)

  sup = walk(N)
  qt = walk(Q)
  tq = torch.randint(0, T, (B, Q), generator=g, device=device)
  xyz = torch.gather(qt, 2, tq[:, :, None, None].expand(B, Q, 1, 3))[:, :, 0]
  batch = {
      'support_tracks': sup,
      'support_tracks_visible': (torch.rand(B, N, T, 1, generator=g, device=device) < 0.9).float(),
      'query_points': torch.cat([tq[..., None].float(), xyz], dim=-1),
      'boundary_frame': torch.full((B,), T, dtype=torch.int32, device=device),
      'query_tracks': qt,
      'query_tracks_visible': (torch.rand(B, Q, T, 1, generator=g, device=device) < 0.9).float(),
  }
  if depth_dim:
    batch['depth_features'] = sup[..., 2:3].expand(B, N, T, depth_dim).to(torch.bfloat16).contiguous()
  if dino_dim:
    d = torch.empty(B, N, T, dino_dim, dtype=torch.bfloat16, device=device)
    for b in range(B):  # 0.47 GB per sample; generated in place
      d[b] = torch.randn(N, T, dino_dim, generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)
    batch['dino_features'] = d
  return batch


def pmc_traffic(kernel_class: str, launches_per_step: float):
  """HBM traffic per launch of the dominant kernel class from the committed rocprofv3 --pmc passes of this same command
  (profiles/r01_bench_b64_pmc_{fetch,write}.csv: separate passes, KiB units, FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM").
  Returns (bytes_per_launch, source) or (None, None) when the summaries are absent or the class is unknown."""
  import csv
  prefix = {'gemm_nt_bf16': 'gemm_nt', 'gemm_tn_bf16': 'gemm_tn', 'attention_fused_bwd': 'attn_bwd', 'attention_fused_fwd': 'attn_fwd',
            'gemm_generic': 'gemm_generic'}.get(kernel_class.split(' ')[0])
  f_fetch = os.path.join(ROOT, 'profiles', 'r01_bench_b64_pmc_fetch.csv'); f_write = os.path.join(ROOT, 'profiles', 'r01_bench_b64_pmc_write.csv')
  if prefix is None or not (os.path.exists(f_fetch) and os.path.exists(f_write)):
    return None, None
  def total(path, counter):
    t, n = 0.0, 0
    for r in csv.DictReader(open(path)):
      if r['counter'] == counter and r['kernel'].startswith(prefix):
        t += float(r['total']); n += int(r['dispatches'])
    return t, n
  fetch, n = total(f_fetch, 'FETCH_SIZE'); write, _ = total(f_write, 'WRITE_SIZE')
  if n == 0:
    return None, None
  return (2.0 * fetch + write) * 1024.0 / n, 'profiles/r01_bench_b64_pmc_fetch.csv + _write.csv (rocprofv3 --pmc, bench.py --steps 1 --warmup 0)'


def cpu_baseline():
  """The CPU restatement of the reference graph (oracle, kind "port") timed on this host: BASELINE.json configs[0]
  (B=2, 64+16 tracks, T=24, xyz-only, fp32), one fwd+bwd step.  The reference's own JAX path cannot run here
  (SURVEY F2/F3)."""
  from oracle import spa3d_oracle as O
  # the box's CPU share, not the host's core count: os.cpu_count() over-subscribes a cgroup-limited container
  try:
    cores = len(os.sched_getaffinity(0))
  except AttributeError:
    cores = os.cpu_count() or 1
  cores = max(1, min(cores, int(os.environ.get('SPA3D_CPU_THREADS', 16))))
  torch.set_num_threads(cores)
  cfg = O.Config(num_output_frames=24, use_dino=False, use_depth=False)
  p = O.init_params(cfg, seed=0, with_dino=False, with_depth=False)
  b = O.synthetic_batch(2, 64, 16, 24)
  m = O.TrackAutoEncoder3D(cfg)
  noise = torch.rand(2, 128, 96)
  O.loss_and_grads(m, p, b, noise=noise)  # warm-up
  ts = []
  for _ in range(2):
    t0 = time.perf_counter()
    O.loss_and_grads(m, p, b, noise=noise)
    ts.append(time.perf_counter() - t0)
  t = min(ts)
  return {'value': 160.0 / t, 'unit': 'tracks/s', 'cores': torch.get_num_threads(), 'kind': 'port',
          'sample': f'cfg#1 B=2, 64 support+16 query, T=24, xyz-only fp32, 1 fwd+bwd step (best of 2 after 1 warm-up): {t:.2f} s/step, '
                    f'{1.02e12 / t / 1e9:.1f} GFLOP/s of F_ref=1.02 TFLOP'}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=3)
  ap.add_argument('--warmup', type=int, default=1)
  ap.add_argument('--batch', type=int, default=int(os.environ.get('SPA3D_BENCH_B', 64)), help='per-GPU batch (default 64)')
  ap.add_argument('--support', type=int, default=2048)
  ap.add_argument('--query', type=int, default=512)
  ap.add_argument('--frames', type=int, default=150)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  args = ap.parse_args()

  rank = int(os.environ.get('RANK', 0))
  local_rank = int(os.environ.get('LOCAL_RANK', 0))
  world = int(os.environ.get('WORLD_SIZE', 1))
  if world != args.gpus:
    print(f'warning: WORLD_SIZE={world} but --gpus {args.gpus}', file=sys.stderr)
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)
  if world > 1:
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', device_id=dev)

  import spa3d
  B, N, Q, T = args.batch, args.support, args.query, args.frames
  model = spa3d.TrackAutoEncoder3D(num_output_frames=T, dino_feature_dim=768, depth_feature_dim=1, precision='bf16')
  batch = synth_batch(B, N, Q, T, 768, 1, dev, seed=1234 + rank)
  params = model.init(0, batch)['params']  # same seed on every rank: replicas start identical
  state = spa3d.TrainState(model, params, learning_rate=1e-4, warmup_steps=10000, total_steps=1000000)
  lib = spa3d._lib.load()

  def sync():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  for _ in range(args.warmup):
    state.train_step(batch)
  sync()
  prof = hasattr(lib, 'spa3d_prof_enable') and os.environ.get('SPA3D_BENCH_PROF', '1') == '1'
  h = model._handle(768, 1)[0]
  if prof:
    lib.spa3d_prof_enable(h, 1)
  t0 = time.perf_counter()
  for _ in range(args.steps):
    metrics = state.train_step(batch)
  sync()
  dt = time.perf_counter() - t0
  roof = None
  if prof:
    roof = spa3d.profile_summary(model, h)
    lib.spa3d_prof_enable(h, 0)
    if roof is not None:
      roof['traffic'], roof['traffic_source'] = pmc_traffic(roof['kernel'], roof['launches'] / max(1, args.steps))
      roof['algorithmic_bytes_per_launch'] = next(c['bytes'] / max(1, c['launches']) for c in roof['classes'] if c['kernel'] == roof['kernel'])
  tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
  if world > 1:
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
  dt = float(tmax.item())
  loss = float(metrics['train/loss'])

  if rank == 0:
    tracks = world * B * (N + Q) * args.steps
    ms = dt / args.steps * 1e3
    scale = (B / 64.0) * (N / 2048.0) * (T / 150.0)  # F_ref scales ~linearly in B; other dims only for dev runs
    out = {
        'metric': 'train-step tracks/sec (B x N_tracks) at T=150, C=772', 'value': tracks / dt, 'unit': 'tracks/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': f'BASELINE configs[2] per GPU: B={B}, {N} support + {Q} query, T={T}, xyz+depth(1)+DINOv2-768, '
                               'fwd+loss+bwd+clip+AdamW', 'per_gpu_batch': B, 'global_batch': B * world, 'support': N, 'query': Q,
                   'frames': T, 'channels': 772, 'parallelism': f'dp{world}', 'chunk_samples': int(os.environ.get('SPA3D_CHUNK', 0)),
                   'final_loss': loss},
        'step_mfma_frac_F_ref': (3 * F_REF_FWD_PER_STEP_B64 * scale / (ms / 1e3)) / PEAK_BF16_FLOPS,
        'roofline': roof,
    }
    if world == 1 and not args.no_cpu_baseline:
      out['cpu_baseline'] = cpu_baseline()
    print(json.dumps(out), flush=True)
  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
