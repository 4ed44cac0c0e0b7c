#!/usr/bin/env python3
"""bench.py -- train-step tracks/sec of the 3DSPA hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--config {1,2,3,5}] [--batch B]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Called WITHOUT a torchrun environment and --gpus N > 1, bench.py starts the N ranks itself: before anything touches a GPU it runs
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same args>`
as a CHILD process, relays its output and exits with its status.  A rank whose WORLD_SIZE differs from --gpus refuses to run.
--launch-check: every rank joins a gloo group (no GPU call), proves the N-rank communicator with an all-reduce of ones, rank 0 prints
one JSON line and all exit -- the launch path is testable on a CPU-only host (tests/test_bench_launch.py).

One step = forward + compute_loss_3d + backward + (RCCL gradient all-reduce) + clip/AdamW over one synthetic
batch resident in HBM.  Default workload at every N: BASELINE.json configs[2] per GPU (B=64, 2048 support + 512 query,
T=150, xyz+depth+DINOv2-768, bf16) => weak scaling; configs[3] is exactly this at N=8 (global B=512).
--config 2: configs[1] (xyz+depth only, C=4); --config 1: configs[0]'s shape (B=2, 64+16 tracks, T=24, xyz-only, fp32) on the GPU;
--config 5: configs[4], the stress shape (8192 support + 2048 query, T = T_out = 300, C=772, fp16) with --batch samples per GPU (default 8:
3.8 GB of input per sample).
Rank 0 prints ONE JSON line ("Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

# algorithmic work of the reference graph per sample at N=2048,Q=512,T=150,C=772 (SURVEY 0.4 / BASELINE.md 3):
F_REF_FWD_PER_STEP_B64 = {3: 598.6e12, 2: 587.0e12}  # forward FLOPs at B=64 (C=772 / C=4)
PEAK_BF16_FLOPS = 2.5e15  # dense bf16 MFMA, MI355X_MICROARCH.md chip table
PEAK_F32_FLOPS = 157.3e12
PEAK_HBM = 8.0e12

CONFIGS = {
    1: dict(B=2, N=64, Q=16, T=24, dino=0, depth=0, precision='fp32', name='BASELINE configs[0] shape on the GPU: B=2, 64 support + 16 query, T=24, xyz-only, fp32'),
    2: dict(B=64, N=2048, Q=512, T=150, dino=0, depth=1, precision='bf16', name='BASELINE configs[1] per GPU: B=64, 2048 support + 512 query, T=150, xyz+depth(1) (C=4)'),
    3: dict(B=64, N=2048, Q=512, T=150, dino=768, depth=1, precision='bf16', name='BASELINE configs[2] per GPU: B=64, 2048 support + 512 query, T=150, xyz+depth(1)+DINOv2-768'),
    5: dict(B=8, N=8192, Q=2048, T=300, dino=768, depth=1, precision='fp16', name='BASELINE configs[4] per GPU: 8192 support + 2048 query, T=300, xyz+depth(1)+DINOv2-768, fp16 (per-GPU batch unspecified upstream)'),
}

# profile class -> (bound, kernel-name prefixes in the rocprofv3 summaries under profiles/)
CLASS_INFO = {
    'gemm_nt_bf16': ('mfma', ('gemm_nt', 'mlp_fused_fwd', 'gemm_rs')), 'gemm_tn_bf16': ('mfma', ('gemm_tn',)), 'gemm_generic': ('mfma', ('gemm_generic',)),
    'attention_fused_fwd': ('hbm', ('attn_fwd', 'xattn_fwd', 'xattn_combine')), 'attention_fused_bwd': ('hbm', ('attn_bwd', 'xattn_dq_finish')),
    'layernorm_fwd': ('hbm', ('ln_fwd',)), 'layernorm_bwd': ('hbm', ('ln_bwd',)), 'attention_single_query': ('hbm', ('attn_q1',)),
    'embed': ('hbm', ()),  # the input-streaming stage north_star wants at the HBM roofline: live timing only (its kernels are shared with other classes)
}


def synth_batch(B, N, Q, T, dino_dim, depth_dim, device, seed, feat_dtype=torch.bfloat16):
  """random-walk tracks in [0,1]^3, Bernoulli(0.9) visibility, boundary_frame=T, depth=z,
  DINO ~ N(0,1), query point = (random frame, position of the query track at that frame)."""
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
    batch['depth_features'] = sup[..., 2:3].expand(B, N, T, depth_dim).to(feat_dtype).contiguous()
  if dino_dim:
    d = torch.empty(B, N, T, dino_dim, dtype=feat_dtype, device=device)
    for b in range(B):  # 0.47 GB per sample; generated in place
      d[b] = torch.randn(N, T, dino_dim, generator=g, device=device, dtype=torch.float32).to(feat_dtype)
    batch['dino_features'] = d
  return batch


def current_round():
  """The round this tree is being built in = 1 + the newest BENCH_rNN.json the driver has left at the repo root."""
  import re
  done = [int(m.group(1)) for m in (re.match(r'BENCH_r(\d+)\.json$', f) for f in os.listdir(ROOT)) if m]
  return (max(done) if done else 0) + 1


def pmc_tables():
  """Per-kernel FETCH_SIZE / WRITE_SIZE totals of the committed rocprofv3 --pmc passes of this same command (separate passes, KiB
  units; `tools/profile_round.sh`).  Newest round first.  Returns (rows, source, round tag, staleness warning or None)."""
  import csv
  import re
  tags = sorted({m.group(1) for m in (re.match(r'(r\d+)_bench_b64_pmc_fetch\.csv$', f) for f in os.listdir(os.path.join(ROOT, 'profiles'))) if m}, reverse=True)
  for tag in tags:
    f_fetch = os.path.join(ROOT, 'profiles', f'{tag}_bench_b64_pmc_fetch.csv'); f_write = os.path.join(ROOT, 'profiles', f'{tag}_bench_b64_pmc_write.csv')
    if os.path.exists(f_fetch) and os.path.exists(f_write):
      rows = list(csv.DictReader(open(f_fetch))) + list(csv.DictReader(open(f_write)))
      cur = current_round()
      warn = None if int(tag[1:]) >= cur else f'PMC tables are from round {int(tag[1:])}, this tree is round {cur}: traffic figures describe an older build'
      return rows, f'profiles/{tag}_bench_b64_pmc_fetch.csv + _write.csv (rocprofv3 --pmc, bench.py --steps 1)', tag, warn
  return [], None, None, None


def pmc_step_bytes(rows):
  """HBM bytes of ONE whole step over EVERY kernel of the PMC passes: (2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB / steps (same corrections as
  pmc_traffic).  The step's real bound: it moves ~200 x the algorithmic minimum through HBM (activations of every op round-trip)."""
  fetch = sum(float(r['total']) for r in rows if r['counter'] == 'FETCH_SIZE')
  write = sum(float(r['total']) for r in rows if r['counter'] == 'WRITE_SIZE')
  psteps = max([int(r['dispatches']) for r in rows if r['kernel'].startswith('adamw_kernel') and r['counter'] == 'FETCH_SIZE'] or [0])
  return (2.0 * fetch + write) * 1024.0 / psteps if psteps and fetch + write > 0 else None


def pmc_traffic(rows, prefixes):
  """HBM traffic PER STEP of a kernel class: sum over the class's kernels of (2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB, divided by the
  number of steps the PMC pass ran (= its AdamW dispatches) -- FETCH_SIZE doubled as MI355X_MICROARCH.md "HBM" prescribes for wide
  coalesced reads on gfx950 (an upper bound for narrower access shapes)."""
  fetch = write = 0.0
  psteps = 0
  for r in rows:
    if r['kernel'].startswith('adamw_kernel') and r['counter'] == 'FETCH_SIZE':
      psteps = int(r['dispatches'])
    if not any(r['kernel'].startswith(p) for p in prefixes):
      continue
    if r['counter'] == 'FETCH_SIZE':
      fetch += float(r['total'])
    elif r['counter'] == 'WRITE_SIZE':
      write += float(r['total'])
  return (2.0 * fetch + write) * 1024.0 / psteps if psteps and fetch + write > 0 else None


def roofline_from_profile(spa3d, model, handle, steps, peak_flops, pmc=True):
  """Live HIP-event timings of every instrumented kernel class (spa3d_prof_*), each priced against ITS roofline:
  MFMA classes against the dense MFMA peak, HBM classes against 8 TB/s, with the PMC traffic ratio where a committed pass has it."""
  raw = spa3d.profile_summary(model, handle, peak_flops)
  if raw is None:
    return None
  rows, src, tag, warn = pmc_tables() if pmc else ([], None, None, None)  # the committed PMC passes are of the headline workload only
  classes = []
  for c in raw['classes']:
    key = c['kernel'].split(' ')[0]
    bound, prefixes = CLASS_INFO.get(key, ('mfma', ()))
    if c['launches'] == 0:
      continue
    sec = c['ms'] * 1e-3
    alg = c['bytes'] / c['launches']
    step_traffic = pmc_traffic(rows, prefixes) if prefixes else None                      # bytes per step, whole class
    traffic = step_traffic * max(1, steps) / c['launches'] if step_traffic else None     # per launch, like `achieved`
    e = {'kernel': c['kernel'], 'bound': bound, 'launches': c['launches'], 'ms_per_step': round(c['ms'] / max(1, steps), 3),
         'avg_launch_ms': round(c['ms'] / c['launches'], 4)}
    if bound == 'mfma':
      e.update(achieved=round(c['flops'] / sec / 1e12, 2), peak=peak_flops / 1e12, unit='TFLOP/s', frac=round(c['flops'] / sec / peak_flops, 4))
    else:
      e.update(achieved=round(c['bytes'] / sec / 1e9, 1), peak=PEAK_HBM / 1e9, unit='GB/s', frac=round(c['bytes'] / sec / PEAK_HBM, 4))
    e['algorithmic_bytes_per_launch'] = round(alg)
    e['traffic'] = round(traffic) if traffic else None
    e['traffic_ratio'] = round(traffic / alg, 3) if traffic and alg else None
    classes.append(e)
  if not classes:
    return None
  dom = max((c for c in classes if not c['kernel'].startswith('embed')), key=lambda r: r['ms_per_step'])  # 'embed' overlaps the GEMM classes
  out = {k: dom[k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'launches', 'avg_launch_ms',
                             'algorithmic_bytes_per_launch', 'traffic_ratio')}
  out['traffic_source'] = src
  out['traffic_round'] = tag
  if warn:
    out['traffic_warning'] = warn
  out['hbm_bytes_per_step'] = pmc_step_bytes(rows) if rows else None
  out['classes'] = classes
  mfma_flops = sum(c['flops'] for c in raw['classes'] if CLASS_INFO.get(c['kernel'].split(' ')[0], ('mfma',))[0] == 'mfma')
  return out, mfma_flops


def _cgroup_cpu_quota():
  """CPUs this process may use by its cgroup's CPU bandwidth limit (v2 cpu.max, else v1 cfs quota / period); None = unlimited or unreadable."""
  try:
    q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
    return None if q == 'max' else float(q) / float(per)
  except (OSError, ValueError):
    pass
  try:
    q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read()); per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
    return None if q <= 0 else q / per
  except (OSError, ValueError):
    return None


def cpu_threads():
  """Threads for the CPU baseline = the host cores this process really has: min(scheduler affinity, cgroup CPU quota) -- os.cpu_count() over-subscribes a
  cgroup-limited container, a fixed cap under-states a large host.  Returns (threads, affinity, quota)."""
  try:
    aff = len(os.sched_getaffinity(0))
  except AttributeError:
    aff = os.cpu_count() or 1
  quota = _cgroup_cpu_quota()
  n = aff if quota is None else min(aff, max(1, int(quota)))
  if os.environ.get('SPA3D_CPU_THREADS'):
    n = min(n, int(os.environ['SPA3D_CPU_THREADS']))
  return max(1, n), aff, quota


def cpu_baseline():
  """The CPU restatement of the reference graph (oracle, kind "port") timed on this host: BASELINE.json configs[0]
  (B=2, 64+16 tracks, T=24, xyz-only, fp32), FULL step = fwd + loss + bwd + clip + AdamW, median of 5 after 1 warm-up
  (SURVEY 8(d)).  The reference's own JAX path cannot run here."""
  from oracle import spa3d_oracle as O
  nthr, aff, quota = cpu_threads()
  torch.set_num_threads(nthr)
  cfg = O.Config(num_output_frames=24, use_dino=False, use_depth=False)
  p = O.tree_flatten(O.init_params(cfg, seed=0, with_dino=False, with_depth=False))
  m_ = {k: torch.zeros_like(v) for k, v in p.items()}
  v_ = {k: torch.zeros_like(v) for k, v in p.items()}
  b = O.synthetic_batch(2, 64, 16, 24)
  model = O.TrackAutoEncoder3D(cfg)
  noise = torch.rand(2, 128, 96)
  ts = []
  for step in range(6):
    t0 = time.perf_counter()
    _, _, g = O.loss_and_grads(model, O.tree_unflatten(p), b, noise=noise)
    O.adamw_step(p, g, m_, v_, step, 1e-4)
    ts.append(time.perf_counter() - t0)
  t = statistics.median(ts[1:])
  return {'value': 160.0 / t, 'unit': 'tracks/s', 'cores': torch.get_num_threads(), 'kind': 'port', 'cpu_affinity': aff, 'cgroup_cpu_quota': quota,
          'sample': f'cfg#1 B=2, 64 support+16 query, T=24, xyz-only fp32, full step fwd+loss+bwd+clip+AdamW (median of 5 after 1 warm-up): '
                    f'{t:.2f} s/step, {1.02e12 / t / 1e9:.1f} GFLOP/s of F_ref=1.02 TFLOP'}


def gpu_same_shape(spa3d, dev):
  """The cfg#1 shape on the GPU (both precisions), so the GPU/CPU ratio also exists at IDENTICAL shape, not only across configs."""
  out = {}
  c = CONFIGS[1]
  for precision in ('fp32', 'bf16'):
    model = spa3d.TrackAutoEncoder3D(num_output_frames=c['T'], use_dino=False, use_depth=False, precision=precision)
    batch = synth_batch(c['B'], c['N'], c['Q'], c['T'], 0, 0, dev, seed=7)
    st = spa3d.TrainState(model, model.init(0, batch)['params'])
    st.train_step(batch); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
      t0 = time.perf_counter(); st.train_step(batch); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = statistics.median(ts)
    out[precision] = {'ms_per_step': round(t * 1e3, 3), 'tracks_per_s': round(160.0 / t, 1)}
    del st, model
  return out


def spawn_ranks(n: int, argv) -> int:
  """Start the n ranks of this same command under torch.distributed.run as a CHILD process (never an exec: the parent has not
  touched the GPU and does not need to, but a child keeps that true by construction), relay its output, return its exit status."""
  import socket
  import subprocess
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
         '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
  print('bench.py: launching', ' '.join(cmd), file=sys.stderr, flush=True)
  return subprocess.run(cmd, env=env).returncode


def launch_check(rank: int, world: int, gpus: int) -> int:
  """No GPU call anywhere: gloo group over 127.0.0.1, all-reduce of ones must equal the world size."""
  ok = True
  if world > 1:
    dist.init_process_group('gloo')
    t = torch.ones(1)
    dist.all_reduce(t)
    ok = int(t.item()) == world
    dist.barrier()
    dist.destroy_process_group()
  if rank == 0:
    print(json.dumps({'launch_check': bool(ok and world == gpus), 'n_gpus': world, 'ranks_verified': world if ok else 0,
                      'requested_gpus': gpus, 'backend': 'gloo'}), flush=True)
  return 0 if ok and world == gpus else 3


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=3)
  ap.add_argument('--warmup', type=int, default=1)
  ap.add_argument('--config', type=int, default=3, choices=(1, 2, 3, 5), help='BASELINE.json configs[N-1] (default 3 = the headline C=772 workload; 5 = the fp16 stress shape)')
  ap.add_argument('--batch', type=int, default=None, help='per-GPU batch override (dev runs)')
  ap.add_argument('--support', type=int, default=None)
  ap.add_argument('--query', type=int, default=None)
  ap.add_argument('--frames', type=int, default=None)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--launch-check', action='store_true', help='prove the N-rank launch path over gloo and exit before any GPU call')
  args = ap.parse_args()

  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:  # plain `python bench.py --gpus N`: start the ranks; nothing has touched a GPU
    sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
  rank = int(os.environ.get('RANK', 0))
  local_rank = int(os.environ.get('LOCAL_RANK', 0))
  world = int(os.environ.get('WORLD_SIZE', 1))
  if world != args.gpus:
    print(f'bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a number for a job of the wrong size', file=sys.stderr)
    sys.exit(2)
  if args.launch_check:
    sys.exit(launch_check(rank, world, args.gpus))
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)
  rccl_ranks = 1
  if world > 1:
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', device_id=dev)
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)  # the communicator really spans `world` ranks, each on its own GPU
    rccl_ranks = int(ones.item())
    if rccl_ranks != world:
      print(f'bench.py: RCCL all-reduce of ones gave {rccl_ranks}, expected {world}', file=sys.stderr)
      sys.exit(2)

  import spa3d
  cfg = dict(CONFIGS[args.config])
  B = args.batch or int(os.environ.get('SPA3D_BENCH_B', cfg['B']))
  N, Q, T = args.support or cfg['N'], args.query or cfg['Q'], args.frames or cfg['T']
  dino, depth, precision = cfg['dino'], cfg['depth'], cfg['precision']
  fdt = {'bf16': torch.bfloat16, 'fp16': torch.float16}.get(precision, torch.float32)
  model = spa3d.TrackAutoEncoder3D(num_output_frames=T, dino_feature_dim=max(dino, 1), depth_feature_dim=max(depth, 1), use_dino=dino > 0,
                                   use_depth=depth > 0, precision=precision, workspace_fraction=float(os.environ.get('SPA3D_WS_FRACTION', 0.80)))
  batch = synth_batch(B, N, Q, T, dino, depth, dev, seed=1234 + rank, feat_dtype=fdt)
  params = model.init(0, batch)['params']  # TrainState broadcasts rank 0's parameters: replicas start identical whatever the seed
  state = spa3d.TrainState(model, params, learning_rate=1e-4, warmup_steps=10000, total_steps=1000000)
  lib = spa3d._lib.load()

  def sync():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  for _ in range(args.warmup):
    state.train_step(batch)
  sync()
  prof = os.environ.get('SPA3D_BENCH_PROF', '1') == '1'
  h = model._handle(dino, depth)[0]
  if prof:
    lib.spa3d_prof_enable(h, 1)
  t0 = time.perf_counter()
  for _ in range(args.steps):
    metrics = state.train_step(batch)
  sync()
  dt = time.perf_counter() - t0
  peak = PEAK_BF16_FLOPS if precision in ('bf16', 'fp16') else PEAK_F32_FLOPS
  roof, mfma_flops = None, None
  if prof:
    r = roofline_from_profile(spa3d, model, h, args.steps, peak, pmc=(args.config == 3 and (B, N, Q, T) == (64, 2048, 512, 150)))
    lib.spa3d_prof_enable(h, 0)
    if r is not None:
      roof, mfma_flops = r
  # the two data-dependent savings the step time rests on: fraction of track-encoder token rows kept by the pruning, and distinct
  # (sample, query frame) slots per query of the shared readout rows -- at this synthetic distribution (Bernoulli(0.9) visibility, uniform frames)
  ps = (spa3d._lib.C.c_double * 4)()
  lib.spa3d_plan_stats(h, ps)
  plan = {'encoder_rows_kept_fraction': round(ps[0] / ps[1], 4) if ps[1] else 1.0, 'readout_slots_per_query': round(ps[2] / ps[3], 4) if ps[3] else 1.0}
  tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
  if world > 1:
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
  dt = float(tmax.item())
  loss = float(metrics['train/loss'])
  # N > 1 only, OUTSIDE the timed region: the same step with the gradient all-reduce in stream order behind the backward instead of under its last
  # chunk -- the A/B of the overlap ("record the A/B in the JSON when a node appears").  Two extra steps per rank.
  overlap_ab = None
  if world > 1 and state._overlap is not None:
    ov, state._overlap = state._overlap, None
    sync()
    t1 = time.perf_counter()
    for _ in range(2):
      state.train_step(batch)
    sync()
    t_off = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
    dist.all_reduce(t_off, op=dist.ReduceOp.MAX)
    state._overlap = ov
    overlap_ab = {'overlapped_ms_per_step': dt / args.steps * 1e3, 'stream_ordered_ms_per_step': float(t_off.item()) / 2 * 1e3, 'stream_ordered_steps': 2}

  if rank == 0:
    tracks = world * B * (N + Q) * args.steps
    ms = dt / args.steps * 1e3
    C = 3 + depth + dino
    out = {
        'metric': f'train-step tracks/sec (B x N_tracks) at T={T}, C={C}', 'value': tracks / dt, 'unit': 'tracks/s',
        'n_gpus': world, 'rccl_ranks': rccl_ranks, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': {'bf16': 'bf16', 'fp16': 'f16'}.get(precision, 'f32'), 'data': 'synthetic',
        'config': {'workload': cfg['name'] + ', fwd+loss+bwd+clip+AdamW' + (f' [overrides: B={B}, N={N}, Q={Q}, T={T}]' if (B, N, Q, T) != (cfg['B'], cfg['N'], cfg['Q'], cfg['T']) else ''),
                   'baseline_config': args.config, 'per_gpu_batch': B, 'global_batch': B * world, 'support': N, 'query': Q,
                   'frames': T, 'channels': C, 'parallelism': f'dp{world}', 'chunk_samples': int(os.environ.get('SPA3D_CHUNK', 0)),
                   'final_loss': loss, **plan,
                   'grad_allreduce': ('none (1 rank)' if world == 1 else ('RCCL SUM, overlapped with the last chunk\'s backward (3 segments)' if state._overlap is not None
                                                                            else 'RCCL SUM after the backward'))},
        'roofline': roof,
    }
    if overlap_ab:
      out['config']['grad_allreduce_overlap_ab'] = overlap_ab
    if args.config in F_REF_FWD_PER_STEP_B64:
      scale = (B / 64.0) * (N / 2048.0) * (T / 150.0)  # F_ref scales ~linearly in B; other dims only for dev runs
      out['step_mfma_frac_F_ref'] = (3 * F_REF_FWD_PER_STEP_B64[args.config] * scale / (ms / 1e3)) / peak  # reference graph, nothing pruned
    if mfma_flops:
      out['step_mfma_frac_executed'] = (mfma_flops / args.steps / (ms / 1e3)) / peak  # FLOPs the MFMA kernels actually ran (pruned last blocks)
    if roof and roof.get('hbm_bytes_per_step'):  # whole-step HBM traffic (PMC, committed pass of this command) over this run's step time
      out['hbm_bytes_per_step'] = roof['hbm_bytes_per_step']
      out['step_hbm_frac'] = roof['hbm_bytes_per_step'] / (ms / 1e3) / PEAK_HBM
    if world == 1 and not args.no_cpu_baseline:
      cb = cpu_baseline()
      if args.config != 1:
        del state, batch, params
        model._ws = None
        torch.cuda.empty_cache()
        cb['gpu_same_shape'] = gpu_same_shape(spa3d, dev)
        cb['gpu_same_shape']['note'] = 'cfg#1 shape, full train step on this GPU; CPU value above is the same shape'
      out['cpu_baseline'] = cb
    print(json.dumps(out), flush=True)
  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
