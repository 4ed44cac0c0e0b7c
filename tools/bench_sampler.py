"""Throughput of the feature-producer kernels at the production shape (2048 tracks x 150 frames, DINOv2-base 37x37x768
patches of a 518x518 video) against their HBM roofline, next to the CPU oracle (vectorised NumPy) on a bounded sample."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import spa3d
from oracle import sampler_oracle as S

T, H, W, N, D = 150, 518, 518, 2048, 768
rng = np.random.default_rng(0)
dn = torch.randn(T, 37, 37, D, device='cuda')
dp = torch.rand(T, H, W, 1, device='cuda') * 9 + 0.1
tr = torch.stack([torch.rand(N, T, device='cuda') * W, torch.rand(N, T, device='cuda') * H], -1)


def timeit(fn, n=20):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n


res = {}
for name, fn, byts in (
    ('sample_dino_f32', lambda: spa3d.sample_dino_features_for_tracks(dn, tr, (T, H, W, 3)), N * T * D * 4 * 5),
    ('sample_dino_bf16_out', lambda: spa3d.sample_dino_features_for_tracks(dn, tr, (T, H, W, 3), out_dtype=torch.bfloat16), N * T * D * (16 + 2)),
    ('sample_depth', lambda: spa3d.sample_depth_features_for_tracks(dp, tr), N * T * (256 * 4 + 32)),
    ('lift_2d_to_3d', lambda: spa3d.lift_2d_to_3d(tr, dp), N * T * (16 + 12 + 8))):
  ms = timeit(fn)
  res[name] = {'ms': ms, 'points_per_s': N * T / ms * 1e3, 'alg_GBps': byts / ms / 1e6, 'frac_of_8TBps': byts / ms / 1e6 / 8000}
# CPU oracle on a bounded sample (64 tracks)
n = 64
dnc, trc = dn.cpu().numpy(), tr[:n].cpu().numpy()
t0 = time.perf_counter(); S.sample_dino_features_for_tracks(dnc, trc, (T, H, W, 3)); t1 = time.perf_counter()
res['cpu_oracle_sample_dino'] = {'points_per_s': n * T / (t1 - t0), 'sample': f'{n} tracks x {T} frames, vectorised NumPy restatement'}
print(json.dumps(res))
