"""BASELINE.json configs[4] on ONE GPU: 8192 support + 2048 query tracks, T = T_out = 300 (S = 301), C = 772, fp16, full train step
(forward + loss + backward + clip + AdamW).  The config is quoted for 8 GPUs with no per-GPU batch; one sample's inputs are 3.8 GB,
so B per GPU defaults to 2 here (argv[1]).  Prints one JSON line with the step time, tracks/s and the live per-class rooflines
(fused S=301 attention forward / split-pass backward among them).
    python tools/stress_cfg5.py [B] [precision]"""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import spa3d

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
precision = sys.argv[2] if len(sys.argv) > 2 else 'fp16'
N, Q, T = 8192, 2048, 300
dev = torch.device('cuda', 0)
fdt = {'fp16': torch.float16, 'bf16': torch.bfloat16}[precision]
model = spa3d.TrackAutoEncoder3D(num_output_frames=T, dino_feature_dim=768, depth_feature_dim=1, precision=precision)
batch = bench.synth_batch(B, N, Q, T, 768, 1, dev, seed=1, feat_dtype=fdt)
state = spa3d.TrainState(model, model.init(0, batch)['params'])
lib = spa3d._lib.load()
m = state.train_step(batch); torch.cuda.synchronize()
print('warm-up step: loss', float(m['train/loss']), 'grad norm', float(m['train/grad_norm']), flush=True)
h = model._handle(768, 1)[0]
lib.spa3d_prof_enable(h, 1)
ts = []
steps = 3
for i in range(steps):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  m = state.train_step(batch)
  torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
  print(f'step {i}: {ts[-1]:.3f} s loss {float(m["train/loss"]):.3f} grad norm {float(m["train/grad_norm"]):.4e}', flush=True)
roof, mfma_flops = bench.roofline_from_profile(spa3d, model, h, steps, bench.PEAK_BF16_FLOPS, pmc=False)
t = statistics.median(ts)
assert all(map(lambda x: x == x and abs(x) < float('inf'), [float(m['train/loss']), float(m['train/grad_norm'])])), 'non-finite loss / gradient'
print(json.dumps({'workload': f'BASELINE configs[4] on 1 GPU: B={B}, {N} support + {Q} query, T={T}, C=772, {precision}, fwd+loss+bwd+clip+AdamW',
                  'ms_per_step': t * 1e3, 'tracks_per_s': B * (N + Q) / t, 'final_loss': float(m['train/loss']),
                  'step_mfma_frac_executed': mfma_flops / steps / t / bench.PEAK_BF16_FLOPS, 'classes': roof['classes']}))
