"""Per-shape time of the tiled GEMMs inside one real train step (B per GPU from argv, default 64): uses the library's
event profiler plus the debug dump (cls, ms, flops, bytes, M, N, K, flags)."""
import ctypes as C, sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
from bench import synth_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda', 0)
model = spa3d.TrackAutoEncoder3D(num_output_frames=150, dino_feature_dim=768, depth_feature_dim=1, precision='bf16')
batch = synth_batch(B, 2048, 512, 150, 768, 1, dev, seed=1)
params = model.init(0, batch)['params']
state = spa3d.TrainState(model, params, learning_rate=1e-4, warmup_steps=10, total_steps=1000)
lib = spa3d._lib.load()
state.train_step(batch); torch.cuda.synchronize()
h = model._handle(768, 1)[0]
lib.spa3d_prof_enable(h, 1)
state.train_step(batch); torch.cuda.synchronize()
lib.spa3d_prof_dump.restype = C.c_int; lib.spa3d_prof_dump.argtypes = [C.c_void_p, C.c_char_p]
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'step_shapes.csv')
os.makedirs(os.path.dirname(path), exist_ok=True)
assert lib.spa3d_prof_dump(h, path.encode()) == 0
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for line in open(path):
  cls, ms, fl, by, m, n, k, fl2 = line.strip().split(',')
  a = agg[(int(cls), int(m), int(n), int(k), int(fl2))]
  a[0] += 1; a[1] += float(ms); a[2] += float(fl); a[3] += float(by)
tot = sum(a[1] for a in agg.values())
names = {0: 'NT', 1: 'TN', 2: 'generic', 3: 'attn_fwd', 4: 'attn_bwd', 5: 'ln_fwd', 6: 'ln_bwd', 7: 'attn_q1'}
print(f'total profiled ms {tot:.1f}')
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
  cls, m, n, k, f = key
  print(f'{names[cls]:8s} M={m:8d} N={n:5d} K={k:6d} fl={f:4d}  x{a[0]:4d} {a[1]:8.2f} ms {100*a[1]/tot:5.1f}%  {a[2]/a[1]/1e9:7.1f} TF/s {a[3]/a[1]/1e6:7.1f} GB/s')
