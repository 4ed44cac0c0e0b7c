"""Aggregate gpurun_out/step_shapes.csv (tools/step_shapes.py) by (class, N, K, flags): total ms and TF/s per shape class, all chunks of the step together."""
import collections, sys, os
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'step_shapes.csv')
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for line in open(path):
  cls, ms, fl, by, m, n, k, f = line.strip().split(',')
  if int(cls) > 1: continue
  a = agg[(int(cls), int(n), int(k), int(f) if int(cls) == 0 else 0)]
  a[0] += 1; a[1] += float(ms); a[2] += float(fl)
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
  if a[1] < 3: continue
  print(f'{"NT" if key[0] == 0 else "TN"} N={key[1]:5d} K={key[2]:5d} fl={key[3]:3d}  x{a[0]:4d} {a[1]:8.2f} ms  {a[2] / a[1] / 1e9:7.1f} TF/s')
