"""HBM streaming rates on this box with library kernels (torch fill / copy / read-reduce): the practical ceilings the memory-bound
classes are compared with (peak 8 TB/s; bench.py prices against the peak, NOTEBOOK.md quotes these beside it)."""
import torch
def timeit(fn, n=10):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
for gb in (2, 8):
  n = gb * (1 << 30) // 2
  x = torch.empty(n, dtype=torch.bfloat16, device='cuda'); y = torch.empty_like(x)
  x.normal_()
  t_w = timeit(lambda: y.zero_())
  t_c = timeit(lambda: y.copy_(x))
  t_r = timeit(lambda: x.view(torch.int16).max())
  t_a = timeit(lambda: torch.add(x, x, out=y))
  print(f'{gb} GB: write-only {2*n/t_w/1e6:7.0f} GB/s   copy (1R+1W) {4*n/t_c/1e6:7.0f} GB/s   read-only {2*n/t_r/1e6:7.0f} GB/s   x+x (1R+1W) {4*n/t_a/1e6:7.0f} GB/s', flush=True)
  del x, y
