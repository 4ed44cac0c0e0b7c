"""Fused sequence-resident MLP forward (csrc/mlp_fused.hip) against the two tiled GEMMs it replaces, at the track encoder's per-chunk
shape (M = 9 samples x 2048 tracks x 151 tokens x 0.9 kept), interleaved rounds in one process (cdna_hip_programming.md rule 24)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spa3d

lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(12 << 30, dtype=torch.uint8, device='cuda')
M = int(os.environ.get('M', (9 * 2048 * 151 * 9 // 10) // 128 * 128))
d, mlp = 384, 1536
na = torch.randn(M, d, device='cuda').bfloat16(); a = torch.randn(M, d, device='cuda').bfloat16()
w_in = (torch.randn(d, mlp, device='cuda') / d ** 0.5).bfloat16(); w_out = (torch.randn(mlp, d, device='cuda') / mlp ** 0.5).bfloat16()
b_in = torch.randn(mlp, device='cuda'); b_out = torch.randn(d, device='cuda')
y = torch.empty(M, d, device='cuda', dtype=torch.bfloat16); h = torch.empty(M, mlp, device='cuda', dtype=torch.bfloat16); hp = torch.empty_like(h)


def fused():
  assert lib.spa3d_op_mlp_fused(na.data_ptr(), a.data_ptr(), w_in.data_ptr(), b_in.data_ptr(), w_out.data_ptr(), b_out.data_ptr(), y.data_ptr(),
                                h.data_ptr(), hp.data_ptr(), M, d, mlp, 1, ws.data_ptr(), ws.numel(), s()) == 0


def pair():
  # impl 2 | 16: MLP-in with its second (pre-activation) output stream, as in the step
  assert lib.spa3d_op_linear(na.data_ptr(), w_in.data_ptr(), b_in.data_ptr(), None, h.data_ptr(), M, mlp, d, 1, 1, 18, ws.data_ptr(), ws.numel(), s()) == 0
  assert lib.spa3d_op_linear(h.data_ptr(), w_out.data_ptr(), b_out.data_ptr(), a.data_ptr(), y.data_ptr(), M, d, mlp, 0, 1, 2, ws.data_ptr(), ws.numel(), s()) == 0


def timeit(fn, n=5):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n


fused(); pair(); torch.cuda.synchronize()
fl = 2 * 2 * M * d * mlp
res = {'fused': [], 'pair': []}
for rnd in range(int(os.environ.get('ROUNDS', 6))):
  res['pair'].append(timeit(pair)); res['fused'].append(timeit(fused))
for k, v in res.items():
  v.sort(); med = v[len(v) // 2]
  byts = (M * (3 * d + 2 * mlp) + (M * mlp if k == 'pair' else 0)) * 2
  print(f'{k:6s} M={M} median {med:8.3f} ms  min {v[0]:8.3f} ms   {fl / med / 1e9:7.1f} TF/s   {byts / med / 1e6:7.1f} GB/s algorithmic')
print(f'ratio fused / pair = {sorted(res["fused"])[len(res["fused"]) // 2] / sorted(res["pair"])[len(res["pair"]) // 2]:.3f}')
