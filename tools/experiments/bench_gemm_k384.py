"""The short-K encoder shapes (QKV / MLP-in), plain epilogue: target for epilogue experiments (SPA3D_NT_DBG)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 28, dtype=torch.uint8, device='cuda')
def timeit(fn, n=10):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
for (M, N, K) in ((2473984, 2304, 384), (2473984, 1536, 384), (528384, 2304, 1280), (528384, 1280, 768)):
  A = torch.randn(M, K, device='cuda').bfloat16()
  B = (torch.randn(K, N, device='cuda') / K ** 0.5).bfloat16()
  Cc = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  f = lambda: lib.spa3d_op_linear(A.data_ptr(), B.data_ptr(), None, None, Cc.data_ptr(), M, N, K, 0, 1, 2, ws.data_ptr(), ws.numel(), s())
  assert f() == 0
  ms = timeit(f)
  print(f'M={M:8d} N={N:5d} K={K:5d} {ms:8.3f} ms {2*M*N*K/ms/1e9:8.1f} TF/s  out {M*N*2/ms/1e6:7.1f} GB/s', flush=True)
  del A, B, Cc
