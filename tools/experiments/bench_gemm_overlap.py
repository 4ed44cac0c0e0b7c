"""Does the epilogue's HBM write overlap with the K-loop when the A operand is cache-resident?  K=384, N=2304, M swept from
Infinity-Cache-resident (A = 50-200 MB) to streamed; SPA3D_NT_DBG=1 removes the epilogue (persistent kernel off: the flag lives in
the non-persistent 8-phase kernel)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 28, dtype=torch.uint8, device='cuda')
def timeit(fn, n=20):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
N, K = 2304, 384
for M in (65536, 131072, 262144, 1048576, 2473984):
  A = torch.randn(M, K, device='cuda').bfloat16()
  B = (torch.randn(K, N, device='cuda') / K ** 0.5).bfloat16()
  Cc = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  f = lambda: lib.spa3d_op_linear(A.data_ptr(), B.data_ptr(), None, None, Cc.data_ptr(), M, N, K, 0, 1, 2, ws.data_ptr(), ws.numel(), s())
  assert f() == 0
  ms = timeit(f)
  print(f'M={M:8d} A={M*K*2/1e6:7.1f} MB C={M*N*2/1e6:8.1f} MB  {ms:8.3f} ms {2*M*N*K/ms/1e9:8.1f} TF/s  out {M*N*2/ms/1e6:7.1f} GB/s', flush=True)
  del A, B, Cc
