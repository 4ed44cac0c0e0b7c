"""The step's plain GEMM shapes on the vendor library (torch.matmul -> hipBLASLt / rocBLAS, bf16) beside this library's
8-phase kernels: a yardstick for the K-loop, not a product path (the product never calls the vendor library)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
def timeit(fn, n=8):
  fn(); fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
print(f'{"shape":34s} {"mine NT":>18s} {"lib x@W":>18s} {"lib linear":>18s}   |  {"mine dW":>18s} {"lib x^T@dy":>18s}', flush=True)
for (M, N, K) in ((726528, 2304, 1280), (726528, 1280, 2304), (726528, 1280, 1536), (726528, 1536, 1280), (726528, 1280, 768),
                  (3063672, 2304, 384), (3063672, 384, 1536), (3063672, 384, 2304), (3063672, 1536, 384), (3063672, 768, 384)):
  A = torch.randn(M, K, device='cuda').bfloat16(); W = (torch.randn(K, N, device='cuda') / K ** 0.5).bfloat16(); Wt = W.t().contiguous()
  Cc = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  mine = lambda: lib.spa3d_op_linear(A.data_ptr(), W.data_ptr(), None, None, Cc.data_ptr(), M, N, K, 0, 1, 2, ws.data_ptr(), ws.numel(), s())
  assert mine() == 0
  t0 = timeit(mine)
  t1 = timeit(lambda: torch.matmul(A, W, out=Cc))
  t2 = timeit(lambda: torch.nn.functional.linear(A, Wt))
  dC = torch.randn(M, N, device='cuda').bfloat16(); dB = torch.empty(K, N, device='cuda')
  dw = lambda: lib.spa3d_op_linear_bwd(A.data_ptr(), W.data_ptr(), dC.data_ptr(), None, dB.data_ptr(), None, M, N, K, 1, 2, ws.data_ptr(), ws.numel(), s())
  assert dw() == 0
  t3 = timeit(dw)
  t4 = timeit(lambda: torch.matmul(A.t(), dC))
  f = 2.0 * M * N * K
  r = lambda t: f'{t:7.3f} ms {f / t / 1e9:6.0f} TF'
  print(f'M={M:8d} N={N:5d} K={K:5d}        {r(t0)}  {r(t1)}  {r(t2)}   |  {r(t3)}  {r(t4)}', flush=True)
  del A, W, Wt, Cc, dC, dB
