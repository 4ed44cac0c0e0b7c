"""Square / large-K NT GEMM calibration (random operands): compares the NT kernels on shapes where tile prologue/epilogue is negligible."""
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
for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 8192), (16384, 8192, 8192), (528384, 2304, 1280), (528384, 1280, 4096)):
  for zero in (0, 1):
    A = torch.zeros(M, K, device='cuda', dtype=torch.bfloat16) if zero else torch.randn(M, K, device='cuda').bfloat16()
    B = torch.zeros(K, N, device='cuda', dtype=torch.bfloat16) if zero else (torch.randn(K, N, device='cuda') / K ** 0.5).bfloat16()
    Cc = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    f = lambda: lib.spa3d_op_linear(A.data_ptr(), B.data_ptr(), None, None, Cc.data_ptr(), M, N, K, 0, 1, 2, ws.data_ptr(), ws.numel(), s())
    assert f() == 0
    ms = timeit(f)
    print(f'M={M:7d} N={N:5d} K={K:5d} {"zeros " if zero else "random"} {ms:8.3f} ms {2*M*N*K/ms/1e9:8.1f} TF/s', flush=True)
    del A, B, Cc
