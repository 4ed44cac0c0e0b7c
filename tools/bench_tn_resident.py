"""TN (dW) kernels with Infinity-Cache-resident operands (small M, repeated) vs streamed (large M): is the K-loop HBM-side or core bound?"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 28, dtype=torch.uint8, device='cuda')
def timeit(fn, n=20):
  assert fn() == 0; torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
for (N, K) in ((1536, 1280), (768, 384), (384, 1536)):
  for M in (16384, 32768, 65536, 262144, 726528):
    A = torch.randn(M, K, device='cuda').bfloat16(); dC = torch.randn(M, N, device='cuda').bfloat16()
    dB = torch.empty(K, N, device='cuda'); Bd = torch.empty(K, N, device='cuda', dtype=torch.bfloat16)
    f = lambda: lib.spa3d_op_linear_bwd(A.data_ptr(), Bd.data_ptr(), dC.data_ptr(), None, dB.data_ptr(), None, M, N, K, 1, 2, ws.data_ptr(), ws.numel(), s())
    ms = timeit(f)
    print(f'N={N:5d} Ki={K:5d} M={M:7d} operands {M*(N+K)*2/1e6:7.1f} MB  {ms:7.3f} ms {2*M*N*K/ms/1e9:7.1f} TF/s', flush=True)
    del A, dC
