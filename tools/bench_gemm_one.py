"""One NT and one TN shape, few launches: target for rocprofv3 --pmc passes."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
M = 8 * 2048 * 151
for (N, K) in ((384, 2304), (2304, 384)):
  A = torch.randn(M, K, device='cuda').bfloat16(); B = (torch.randn(K, N, device='cuda') / K ** 0.5).bfloat16()
  Cc = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  for _ in range(3):
    lib.spa3d_op_linear(A.data_ptr(), B.data_ptr(), None, None, Cc.data_ptr(), M, N, K, 0, 1, 2, ws.data_ptr(), ws.numel(), s())
  dC = torch.randn(M, N, device='cuda').bfloat16(); dB = torch.empty(K, N, device='cuda')
  for _ in range(3):
    lib.spa3d_op_linear_bwd(A.data_ptr(), B.data_ptr(), dC.data_ptr(), None, dB.data_ptr(), None, M, N, K, 1, 2, ws.data_ptr(), ws.numel(), s())
  torch.cuda.synchronize()
  del A, B, Cc, dC, dB
