"""Two dW (TN) shapes, a few launches each: target for rocprofv3 --pmc passes on the 8-phase TN kernels."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 28, dtype=torch.uint8, device='cuda')
for (M, N, K) in ((726528, 1536, 1280), (2473984, 768, 384)):
  A = torch.randn(M, K, device='cuda').bfloat16(); dC = torch.randn(M, N, device='cuda').bfloat16()
  dB = torch.empty(K, N, device='cuda'); Bd = torch.empty(K, N, device='cuda', dtype=torch.bfloat16)
  for _ in range(3):
    assert lib.spa3d_op_linear_bwd(A.data_ptr(), Bd.data_ptr(), dC.data_ptr(), None, dB.data_ptr(), None, M, N, K, 1, 2, ws.data_ptr(), ws.numel(), s()) == 0
  torch.cuda.synchronize()
  del A, dC
