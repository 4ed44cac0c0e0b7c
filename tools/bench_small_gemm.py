"""The small GEMMs of the latent stacks (M = 128 latents x samples of a chunk): dW (TN) and forward (NT) at the step's shapes, standalone, product dispatch.
    python tools/bench_small_gemm.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
if os.environ.get('SPA3D_TOOL_LIB'): spa3d._lib.LIB_PATH = os.environ['SPA3D_TOOL_LIB']
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 26, dtype=torch.uint8, device='cuda')
IMPL = int(os.environ.get('IMPL', 0))   # gemm_impl of the forward / dX calls (3 = every eligible NT GEMM on the 8-phase kernels whatever M)
def timeit(f, n=50):
  assert f() == 0; torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): f()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in ((1408, 768, 512), (1408, 768, 1152), (1408, 2048, 1152), (1408, 1152, 2048), (1408, 512, 768), (1408, 2304, 1152), (22528, 768, 384), (5632, 1280, 12352), (1152, 2048, 1152)):
  A = torch.randn(M, K, device='cuda').bfloat16(); W = (torch.randn(K, N, device='cuda') / K ** 0.5).bfloat16(); dC = torch.randn(M, N, device='cuda').bfloat16()
  dB = torch.zeros(K, N, device='cuda'); Cd = torch.empty(M, N, device='cuda', dtype=torch.bfloat16); dA = torch.empty(M, K, device='cuda', dtype=torch.bfloat16)
  t_tn = timeit(lambda: lib.spa3d_op_linear_bwd(A.data_ptr(), W.data_ptr(), dC.data_ptr(), None, dB.data_ptr(), None, M, N, K, 1, 0, ws.data_ptr(), ws.numel(), s()))
  t_nt = timeit(lambda: lib.spa3d_op_linear(A.data_ptr(), W.data_ptr(), None, None, Cd.data_ptr(), M, N, K, 0, 1, IMPL, ws.data_ptr(), ws.numel(), s()))
  t_dx = timeit(lambda: lib.spa3d_op_linear_bwd(A.data_ptr(), W.data_ptr(), dC.data_ptr(), dA.data_ptr(), None, None, M, N, K, 1, IMPL, ws.data_ptr(), ws.numel(), s()))
  ref = A.float().T @ dC.float()
  lib.spa3d_op_linear_bwd(A.data_ptr(), W.data_ptr(), dC.data_ptr(), None, dB.data_ptr(), None, M, N, K, 1, 0, ws.data_ptr(), ws.numel(), s())
  err = float((dB - ref).norm() / ref.norm())
  fl = 2.0 * M * N * K
  print(f'M={M:6d} N={N:5d} K={K:6d}: dW {t_tn:7.1f} us ({fl / t_tn / 1e6:6.1f} TF/s, rel err {err:.1e})   forward {t_nt:7.1f} us ({fl / t_nt / 1e6:6.1f} TF/s)   dX {t_dx:7.1f} us ({fl / t_dx / 1e6:6.1f} TF/s)'
        f'   (op times include a zero-fill / a transpose of the weight)', flush=True)
