"""LayerNorm backward and the bias-gradient column sums at the latent stacks' sizes (1 408 rows), standalone.  SPA3D_TOOL_LIB selects another library."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
if os.environ.get('SPA3D_TOOL_LIB'): spa3d._lib.LIB_PATH = os.environ['SPA3D_TOOL_LIB']
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 26, dtype=torch.uint8, device='cuda')
def timeit(f, n=100):
  assert f() == 0; torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): f()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
for (M, D) in ((1408, 1152), (1408, 512), (1408, 768), (22528, 384), (3401728, 384)):
  x = torch.randn(M, D, device='cuda').bfloat16(); sc = torch.ones(D, device='cuda'); y = torch.empty_like(x); st = torch.empty(M, 2, device='cuda')
  dy = torch.randn(M, D, device='cuda').bfloat16(); dx = torch.empty_like(x); dsc = torch.zeros(D, device='cuda')
  assert lib.spa3d_op_layernorm(x.data_ptr(), sc.data_ptr(), y.data_ptr(), st.data_ptr(), M, D, 1, s()) == 0
  tb = timeit(lambda: lib.spa3d_op_layernorm_bwd(x.data_ptr(), sc.data_ptr(), st.data_ptr(), dy.data_ptr(), dx.data_ptr(), dsc.data_ptr(), M, D, 1, s()), 100 if M < 1e6 else 10)
  # reference of dscale from one call
  dsc.zero_(); lib.spa3d_op_layernorm_bwd(x.data_ptr(), sc.data_ptr(), st.data_ptr(), dy.data_ptr(), dx.data_ptr(), dsc.data_ptr(), M, D, 1, s())
  xf = x.float(); xh = (xf - xf.mean(1, keepdim=True)) * torch.rsqrt(xf.var(1, unbiased=False, keepdim=True) + 1e-6)
  ref = (dy.float() * xh).sum(0); err = float((dsc - ref).norm() / ref.norm())
  print(f'LayerNorm backward rows={M:8d} d={D:5d}: {tb:8.1f} us   dscale rel err {err:.1e}', flush=True)
for (M, N, K) in ((1408, 2048, 1152), (1408, 768, 512), (1408, 1152, 2048)):
  A = torch.randn(M, K, device='cuda').bfloat16(); W = torch.randn(K, N, device='cuda').bfloat16(); dC = torch.randn(M, N, device='cuda').bfloat16(); db = torch.zeros(N, device='cuda')
  dB = torch.zeros(K, N, device='cuda')
  t1 = timeit(lambda: lib.spa3d_op_linear_bwd(A.data_ptr(), W.data_ptr(), dC.data_ptr(), None, dB.data_ptr(), None, M, N, K, 1, 0, ws.data_ptr(), ws.numel(), s()))
  t2 = timeit(lambda: lib.spa3d_op_linear_bwd(A.data_ptr(), W.data_ptr(), dC.data_ptr(), None, dB.data_ptr(), db.data_ptr(), M, N, K, 1, 0, ws.data_ptr(), ws.numel(), s()))
  err = float((db - dC.float().sum(0)).norm() / dC.float().sum(0).norm())
  print(f'bias gradient rows={M} n={N}: dW alone {t1:7.1f} us, dW + column sums {t2:7.1f} us -> {t2 - t1:6.1f} us for the zero-fill + column sums   rel err {err:.1e}', flush=True)
