"""LayerNorm forward / backward at the step's two large shapes (bf16): achieved GB/s against algorithmic bytes."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
if os.environ.get("SPA3D_LIB"): spa3d._lib.LIB_PATH = os.environ["SPA3D_LIB"]
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=10):
  assert fn() == 0; torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
for (M, D) in ((3401728, 384), (726528, 1280), (3401728 // 8, 384)):
  x = torch.randn(M, D, device='cuda').bfloat16(); sc = torch.ones(D, device='cuda'); y = torch.empty_like(x); st = torch.empty(M, 2, device='cuda')
  dy = torch.randn(M, D, device='cuda').bfloat16(); dx = torch.empty_like(x); dsc = torch.zeros(D, device='cuda')
  f = lambda: lib.spa3d_op_layernorm(x.data_ptr(), sc.data_ptr(), y.data_ptr(), st.data_ptr(), M, D, 1, s())
  b = lambda: lib.spa3d_op_layernorm_bwd(x.data_ptr(), sc.data_ptr(), st.data_ptr(), dy.data_ptr(), dx.data_ptr(), dsc.data_ptr(), M, D, 1, s())
  tf, tb = timeit(f), timeit(b)
  print(f'M={M:8d} D={D:5d}  fwd {tf:7.3f} ms {2*M*D*2/tf/1e6:7.0f} GB/s   bwd {tb:7.3f} ms {3*M*D*2/tb/1e6:7.0f} GB/s', flush=True)
  del x, y, dy, dx
