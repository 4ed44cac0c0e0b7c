"""Row-stationary K = 384 GEMM (csrc/gemm_rs.hip, spa3d_op_linear impl 7) against the tiled kernels (impl 6) at the step's shapes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
if os.environ.get('SPA3D_TOOL_LIB'): spa3d._lib.LIB_PATH = os.environ['SPA3D_TOOL_LIB']
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 28, dtype=torch.uint8, device='cuda')
def timeit(fn, n=8):
  assert fn() == 0; torch.cuda.synchronize()
  ts = []
  for _ in range(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
  ts.sort(); return ts[len(ts) // 2], ts[0]
M = int(os.environ.get('M', 3065160))
for N in (2304, 1536, 768):
  A = torch.randn(M, 384, device='cuda').bfloat16()
  B = (torch.randn(384, N, device='cuda') / 384 ** 0.5).bfloat16()
  bias = torch.randn(N, device='cuda')
  out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  for impl, name in ((7, 'row-stationary'), (6, 'tiled 8-phase ')):
    f = lambda: lib.spa3d_op_linear(A.data_ptr(), B.data_ptr(), bias.data_ptr(), None, out.data_ptr(), M, N, 384, 0, 1, impl, ws.data_ptr(), ws.numel(), s())
    med, mn = timeit(f)
    print(f'M={M} N={N:5d} K=384 {name} median {med:7.3f} ms  min {mn:7.3f} ms  {2.0 * M * N * 384 / med / 1e9:7.1f} TF/s  {(M * (384 + N) * 2.0) / med / 1e6:7.1f} GB/s', flush=True)
  del A, B, out

# the MLP backward's dh = (dy . W_out^T) o gelu'(hpre): act 2
N = 1536
A = torch.randn(M, 384, device='cuda').bfloat16(); B = (torch.randn(384, N, device='cuda') / 384 ** 0.5).bfloat16()
pre = torch.randn(M, N, device='cuda').bfloat16(); out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
for impl, name in ((7, 'row-stationary'), (6, 'tiled 8-phase ')):
  f = lambda: lib.spa3d_op_linear(A.data_ptr(), B.data_ptr(), None, pre.data_ptr(), out.data_ptr(), M, N, 384, 2, 1, impl, ws.data_ptr(), ws.numel(), s())
  med, mn = timeit(f)
  print(f"M={M} N={N:5d} K=384 gelu' epilogue {name} median {med:7.3f} ms  min {mn:7.3f} ms  {2.0 * M * N * 384 / med / 1e9:7.1f} TF/s", flush=True)
