"""NT GEMM shapes of the step: the round-5 large-register-tile kernel (impl 10, csrc/gemm_ntb.hip) against the product dispatch (impl 0), standalone.
    python tools/bench_ntb.py [n_shapes]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
if os.environ.get('SPA3D_TOOL_LIB'): spa3d._lib.LIB_PATH = os.environ['SPA3D_TOOL_LIB']
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 28, dtype=torch.uint8, device='cuda')
# (M, N, K, bias, what)
shapes = [(3401728, 384, 1536, 0, 'dX MLP-in (track encoder)'), (3401728, 384, 2304, 0, 'dX q|k|v'), (3401728, 384, 768, 1, 'out-projection (no residual here)'),
          (726528, 1280, 1536, 0, 'readout dX / MLP-out'), (726528, 1280, 2304, 0, 'readout dX q|k|v'), (726528, 1536, 1280, 1, 'readout MLP-in (no gelu here)'),
          (726528, 2304, 1280, 0, 'readout q|k|v'), (726528, 768, 1280, 0, 'readout dX out-projection'), (726528, 1280, 768, 1, 'readout out-projection')]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
only = int(os.environ.get('ONLY', -1))
for (M, N, K, bias, what) in shapes:
  A = torch.randn(M, K, device='cuda').bfloat16(); W = (torch.randn(K, N, device='cuda') / K ** 0.5).bfloat16()
  b = torch.randn(N, device='cuda') if bias else None
  res = {}
  for impl in ((0, 10) if only < 0 else (only,)):
    Cd = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    f = lambda: lib.spa3d_op_linear(A.data_ptr(), W.data_ptr(), b.data_ptr() if bias else None, None, Cd.data_ptr(), M, N, K, 0, 1, impl, ws.data_ptr(), ws.numel(), s())
    rc = f(); assert rc == 0, rc
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); res[impl] = (ts[3], Cd)
  if only < 0:
    same = bool(torch.equal(res[0][1], res[10][1]))   # the WHOLE output, bit for bit (no atomics in either kernel: any difference is a race or an addressing bug)
    d = (res[0][1][:65536].float() - res[10][1][:65536].float()); err = float(d.norm() / res[0][1][:65536].float().norm())
    ref = (A[:4096].float() @ W.float()) + (b if bias else 0); e10 = float((res[10][1][:4096].float() - ref).norm() / ref.norm())
    print(f'NT M={M} N={N} K={K} bias={bias} [{what}]: product {res[0][0]:7.3f} ms ({2.0 * M * N * K / res[0][0] / 1e9:7.1f} TF/s)   large tile {res[10][0]:7.3f} ms '
          f'({2.0 * M * N * K / res[10][0] / 1e9:7.1f} TF/s)   x{res[0][0] / res[10][0]:.3f}   rel diff {err:.2e}  whole output identical: {same}  vs fp32 ref {e10:.2e}', flush=True)
  else:
    t = res[only][0]; print(f'NT M={M} N={N} K={K} impl {only}: {t:7.3f} ms ({2.0 * M * N * K / t / 1e9:7.1f} TF/s)', flush=True)
  del A, W
