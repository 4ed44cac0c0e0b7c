"""dW (TN) shapes of the step: the round-5 large-register-tile kernel (gemm_impl 0 / 9) against the 8-wave kernels (gemm_impl 8), standalone.
    python tools/bench_tnb.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 28, dtype=torch.uint8, device='cuda')
shapes = [(3065160, 1536, 384), (3065160, 2304, 384), (3065160, 768, 384), (3065160, 384, 1536), (3065160, 384, 768), (726528, 1536, 1280), (726528, 1280, 1536),
          (726528, 2304, 1280), (726528, 1280, 768)]
if len(sys.argv) > 1:
  shapes = shapes[:int(sys.argv[1])]
for (M, N, K) in shapes:
  A = torch.randn(M, K, device='cuda').bfloat16(); dC = torch.randn(M, N, device='cuda').bfloat16()
  dB = torch.empty(K, N, device='cuda'); Bd = torch.empty(K, N, device='cuda', dtype=torch.bfloat16)
  res = {}
  for impl in (8, 0):
    f = lambda: lib.spa3d_op_linear_bwd(A.data_ptr(), Bd.data_ptr(), dC.data_ptr(), None, dB.data_ptr(), None, M, N, K, 1, impl, ws.data_ptr(), ws.numel(), s())
    assert f() == 0; torch.cuda.synchronize()
    ts = []
    for _ in range(7):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); res[impl] = (ts[3], dB.clone())
  err = float((res[0][1] - res[8][1]).norm() / res[8][1].norm())
  print(f'TN M={M} N={N} Ki={K}: 8-wave {res[8][0]:7.3f} ms ({2.0 * M * N * K / res[8][0] / 1e9:7.1f} TF/s)   large tile {res[0][0]:7.3f} ms ({2.0 * M * N * K / res[0][0] / 1e9:7.1f} TF/s)'
        f'   x{res[8][0] / res[0][0]:.3f}   rel diff {err:.2e}', flush=True)
  del A, dC
