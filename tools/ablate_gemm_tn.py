"""Diagnostic only: the 8-phase dW (TN) kernel with parts removed (SEPARATE libraries built with -DSPA3D_TN_ABL=mask, never the product; results of a masked
build are wrong): 1 half of the transposed LDS reads, 2 no MFMAs.  Tests whether the kernel is bound by LDS read bandwidth (NOTEBOOK.md, end of round 4).
    python tools/ablate_gemm_tn.py"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
b = importlib.import_module('3dspa_code_amd.build')
out = os.path.join(ROOT, 'tools', '_ablate'); os.makedirs(out, exist_ok=True)
b.build(verbose=False)
objs = [os.path.join(b.HERE, 'build', o) for o in sorted(os.listdir(os.path.join(b.HERE, 'build'))) if o.endswith('.o') and o != 'gemm_fast.o']
import torch
libs = {}
for m in (0, 1, 2, 3):
  ao = os.path.join(out, f'gemm_fast_tnabl{m}.o')
  subprocess.check_call([b._hipcc()] + b.FLAGS + ['-DSPA3D_ABLATION_BUILD', f'-DSPA3D_ABL_TN={m}', '-c', os.path.join(b.CSRC, 'gemm_fast.hip'), '-o', ao])
  lp = os.path.join(out, f'libspa3d_tnabl{m}.so')
  subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lp] + objs + [ao])
  libs[m] = C.CDLL(lp)
  libs[m].spa3d_op_linear_bwd.argtypes = [C.c_void_p] * 6 + [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 28, dtype=torch.uint8, device='cuda')
names = {0: 'full kernel', 1: 'half of the transposed reads', 2: 'no MFMA', 3: 'half reads + no MFMA'}
for (M, N, K) in ((3065160, 1536, 384), (3065160, 2304, 384), (726528, 1536, 1280)):
  A = torch.randn(M, K, device='cuda').bfloat16(); dC = torch.randn(M, N, device='cuda').bfloat16()
  dB = torch.empty(K, N, device='cuda'); Bd = torch.empty(K, N, device='cuda', dtype=torch.bfloat16)
  for m in (0, 1, 2, 3):
    f = lambda: libs[m].spa3d_op_linear_bwd(A.data_ptr(), Bd.data_ptr(), dC.data_ptr(), None, dB.data_ptr(), None, M, N, K, 1, 2, ws.data_ptr(), ws.numel(), s())
    assert f() == 0; torch.cuda.synchronize()
    ts = []
    for _ in range(5):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f'TN M={M} N={N} Ki={K} mask {m} {names[m]:32s} median {ts[2]:7.3f} ms  ({2.0 * M * N * K / ts[2] / 1e9:7.1f} TF/s)', flush=True)
  del A, dC
