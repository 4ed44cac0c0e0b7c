"""Diagnostic only: the large-register-tile NT kernel (csrc/gemm_ntb.hip) with parts removed or re-placed.  SEPARATE libraries under tools/_ablate/ built with
-DSPA3D_ABLATION_BUILD -DSPA3D_ABL_NTB=mask (csrc/ablate.inc; results of a masked build are WRONG: timing only) and structure variants (-DSPA3D_NTB_*; correct results).
    python tools/ablate_ntb.py [variant ...]       variant = name:flag[,flag]       MASKS=1,2,3 restricts the masks"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
b = importlib.import_module('3dspa_code_amd.build')
out = os.path.join(ROOT, 'tools', '_ablate'); os.makedirs(out, exist_ok=True)
b.build(verbose=False)
objs = [os.path.join(b.HERE, 'build', o) for o in sorted(os.listdir(os.path.join(b.HERE, 'build'))) if o.endswith('.o') and o != 'gemm_ntb.o']
import torch
masks = [int(x) for x in os.environ.get('MASKS', '1,2,4,8,16,3,5,6,7,18,23').split(',') if x]
builds = [('product', [])]
builds += [(f'mask {m}', ['-DSPA3D_ABLATION_BUILD', f'-DSPA3D_ABL_NTB={m}']) for m in masks]
builds += [(v.split(':')[0], v.split(':')[1].split(',')) for v in sys.argv[1:]]
names = {1: 'no LDS-DMA in the loop', 2: 'no MFMA', 4: 'no fragment reads', 8: 'no barriers / waits', 16: 'no output stores'}
libs = []
for i, (name, flags) in enumerate(builds):
  ao = os.path.join(out, f'gemm_ntb_v{i}.o')
  subprocess.check_call([b._hipcc()] + b.FLAGS + flags + ['-c', os.path.join(b.CSRC, 'gemm_ntb.hip'), '-o', ao])
  lp = os.path.join(out, f'libspa3d_ntb_v{i}.so')
  subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lp] + objs + [ao])
  L = C.CDLL(lp)
  L.spa3d_op_linear.argtypes = [C.c_void_p] * 5 + [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
  if name.startswith('mask '):
    m = int(name.split()[1]); name = f'mask {m:2d}: ' + ' + '.join(names[k] for k in (1, 2, 4, 8, 16) if m & k)
  libs.append((name, L))
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 28, dtype=torch.uint8, device='cuda')
for (M, N, K) in ((3401728, 384, 1536), (726528, 1280, 1536)):
  A = torch.randn(M, K, device='cuda').bfloat16(); W = (torch.randn(K, N, device='cuda') / K ** 0.5).bfloat16()
  Cd = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  for name, L in libs:
    f = lambda: L.spa3d_op_linear(A.data_ptr(), W.data_ptr(), None, None, Cd.data_ptr(), M, N, K, 0, 1, 10, ws.data_ptr(), ws.numel(), s())
    assert f() == 0; torch.cuda.synchronize()
    ts = []
    for _ in range(7):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f'NT M={M} N={N} K={K}  {name:64s} median {ts[3]:7.3f} ms  ({2.0 * M * N * K / ts[3] / 1e9:7.1f} TF/s)', flush=True)
  del A, W, Cd
