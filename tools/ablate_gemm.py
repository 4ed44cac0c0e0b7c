"""Diagnostic only: the persistent NT GEMM with its output stores removed (a SEPARATE library built with -DSPA3D_ABLATE, never the
product) = the loop-only rate, i.e. the ceiling of any scheme that hides the stores under the next tile's K-loop.
(Round 2 also timed a persistent 128x256 instance this way -- the only tile whose finished rows fit in registers beside the next tile's
accumulators: loop-only 995 TF/s at the QKV shape against 799 TF/s for the shipped 256x256 kernel WITH its stores, see NOTEBOOK.md.)"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
b = importlib.import_module('3dspa_code_amd.build')
out = os.path.join(ROOT, 'tools', '_ablate'); os.makedirs(out, exist_ok=True)
b.build(verbose=False)
objs = [os.path.join(b.HERE, 'build', o) for o in sorted(os.listdir(os.path.join(b.HERE, 'build'))) if o.endswith('.o') and o != 'gemm_fast.o']
ao = os.path.join(out, 'gemm_fast_ablate.o')
subprocess.check_call([b._hipcc()] + b.FLAGS + ['-DSPA3D_ABLATION_BUILD', '-DSPA3D_ABL_NT=1', '-c', os.path.join(b.CSRC, 'gemm_fast.hip'), '-o', ao])
lib_path = os.path.join(out, 'libspa3d_ablate_gemm.so')
subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib_path] + objs + [ao])
import torch, spa3d
spa3d._lib.LIB_PATH = lib_path
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 28, dtype=torch.uint8, device='cuda')
def timeit(fn, n=10):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
M = 3401728
for (N, K, act, res) in ((2304, 384, 0, False), (1536, 384, 0, True), (768, 384, 0, False), (2304, 1280, 0, False)):
  A = torch.randn(M if K == 384 else M // 4, K, device='cuda').bfloat16(); Mm = A.shape[0]
  B = (torch.randn(K, N, device='cuda') / K ** 0.5).bfloat16()
  R = torch.randn(Mm, N, device='cuda').bfloat16() if res else None
  Cc = torch.empty(Mm, N, device='cuda', dtype=torch.bfloat16)
  f = lambda: lib.spa3d_op_linear(A.data_ptr(), B.data_ptr(), None, R.data_ptr() if res else None, Cc.data_ptr(), Mm, N, K, act, 1, 2, ws.data_ptr(), ws.numel(), s())
  for pp in ('1',):
    for abl in ('0', '1'):
      os.environ['SPA3D_ABLATE'] = abl
      assert f() == 0
      ms = timeit(f)
      print(f'M={Mm:8d} N={N:5d} K={K:5d} aux={int(res)} tile={"256x256" if pp == "1" else "128x256"} stores={"off" if abl == "1" else "on "} {ms:8.3f} ms {2*Mm*N*K/ms/1e9:8.1f} TF/s', flush=True)
  del A, B, Cc, R
