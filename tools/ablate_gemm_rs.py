"""Diagnostic only: per-phase ablation of the row-stationary K = 384 GEMM (csrc/gemm_rs.hip).  One SEPARATE library per compile-time mask
(-DSPA3D_RS_ABLATE=mask), never the product: 1 stores into a 1-MiB window, 4 no LDS-DMA, 8 no MFMAs, 16 no staging / stores, 32 s_memtime stamps,
128 no counted waits, 256 no barriers (the last two: wrong results, timing only).  Extra -D flags for structure variants: VARIANT="-DX=1,-DY=2".
    python tools/ablate_gemm_rs.py 0,32,1,4,8,16      (N, M from the environment)"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
b = importlib.import_module('3dspa_code_amd.build')
out = os.path.join(ROOT, 'tools', '_ablate'); os.makedirs(out, exist_ok=True)
b.build(verbose=False)
objs = [os.path.join(b.HERE, 'build', o) for o in sorted(os.listdir(os.path.join(b.HERE, 'build'))) if o.endswith('.o') and o != 'gemm_rs.o']
masks = [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else '0,32,1,4,8,16'.split(','))]
extra = [x for x in os.environ.get('VARIANT', '').split(',') if x]
import torch
libs = {}
for m in masks:
  ao = os.path.join(out, f'gemm_rs_abl{m}.o')
  subprocess.check_call([b._hipcc()] + b.FLAGS + ['-DSPA3D_ABLATION_BUILD', f'-DSPA3D_ABL_RS={m}'] + extra + ['-c', os.path.join(b.CSRC, 'gemm_rs.hip'), '-o', ao])
  lp = os.path.join(out, f'libspa3d_rs_abl{m}.so')
  subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lp] + objs + [ao])
  libs[m] = C.CDLL(lp)
  libs[m].spa3d_op_linear.argtypes = [C.c_void_p] * 5 + [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 26, dtype=torch.uint8, device='cuda')
M = int(os.environ.get('M', 3065160)); N = int(os.environ.get('N', 2304))
A = torch.randn(M, 384, device='cuda').bfloat16(); B = (torch.randn(384, N, device='cuda') / 384 ** 0.5).bfloat16(); bias = torch.randn(N, device='cuda')
out_t = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
AUX = int(os.environ.get('AUX', 0)); pre = torch.randn(M, N, device='cuda').bfloat16() if AUX else None
def run(m):
  assert libs[m].spa3d_op_linear(A.data_ptr(), B.data_ptr(), None if AUX else bias.data_ptr(), pre.data_ptr() if AUX else None, out_t.data_ptr(), M, N, 384, 2 if AUX else 0, 1, 7, ws.data_ptr(), ws.numel(), s()) == 0
def timeit(fn, n=5):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
dbg = torch.zeros(256 * 4 * 4, dtype=torch.int64, device='cuda')
os.environ['SPA3D_RS_DBG'] = str(dbg.data_ptr())
for m in masks: run(m)
torch.cuda.synchronize()
if 0 in masks:  # correctness of the (variant) build on a row sample
  out_t.fill_(float('nan')); run(0); torch.cuda.synchronize()
  idx = torch.cat([torch.arange(0, 700, device='cuda'), torch.randint(0, M, (3000,), device='cuda'), torch.arange(M - 700, M, device='cuda')])
  ref = A[idx].float() @ B.float() + (0 if AUX else bias)
  if AUX:
    x = pre[idx].float(); c = 0.7978845608028654; u = c * (x + 0.044715 * x ** 3); t = torch.tanh(u)
    ref = ref * (0.5 * (1 + t) + 0.5 * x * (1 - t * t) * c * (1 + 3 * 0.044715 * x * x))
  err = (out_t[idx].float() - ref).abs().max().item(); nan = int(torch.isnan(out_t.float()).sum().item())
  print(f'check: max abs err on {idx.numel()} rows {err:.4f} (bf16 rounding ~ {ref.abs().max().item() * 2 ** -8:.4f}), NaNs in the whole output {nan}', flush=True)
res = {m: [] for m in masks}
for rnd in range(5):
  for m in masks: res[m].append(timeit(lambda: run(m)))
names = {2: "no gelu' math", 64: 'no aux loads', 256: 'no barriers (wrong)', 128: 'no counted waits (wrong)', 32: 'stamps', 1: 'stores to a 1-MiB window', 4: 'no LDS-DMA', 8: 'no MFMA', 16: 'no staging / stores'}
for m in masks:
  v = sorted(res[m]); lab = ' + '.join(names[k] for k in names if m & k) or 'full kernel'
  print(f'N={N} {" ".join(extra)} mask {m:3d} {lab:50s} median {v[len(v)//2]:7.3f} ms  min {v[0]:7.3f} ms   ({2*M*N*384/v[len(v)//2]/1e9:7.1f} TF/s)', flush=True)
  if m & 32:
    dbg.zero_(); run(m); torch.cuda.synchronize()
    t = dbg.view(256, 4, 4).double().cpu(); tot = t.sum(dim=(0, 1)); lab4 = ['barrier', 'phase body', 'end wait (vmcnt)', 'tile head (A rows)']
    ntile = M / 256 / 256
    print('  stamps (cycles per wave per 256-row tile; MFMA floor %d):' % (N // 64 * 96 * 32))
    for k in range(4): print(f'    {lab4[k]:20s} {tot[k] / 1024 / ntile:10.0f}   {100 * tot[k] / tot.sum():5.1f} %')
    print(f'    sum {tot.sum() / 1024 / ntile:10.0f}')
