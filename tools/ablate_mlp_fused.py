"""Diagnostic only: per-phase ablation of the fused MLP forward (csrc/mlp_fused.hip).  One SEPARATE library per compile-time mask
(-DSPA3D_MF_ABLATE=mask; a run-time mask changes the register allocation of what remains), never the product:
1 h / hpre stores into a 1-MiB window (no HBM write stream), 2 no gelu arithmetic, 4 no LDS-DMA, 8 no MFMAs, 16 no y stores,
32 s_memtime stamps at the phase seams (printed as shares of the summed wave time).  Interleaved rounds in one process."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
b = importlib.import_module('3dspa_code_amd.build')
out = os.path.join(ROOT, 'tools', '_ablate'); os.makedirs(out, exist_ok=True)
b.build(verbose=False)
objs = [os.path.join(b.HERE, 'build', o) for o in sorted(os.listdir(os.path.join(b.HERE, 'build'))) if o.endswith('.o') and o != 'mlp_fused.o']
masks = [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else '0,1,17,2,3,19,4,8,12,31'.split(','))]
extra = [x for x in os.environ.get('VARIANT', '').split(',') if x]  # structure variants: VARIANT=-DSPA3D_MF_YSCHED=1
import torch
libs = {}
for m in masks:
  ao = os.path.join(out, f'mlp_fused_abl{m}.o')
  subprocess.check_call([b._hipcc()] + b.FLAGS + ['-DSPA3D_ABLATION_BUILD', f'-DSPA3D_ABL_MF={m}'] + extra + ['-c', os.path.join(b.CSRC, 'mlp_fused.hip'), '-o', ao])
  lp = os.path.join(out, f'libspa3d_mf_abl{m}.so')
  subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lp] + objs + [ao])
  libs[m] = C.CDLL(lp)
  libs[m].spa3d_op_mlp_fused.argtypes = [C.c_void_p] * 9 + [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 26, dtype=torch.uint8, device='cuda')
M = int(os.environ.get('M', (9 * 2048 * 151 * 9 // 10) // 128 * 128)); d, mlp = 384, 1536
na = torch.randn(M, d, device='cuda').bfloat16(); a = torch.randn(M, d, device='cuda').bfloat16()
w_in = (torch.randn(d, mlp, device='cuda') / d ** 0.5).bfloat16(); w_out = (torch.randn(mlp, d, device='cuda') / mlp ** 0.5).bfloat16()
b_in = torch.randn(mlp, device='cuda'); b_out = torch.randn(d, device='cuda')
y = torch.empty(M, d, device='cuda', dtype=torch.bfloat16); h = torch.empty(M, mlp, device='cuda', dtype=torch.bfloat16); hp = torch.empty_like(h)
def run(m):
  assert libs[m].spa3d_op_mlp_fused(na.data_ptr(), a.data_ptr(), w_in.data_ptr(), b_in.data_ptr(), w_out.data_ptr(), b_out.data_ptr(), y.data_ptr(),
                                    h.data_ptr(), hp.data_ptr(), M, d, mlp, 1, ws.data_ptr(), ws.numel(), s()) == 0
def timeit(fn, n=5):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
dbg = torch.zeros(256 * 4 * 8, dtype=torch.int64, device='cuda')
os.environ['SPA3D_MF_DBG'] = str(dbg.data_ptr())
for m in masks: run(m)
torch.cuda.synchronize()
res = {m: [] for m in masks}
for rnd in range(5):
  for m in masks: res[m].append(timeit(lambda: run(m)))
names = {256: 'no phase barriers (wrong results)', 128: 'no counted waits (wrong results)', 64: 'plain (not nt) stores', 32: 'stamps', 1: 'h/hpre stores to a 1-MiB window', 2: 'no gelu math', 4: 'no LDS-DMA', 8: 'no MFMA', 16: 'no y stores'}
for m in masks:
  v = sorted(res[m]); lab = ' + '.join(names[k] for k in names if m & k) or 'full kernel'
  print(f'{" ".join(extra)} mask {m:2d} {lab:60s} median {v[len(v)//2]:7.3f} ms  min {v[0]:7.3f} ms   ({2*2*M*d*mlp/v[len(v)//2]/1e9:7.1f} TF/s-equivalent)', flush=True)
if any(m & 32 for m in masks):
  t = dbg.view(256, 4, 8).double().cpu()
  tot = t.sum(dim=(0, 1)); lab = ['barrier', 'X body', 'Y body', 'end wait (vmcnt)', 'tile head', 'exposed gelu', 'epilogue', 'h/hpre stores']
  print('stamps (cycles per wave per tile, mean over waves; share of their sum):')
  ntile = M / 128 / 256
  for k in range(8): print(f'  {lab[k]:18s} {tot[k] / 1024 / ntile:10.0f}   {100 * tot[k] / tot.sum():5.1f} %')
  print(f'  sum {tot.sum() / 1024 / ntile:10.0f} cycles per tile = {tot.sum() / 1024 / ntile / 2.1e3:.1f} us at 2.1 GHz (s_memtime ticks at 100 MHz x ? -- compare shares)')
