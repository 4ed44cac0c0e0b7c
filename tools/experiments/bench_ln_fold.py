"""Round-3 measurement for review item 6 (LayerNorm folded into the 128 x 384-tile GEMM epilogues): what the fold would ADD to the GEMM is a second
768-byte-per-row output stream (LN(a) next to a) -- measured here with the kernel's existing dual-store path (pre-activation copy) on the out-projection
(K = 768) and MLP-out (K = 1536) shapes with their residual operand -- against what it would REMOVE: the stand-alone LayerNorm forward on the same rows."""
import ctypes as C, sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if not os.environ.get('CHILD'):
  for pre in ('', '1'):
    env = dict(os.environ, CHILD='1')
    if pre: env['SPA3D_OP_PREOUT'] = '1'
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
    print(f'== second output stream: {"on" if pre else "off"}\n' + '\n'.join(l for l in (r.stdout + r.stderr[-600:]).splitlines() if 'amdgpu.ids' not in l), flush=True)
  sys.exit(0)
import torch, spa3d
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=10):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
M, N = 3063168, 384
ws = torch.empty(M * N * 2 + (64 << 20), dtype=torch.uint8, device='cuda')
res = torch.randn(M, N, device='cuda').bfloat16(); bias = torch.randn(N, device='cuda')
Cc = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
for K in (768, 1536):
  A = torch.randn(M, K, device='cuda').bfloat16(); B = (torch.randn(K, N, device='cuda') / K ** 0.5).bfloat16()
  f = lambda: lib.spa3d_op_linear(A.data_ptr(), B.data_ptr(), bias.data_ptr(), res.data_ptr(), Cc.data_ptr(), M, N, K, 0, 1, 2, ws.data_ptr(), ws.numel(), s())
  assert f() == 0
  ms = timeit(f)
  print(f'GEMM + bias + residual M={M} N={N} K={K}: {ms:.3f} ms ({2*M*N*K/ms/1e9:.0f} TF/s)', flush=True)
  del A, B
if not os.environ.get('SPA3D_OP_PREOUT'):
  x = torch.randn(M, N, device='cuda').bfloat16(); y = torch.empty_like(x); sc = torch.ones(N, device='cuda'); st = torch.empty(M, 2, device='cuda')
  g = lambda: lib.spa3d_op_layernorm(x.data_ptr(), sc.data_ptr(), y.data_ptr(), st.data_ptr(), M, N, 1, s())
  assert g() == 0
  print(f'stand-alone LayerNorm forward rows={M} d={N}: {timeit(g):.3f} ms', flush=True)
