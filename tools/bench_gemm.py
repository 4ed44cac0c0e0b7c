"""Per-shape throughput of the tiled bf16 GEMMs on the shapes of the 3DSPA step (random operands)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spa3d

lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')


def timeit(fn, n=10):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n


M_ENC = 8 * 2048 * 151
M_RO = 8 * 512 * 129
shapes = [('enc qkv', M_ENC, 2304, 384), ('enc out', M_ENC, 384, 768), ('enc mlp_in', M_ENC, 1536, 384), ('enc mlp_out', M_ENC, 384, 1536),
          ('enc dqkv->dx', M_ENC, 384, 2304), ('ro qkv', M_RO, 2304, 1280), ('ro out', M_RO, 1280, 768), ('ro mlp_in', M_RO, 1536, 1280),
          ('ro mlp_out', M_RO, 1280, 1536), ('dino', 8 * 2048 * 150, 384, 768), ('square 8k', 8192, 8192, 8192)]
print('NT  (Y = X.W):')
for name, M, N, K in ([] if os.environ.get('TN_ONLY') else shapes):
  A = torch.randn(M, K, device='cuda').bfloat16()
  B = (torch.randn(K, N, device='cuda') / K ** 0.5).bfloat16()
  Cc = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  f = lambda: lib.spa3d_op_linear(A.data_ptr(), B.data_ptr(), None, None, Cc.data_ptr(), M, N, K, 0, 1, 2, ws.data_ptr(), ws.numel(), s())
  assert f() == 0
  ms = timeit(f)
  byts = (M * K + K * N + M * N) * 2
  print(f'  {name:14s} M={M:8d} N={N:5d} K={K:5d}  {ms:8.3f} ms  {2*M*N*K/ms/1e9:8.1f} TF/s  {byts/ms/1e6:7.1f} GB/s')
  del A, B, Cc
print('TN  (dW = X^T.dY):')
for name, M, N, K in ([] if os.environ.get('NT_ONLY') else shapes[:-1]):
  A = torch.randn(M, K, device='cuda').bfloat16()
  dC = torch.randn(M, N, device='cuda').bfloat16()
  dB = torch.empty(K, N, device='cuda')
  Bd = torch.empty(K, N, device='cuda', dtype=torch.bfloat16)
  f = lambda: lib.spa3d_op_linear_bwd(A.data_ptr(), Bd.data_ptr(), dC.data_ptr(), None, dB.data_ptr(), None, M, N, K, 1, 2, ws.data_ptr(), ws.numel(), s())
  assert f() == 0
  ms = timeit(f)
  print(f'  {name:14s} M={M:8d} N={N:5d} K={K:5d}  {ms:8.3f} ms  {2*M*N*K/ms/1e9:8.1f} TF/s  {(M*K+M*N)*2/ms/1e6:7.1f} GB/s')
  del A, dC, dB, Bd
