"""One kernel class launched back to back for ~25 s (tools/power_probe.py samples rocm-smi beside it)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(1 << 28, dtype=torch.uint8, device='cuda')
what = sys.argv[1]
if what == 'tnb':
  M, N, K = 726528, 2304, 1280
  A = torch.randn(M, K, device='cuda').bfloat16(); dC = torch.randn(M, N, device='cuda').bfloat16(); dB = torch.zeros(K, N, device='cuda'); W = torch.empty(K, N, device='cuda', dtype=torch.bfloat16)
  f = lambda: lib.spa3d_op_linear_bwd(A.data_ptr(), W.data_ptr(), dC.data_ptr(), None, dB.data_ptr(), None, M, N, K, 1, 0, ws.data_ptr(), ws.numel(), s())
elif what == 'ntb':
  M, N, K = 726528, 1280, 2304
  A = torch.randn(M, K, device='cuda').bfloat16(); W = (torch.randn(K, N, device='cuda') / K ** 0.5).bfloat16(); Cd = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  f = lambda: lib.spa3d_op_linear(A.data_ptr(), W.data_ptr(), None, None, Cd.data_ptr(), M, N, K, 0, 1, 10, ws.data_ptr(), ws.numel(), s())
elif what == 'nt8':
  M, N, K = 726528, 1280, 2304
  A = torch.randn(M, K, device='cuda').bfloat16(); W = (torch.randn(K, N, device='cuda') / K ** 0.5).bfloat16(); Cd = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  f = lambda: lib.spa3d_op_linear(A.data_ptr(), W.data_ptr(), None, None, Cd.data_ptr(), M, N, K, 0, 1, 0, ws.data_ptr(), ws.numel(), s())
elif what == 'attn':
  nseq, S, H, Dh = 20298, 151, 8, 96; E = H * Dh
  qkv = torch.randn(nseq, S, 3 * E, device='cuda').bfloat16(); sq = torch.ones(Dh, device='cuda'); sk = torch.ones(Dh, device='cuda')
  o = torch.empty(nseq, S, E, device='cuda', dtype=torch.bfloat16); lse = torch.empty(nseq, H, S, 2, device='cuda'); d_o = torch.randn(nseq, S, E, device='cuda').bfloat16(); dqkv = torch.empty_like(qkv)
  dsq = torch.zeros(Dh, device='cuda'); dsk = torch.zeros(Dh, device='cuda')
  assert lib.spa3d_op_attention(qkv[..., :E].data_ptr(), qkv[..., E:2*E].data_ptr(), qkv[..., 2*E:].data_ptr(), 3*E, 3*E, 3*E, sq.data_ptr(), sk.data_ptr(), None, nseq, S, S, H, Dh, o.data_ptr(), lse.data_ptr(), 1, 2, ws.data_ptr(), ws.numel(), s()) == 0
  f = lambda: lib.spa3d_op_attention_bwd(qkv[..., :E].data_ptr(), qkv[..., E:2*E].data_ptr(), qkv[..., 2*E:].data_ptr(), 3*E, 3*E, 3*E, sq.data_ptr(), sk.data_ptr(), None, nseq, S, S, H, Dh, o.data_ptr(), lse.data_ptr(), d_o.data_ptr(), dqkv[..., :E].data_ptr(), dqkv[..., E:2*E].data_ptr(), dqkv[..., 2*E:].data_ptr(), dsq.data_ptr(), dsk.data_ptr(), 1, 2, ws.data_ptr(), ws.numel(), s())
else:
  M, D = 3401728, 384
  x = torch.randn(M, D, device='cuda').bfloat16(); sc = torch.ones(D, device='cuda'); y = torch.empty_like(x); st = torch.empty(M, 2, device='cuda'); dy = torch.randn(M, D, device='cuda').bfloat16(); dx = torch.empty_like(x); dsc = torch.zeros(D, device='cuda')
  assert lib.spa3d_op_layernorm(x.data_ptr(), sc.data_ptr(), y.data_ptr(), st.data_ptr(), M, D, 1, s()) == 0
  f = lambda: lib.spa3d_op_layernorm_bwd(x.data_ptr(), sc.data_ptr(), st.data_ptr(), dy.data_ptr(), dx.data_ptr(), dsc.data_ptr(), M, D, 1, s())
assert f() == 0; torch.cuda.synchronize()
t0 = time.time()
while time.time() - t0 < 25:
  for _ in range(50): f()
  torch.cuda.synchronize()
