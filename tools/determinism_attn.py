"""Race screen for the fused attention kernels: o / lse / dq / dk / dv of repeated launches on the same inputs must be bit-identical
(the kernels have no atomics on these outputs; only the two RMSNorm scale gradients are accumulated atomically)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
H, Dh = 8, 96; E = H * Dh
ok = True
for (nseq, S, dt, dcode) in ((4096, 151, torch.bfloat16, 1), (2048, 129, torch.bfloat16, 1), (1024, 301, torch.float16, 2), (2048, 176, torch.bfloat16, 1)):
  qkv = torch.randn(nseq, S, 3 * E, device='cuda').to(dt)
  sq = 1 + 0.1 * torch.randn(Dh, device='cuda'); sk = 1 + 0.1 * torch.randn(Dh, device='cuda')
  km = (torch.rand(nseq, S, device='cuda') < 0.9).float(); km[:, 0] = 1
  kmp = None if os.environ.get('NOMASK') else km.data_ptr()  # NOMASK=1: the no-mask kernels (pruned encoder / readout)
  d_o = torch.randn(nseq, S, E, device='cuda').to(dt)
  ws = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
  outs = []
  for rep in range(4):
    o = torch.zeros(nseq, S, E, device='cuda', dtype=dt); lse = torch.zeros(nseq, H, S, 2, device='cuda')
    dqkv = torch.zeros_like(qkv); dsq = torch.zeros(Dh, device='cuda'); dsk = torch.zeros(Dh, device='cuda')
    assert lib.spa3d_op_attention(qkv[..., :E].data_ptr(), qkv[..., E:2*E].data_ptr(), qkv[..., 2*E:].data_ptr(), 3*E, 3*E, 3*E, sq.data_ptr(), sk.data_ptr(), kmp, nseq, S, S, H, Dh, o.data_ptr(), lse.data_ptr(), dcode, 2, ws.data_ptr(), ws.numel(), s()) == 0
    assert lib.spa3d_op_attention_bwd(qkv[..., :E].data_ptr(), qkv[..., E:2*E].data_ptr(), qkv[..., 2*E:].data_ptr(), 3*E, 3*E, 3*E, sq.data_ptr(), sk.data_ptr(), kmp, nseq, S, S, H, Dh, o.data_ptr(), lse.data_ptr(), d_o.data_ptr(), dqkv[..., :E].data_ptr(), dqkv[..., E:2*E].data_ptr(), dqkv[..., 2*E:].data_ptr(), dsq.data_ptr(), dsk.data_ptr(), dcode, 2, ws.data_ptr(), ws.numel(), s()) == 0
    torch.cuda.synchronize()
    outs.append((o.clone(), lse.clone(), dqkv.clone()))
  same = all(torch.equal(outs[0][i].view(torch.int16 if i != 1 else torch.int32), outs[r][i].view(torch.int16 if i != 1 else torch.int32)) for r in range(1, 4) for i in range(3))
  fin = all(bool(torch.isfinite(t.float()).all()) for t in outs[0])
  print(f'S={S} nseq={nseq} {dt}: repeat-identical {same} finite {fin}', flush=True)
  ok = ok and same and fin
sys.exit(0 if ok else 1)
