"""Fused attention forward/backward at the track-encoder shape (16384 sequences x 8 heads, S=151)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
if os.environ.get('SPA3D_TOOL_LIB'): spa3d._lib.LIB_PATH = os.environ['SPA3D_TOOL_LIB']  # tools/variants_attn.py: a diagnostic build
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
nseq, S, H, Dh = int(os.environ.get('NSEQ', 16384)), int(os.environ.get('S', 151)), 8, 96
E = H * Dh
qkv = torch.randn(nseq, S, 3 * E, device='cuda').bfloat16()
sq = torch.ones(Dh, device='cuda'); sk = torch.ones(Dh, device='cuda')
km = (torch.rand(nseq, S, device='cuda') < 0.9).float(); km[:, 0] = 1
kmp = None if os.environ.get('NOMASK') else km.data_ptr()  # NOMASK=1: no key mask (the pruned track encoder and the readout stack run without one)
o = torch.empty(nseq, S, E, device='cuda', dtype=torch.bfloat16); lse = torch.empty(nseq, H, S, 2, device='cuda')
d_o = torch.randn(nseq, S, E, device='cuda').bfloat16(); dqkv = torch.empty_like(qkv)
dsq = torch.zeros(Dh, device='cuda'); dsk = torch.zeros(Dh, device='cuda')
ws = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
fwd = lambda: lib.spa3d_op_attention(qkv[..., :E].data_ptr(), qkv[..., E:2*E].data_ptr(), qkv[..., 2*E:].data_ptr(), 3*E, 3*E, 3*E, sq.data_ptr(), sk.data_ptr(), kmp, nseq, S, S, H, Dh, o.data_ptr(), lse.data_ptr(), 1, 2, ws.data_ptr(), ws.numel(), s())
bwd = lambda: lib.spa3d_op_attention_bwd(qkv[..., :E].data_ptr(), qkv[..., E:2*E].data_ptr(), qkv[..., 2*E:].data_ptr(), 3*E, 3*E, 3*E, sq.data_ptr(), sk.data_ptr(), kmp, nseq, S, S, H, Dh, o.data_ptr(), lse.data_ptr(), d_o.data_ptr(), dqkv[..., :E].data_ptr(), dqkv[..., E:2*E].data_ptr(), dqkv[..., 2*E:].data_ptr(), dsq.data_ptr(), dsk.data_ptr(), 1, 2, ws.data_ptr(), ws.numel(), s())
def timeit(fn, n=5):
  assert fn() == 0; torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
tf, tb = timeit(fwd), timeit(bwd)
rb_f = nseq * H * S * Dh * 2 * 4; rb_b = nseq * H * S * Dh * 2 * 8
print(f'S={S} nseq={nseq} {"nomask" if kmp is None else "mask"} BWD_MODE={os.environ.get("SPA3D_ATTN_BWD_MODE","0")} fwd {tf:.3f} ms ({rb_f/tf/1e6:.0f} GB/s)  bwd {tb:.3f} ms ({rb_b/tb/1e6:.0f} GB/s)')
