"""Fused QKV projection + attention forward (csrc/qkv_attn.hip) against the two launches it replaces (row-stationary K = 384 GEMM + fused attention kernel), at the
track encoder's shape: python tools/bench_qkv_attn.py [nseq] [S]"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 20300
S = int(sys.argv[2]) if len(sys.argv) > 2 else 151
H, E = 8, 768
rows = nseq * S
nq = torch.randn(rows, 384, device='cuda').bfloat16()
w = [(torch.randn(384, E, device='cuda') / math.sqrt(384)).bfloat16() for _ in range(3)]
wcat = torch.cat(w, dim=1).contiguous()
sq = torch.ones(96, device='cuda'); sk = torch.ones(96, device='cuda')
qkv = torch.empty(rows, 3 * E, device='cuda', dtype=torch.bfloat16); o = torch.empty(rows, E, device='cuda', dtype=torch.bfloat16)
lse = torch.empty(rows * H * 2, device='cuda')
ws = torch.empty(256 << 20, dtype=torch.uint8, device='cuda')
def fused():
  assert lib.spa3d_op_qkv_attention(nq.data_ptr(), 384, w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(), sq.data_ptr(), sk.data_ptr(), None, None, nseq, S, H,
                                    qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), 1, ws.data_ptr(), ws.numel(), s()) == 0
def gemm():
  assert lib.spa3d_op_linear(nq.data_ptr(), wcat.data_ptr(), None, None, qkv.data_ptr(), rows, 3 * E, 384, 0, 1, 0, ws.data_ptr(), ws.numel(), s()) == 0
def attn():
  assert lib.spa3d_op_attention(qkv.data_ptr(), qkv[:, E:].data_ptr(), qkv[:, 2 * E:].data_ptr(), 3 * E, 3 * E, 3 * E, sq.data_ptr(), sk.data_ptr(), None, nseq, S, S, H, 96,
                                o.data_ptr(), lse.data_ptr(), 1, 2, ws.data_ptr(), ws.numel(), s()) == 0
def t(f, n=5):
  f(); torch.cuda.synchronize(); ts = []
  for _ in range(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
  return sorted(ts)[n // 2]
tg, ta, tf = t(gemm), t(attn), t(fused)
fl = 2.0 * rows * 384 * 2304 + 4.0 * rows * H * S * 96
print(f'nseq {nseq} S {S}: projection GEMM (incl. its weight pack) {tg:.3f} ms + attention {ta:.3f} ms = {tg + ta:.3f} ms;  fused (incl. pack) {tf:.3f} ms  = x{tf / (tg + ta):.3f};  '
      f'fused {fl / tf / 1e9:.0f} TF/s, {tf * 1e3 / (nseq * H / 256):.2f} us per CU-problem', flush=True)
