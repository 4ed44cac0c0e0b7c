"""Diagnostic only: the fused QKV projection + attention forward (csrc/qkv_attn.hip) with parts removed.  SEPARATE libraries under tools/_ablate/ built with
-DSPA3D_ABLATION_BUILD -DSPA3D_ABL_QKVA=mask (csrc/ablate.inc; results of a masked build are WRONG: timing only) and structure variants (name:flag[,flag]).
    MASKS=1,2,4 python tools/ablate_qkv_attn.py [variant ...]"""
import ctypes as C, math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
b = importlib.import_module('3dspa_code_amd.build')
out = os.path.join(ROOT, 'tools', '_ablate'); os.makedirs(out, exist_ok=True)
b.build(verbose=False)
objs = [os.path.join(b.HERE, 'build', o) for o in sorted(os.listdir(os.path.join(b.HERE, 'build'))) if o.endswith('.o') and o != 'qkv_attn.o']
import torch
masks = [int(x) for x in os.environ.get('MASKS', '1,2,3,4,8,12,16,17,31').split(',') if x]
names = {1: 'no attention phase', 2: 'no q|k|v stores', 4: 'no MFMA / fragment reads in the k-loop', 8: 'no LDS-DMA', 16: 'no epilogue'}
builds = [('product', [])] + [(f'mask {m:2d}: ' + ' + '.join(names[k] for k in (1, 2, 4, 8, 16) if m & k), ['-DSPA3D_ABLATION_BUILD', f'-DSPA3D_ABL_QKVA={m}']) for m in masks]
builds += [(v.split(':')[0], v.split(':')[1].split(',')) for v in sys.argv[1:]]
nseq, S, H, E = int(os.environ.get('NSEQ', 20300)), int(os.environ.get('S', 151)), 8, 768
rows = nseq * S
nq = torch.randn(rows, 384, device='cuda').bfloat16()
w = [(torch.randn(384, E, device='cuda') / math.sqrt(384)).bfloat16() for _ in range(3)]
sq = torch.ones(96, device='cuda'); sk = torch.ones(96, device='cuda')
qkv = torch.empty(rows, 3 * E, device='cuda', dtype=torch.bfloat16); o = torch.empty(rows, E, device='cuda', dtype=torch.bfloat16)
lse = torch.empty(rows * H * 2, device='cuda')
ws = torch.empty(256 << 20, dtype=torch.uint8, device='cuda')
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for i, (name, flags) in enumerate(builds):
  ao = os.path.join(out, f'qkv_attn_v{i}.o')
  subprocess.check_call([b._hipcc()] + b.FLAGS + flags + ['-c', os.path.join(b.CSRC, 'qkv_attn.hip'), '-o', ao])
  lp = os.path.join(out, f'libspa3d_qa_v{i}.so')
  subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lp] + objs + [ao])
  L = C.CDLL(lp)
  L.spa3d_op_qkv_attention.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 7 + [C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
  f = lambda: L.spa3d_op_qkv_attention(nq.data_ptr(), 384, w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(), sq.data_ptr(), sk.data_ptr(), None, None, nseq, S, H,
                                       qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), 1, ws.data_ptr(), ws.numel(), s())
  assert f() == 0; torch.cuda.synchronize(); ts = []
  for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
  ts.sort()
  print(f'fused QKV + attention, nseq {nseq} S {S}  {name:70s} median {ts[2]:7.3f} ms  ({ts[2] * 1e3 / (nseq * H / 256):6.2f} us per CU-problem)', flush=True)
