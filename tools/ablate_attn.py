"""Diagnostic only: builds a SEPARATE library (tools/_ablate/libspa3d_ablate.so) whose attention_fused.hip is compiled with
-DSPA3D_ABLATE (work-skipping masks; never defined for the product libspa3d_hip.so) and times the fused attention backward with parts
removed, to see where a problem's time goes.  Outputs of an ablated run are wrong by construction; read SHARES only.
    python tools/ablate_attn.py            (S, NSEQ from the environment as in tools/bench_attn.py)"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
b = importlib.import_module('3dspa_code_amd.build')
out = os.path.join(ROOT, 'tools', '_ablate'); os.makedirs(out, exist_ok=True)
MODE4 = False  # (the single-orientation kernel these masks ablated moved to tools/experiments/ in round 4)
lib_path = os.path.join(out, 'libspa3d_ablate.so')
if not os.environ.get('ABL1_CHILD'):
  b.build(verbose=False)
  objs = [os.path.join(b.HERE, 'build', o) for o in sorted(os.listdir(os.path.join(b.HERE, 'build'))) if o.endswith('.o') and o != 'attention_fused.o']
  if not MODE4:
    ao = os.path.join(out, 'attention_fused_ablate.o')
    subprocess.check_call([b._hipcc()] + b.FLAGS + ['-DSPA3D_ABLATION_BUILD', '-DSPA3D_ABL_ATTN=1', '-c', os.path.join(b.CSRC, 'attention_fused.hip'), '-o', ao])
    subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib_path] + objs + [ao])
if MODE4 and not os.environ.get('ABL1_CHILD'):
  from concurrent.futures import ThreadPoolExecutor
  masks = [(0, 'full'), (1, 'staging only'), (2, 'no staging'), (16, 'first round of tiles only (tiles 0-7)'), (32, 'no phase 2'), (128, 'no dk/dv stores'),
           (256, 'no scale-gradient flushes'), (2 + 32 + 128 + 256, 'phase 1 alone (no staging, stores, flushes, phase 2)'),
           (2 + 16 + 32 + 128 + 256, 'phase 1 first round alone'), (1 + 2, 'barriers + small loads only')]
  if os.environ.get('VARIANTS'):  # structure experiments: "name:-DA=1,-DB=2;name2:..." each timed in full (mask 0)
    masks = []
    for v in os.environ['VARIANTS'].split(';'):
      nm, fl = v.split(':')
      masks.append((nm, fl.split(',') if fl else []))
  def mk(m):
    extra = [f'-DSPA3D_ABL1={m}']
    if isinstance(m, str): extra = [x for x in dict(masks)[m] if x]
    o_ = os.path.join(out, f'attention_fused_abl1_{m}.o'); l_ = os.path.join(out, f'libspa3d_abl1_{m}.so')
    subprocess.check_call([b._hipcc()] + b.FLAGS + extra + ['-c', os.path.join(b.CSRC, 'attention_fused.hip'), '-o', o_])
    subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', l_] + objs + [o_])
    return l_
  with ThreadPoolExecutor(max_workers=8) as ex:
    libs = list(ex.map(mk, [m for m, _ in masks]))
  for (m, what), l_ in zip(masks, libs):
    env = dict(os.environ, ABL1_CHILD=str(m), ABL1_LIB=l_)
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
    print((r.stdout.strip() or r.stderr[-400:]) + f'  [{what}]', flush=True)
  sys.exit(0)
import torch, spa3d
spa3d._lib.LIB_PATH = os.environ.get('ABL1_LIB', lib_path)
lib = spa3d._lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
nseq, S, H, Dh = int(os.environ.get('NSEQ', 16384)), int(os.environ.get('S', 151)), 8, 96
E = H * Dh
qkv = torch.randn(nseq, S, 3 * E, device='cuda').bfloat16()
sq = torch.ones(Dh, device='cuda'); sk = torch.ones(Dh, device='cuda')
km = (torch.rand(nseq, S, device='cuda') < 0.9).float(); km[:, 0] = 1
o = torch.empty(nseq, S, E, device='cuda', dtype=torch.bfloat16); lse = torch.empty(nseq, H, S, 2, device='cuda')
d_o = torch.randn(nseq, S, E, device='cuda').bfloat16(); dqkv = torch.empty_like(qkv)
dsq = torch.zeros(Dh, device='cuda'); dsk = torch.zeros(Dh, device='cuda')
ws = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
fwd = lambda: lib.spa3d_op_attention(qkv[..., :E].data_ptr(), qkv[..., E:2*E].data_ptr(), qkv[..., 2*E:].data_ptr(), 3*E, 3*E, 3*E, sq.data_ptr(), sk.data_ptr(), km.data_ptr(), nseq, S, S, H, Dh, o.data_ptr(), lse.data_ptr(), 1, 2, ws.data_ptr(), ws.numel(), s())
bwd = lambda: lib.spa3d_op_attention_bwd(qkv[..., :E].data_ptr(), qkv[..., E:2*E].data_ptr(), qkv[..., 2*E:].data_ptr(), 3*E, 3*E, 3*E, sq.data_ptr(), sk.data_ptr(), km.data_ptr(), nseq, S, S, H, Dh, o.data_ptr(), lse.data_ptr(), d_o.data_ptr(), dqkv[..., :E].data_ptr(), dqkv[..., E:2*E].data_ptr(), dqkv[..., 2*E:].data_ptr(), dsq.data_ptr(), dsk.data_ptr(), 1, 2, ws.data_ptr(), ws.numel(), s())
def timeit(fn, n=5):
  assert fn() == 0; torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n
assert fwd() == 0
if MODE4:  # single-orientation kernel (round 3): ONE OBJECT PER MASK (-DSPA3D_ABL1=mask, a compile-time
  # constant), so the code that remains is compiled exactly as in the product; each variant runs in a child process
  if os.environ.get('ABL1_CHILD'):
    kmp = None if os.environ.get('NOMASK') else km.data_ptr()
    bwd = lambda: lib.spa3d_op_attention_bwd(qkv[..., :E].data_ptr(), qkv[..., E:2*E].data_ptr(), qkv[..., 2*E:].data_ptr(), 3*E, 3*E, 3*E, sq.data_ptr(), sk.data_ptr(), kmp, nseq, S, S, H, Dh, o.data_ptr(), lse.data_ptr(), d_o.data_ptr(), dqkv[..., :E].data_ptr(), dqkv[..., E:2*E].data_ptr(), dqkv[..., 2*E:].data_ptr(), dsq.data_ptr(), dsk.data_ptr(), 1, 2, ws.data_ptr(), ws.numel(), s())
    assert lib.spa3d_op_attention(qkv[..., :E].data_ptr(), qkv[..., E:2*E].data_ptr(), qkv[..., 2*E:].data_ptr(), 3*E, 3*E, 3*E, sq.data_ptr(), sk.data_ptr(), kmp, nseq, S, S, H, Dh, o.data_ptr(), lse.data_ptr(), 1, 2, ws.data_ptr(), ws.numel(), s()) == 0
    print(f'S={S} nseq={nseq} mode {os.environ["SPA3D_ATTN_BWD_MODE"]} {"nomask" if kmp is None else "mask"} ablate={os.environ["ABL1_CHILD"]}: bwd {timeit(bwd):.3f} ms', flush=True)
  sys.exit(0)
for mask, what in ((0, 'full'), (1, 'staging only (no tile work)'), (2, 'tile work only (no staging)'), (4, 'no dq/dk/dv stores'), (6, 'tile work, no staging, no stores'),
                   (8, 'dQ role without its own S / dP (dS read from LDS): upper bound of a shared-dS structure'), (14, 'the same, tile work only')):
  os.environ['SPA3D_ABLATE'] = str(mask)
  print(f'S={S} nseq={nseq} ablate={mask} [{what}]: bwd {timeit(bwd):.3f} ms', flush=True)
