"""Diagnostic only: time tools/bench_attn.py against SEPARATE libraries whose attention_fused.hip is compiled with extra -D flags
(structure experiments; never the product library).
    python tools/variants_attn.py "rot0:-DSPA3D_FWD_ROT=0;rot1:-DSPA3D_FWD_ROT=1"      (S, NSEQ, NOMASK from the environment)"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
b = importlib.import_module('3dspa_code_amd.build')
out = os.path.join(ROOT, 'tools', '_ablate'); os.makedirs(out, exist_ok=True)
b.build(verbose=False)
objs = [os.path.join(b.HERE, 'build', o) for o in sorted(os.listdir(os.path.join(b.HERE, 'build'))) if o.endswith('.o') and o != 'attention_fused.o']
variants = [v.split(':') for v in sys.argv[1].split(';')]
def mk(v):
  nm, fl = v
  o_ = os.path.join(out, f'attention_fused_var_{nm}.o'); l_ = os.path.join(out, f'libspa3d_var_{nm}.so')
  subprocess.check_call([b._hipcc()] + b.FLAGS + [x for x in fl.split(',') if x] + ['-c', os.path.join(b.CSRC, 'attention_fused.hip'), '-o', o_])
  subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', l_] + objs + [o_])
  return l_
with ThreadPoolExecutor(max_workers=8) as ex:
  libs = list(ex.map(mk, variants))
for rep in range(int(os.environ.get('REPS', 2))):
  for (nm, fl), l_ in zip(variants, libs):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bench_attn.py')], env=dict(os.environ, SPA3D_TOOL_LIB=l_), capture_output=True, text=True)
    print(f'{nm:10s} ' + (r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else r.stderr[-400:]), flush=True)
