"""Build libspa3d_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and on first import."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libspa3d_hip.so')
SOURCES = ['kernels.hip', 'gemm_generic.hip', 'gemm_fast.hip', 'gemm_tnb.hip', 'gemm_ntb.hip', 'gemm_rs.hip', 'mlp_fused.hip', 'attention.hip', 'attention_fused.hip', 'qkv_attn.hip', 'ops.hip', 'samplers.hip', 'model.hip']
F16_SOURCES = ['kernels.hip', 'gemm_generic.hip', 'gemm_fast.hip', 'gemm_tnb.hip', 'gemm_ntb.hip', 'gemm_rs.hip', 'mlp_fused.hip', 'attention.hip', 'attention_fused.hip', 'qkv_attn.hip', 'ops.hip', 'model.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall', '-Wno-unused-function', '-Wno-inline-asm',
         '-Wno-unused-variable', '-Wno-unused-but-set-variable']


def _hipcc():
  for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
    if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
      return c
  raise RuntimeError('hipcc not found')


def _stale(out, deps):
  if not os.path.exists(out):
    return True
  t = os.path.getmtime(out)
  return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
  hipcc = _hipcc()
  hdrs = [os.path.join(CSRC, 'common.hpp'), os.path.join(CSRC, 'tn_args.hpp'), os.path.join(CSRC, 'ablate.inc'), os.path.join(CSRC, 'attn_common.hpp'), os.path.join(HERE, '..', 'include', 'spa3d.h')]
  objdir = os.path.join(HERE, 'build')
  os.makedirs(objdir, exist_ok=True)
  srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
  # every source with 16-bit device code is compiled twice: bf16 (default) and, with -DSPA_F16=1, IEEE fp16 (common.hpp)
  units = [(s, s.replace('.hip', '.o'), []) for s in srcs] + [(s, s.replace('.hip', '_f16.o'), ['-DSPA_F16=1']) for s in srcs if s in F16_SOURCES]
  jobs = []
  for s, o, extra in units:
    src = os.path.join(CSRC, s)
    obj = os.path.join(objdir, o)
    if force or _stale(obj, [src] + hdrs):
      jobs.append((src, obj, extra))

  def cc(job):
    src, obj, extra = job
    cmd = [hipcc] + FLAGS + extra + ['-USPA3D_ABLATION_BUILD', '-c', src, '-o', obj]
    # the product library never carries a work-skipping diagnostic switch (csrc/ablate.inc)
    assert not any('ABL' in f for f in FLAGS + extra), 'ablation flags are for tools/ablate_*.py only'
    if verbose:
      print(' '.join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError(f'hipcc failed on {src}:\n{r.stdout}\n{r.stderr}')
    if verbose and r.stderr.strip():
      print(r.stderr, file=sys.stderr)
    return obj

  with ThreadPoolExecutor(max_workers=6) as ex:
    list(ex.map(cc, jobs))
  objs = [os.path.join(objdir, o) for _, o, _ in units]
  if force or jobs or _stale(LIB, objs):
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
      print(' '.join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
  return LIB


if __name__ == '__main__':
  print(build(force='--force' in sys.argv))
