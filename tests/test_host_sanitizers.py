"""SURVEY.md 5 "ASAN on the host shim": the host orchestration of libspa3d_hip.so (csrc/model.hip, csrc/ops.hip: leaf tree, bump arena, the dry run
that sizes the workspace by executing the whole forward / backward orchestration with launches disabled) compiled with
-fsanitize=address,undefined (host only: -fno-gpu-sanitize; GPU sanitizers are not available on this pool) and driven by a plain C++ host
(tests/host/spa3d_host_dryrun.cpp) through the C-ABI for the five BASELINE.json shapes, the 2-D twin and the refused configurations.
No GPU call is made; runs on the CPU box."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _build():
  b = importlib.import_module('3dspa_code_amd.build')
  b.build(verbose=False)
  out = os.path.join(b.HERE, 'build', 'asan')
  os.makedirs(out, exist_ok=True)
  san = ['-fsanitize=address,undefined', '-fno-gpu-sanitize', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer', '-g']
  flags = [f for f in b.FLAGS if f != '-O3'] + ['-O1'] + san
  hdrs = [os.path.join(b.CSRC, 'common.hpp'), os.path.join(ROOT, 'include', 'spa3d.h')]
  jobs = []
  for src in ('model.hip', 'ops.hip'):
    for suffix, extra in (('', []), ('_f16', ['-DSPA_F16=1'])):
      obj = os.path.join(out, src.replace('.hip', suffix + '.o'))
      deps = [os.path.join(b.CSRC, src)] + hdrs
      if not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps):
        jobs.append(subprocess.Popen([b._hipcc()] + flags + extra + ['-c', os.path.join(b.CSRC, src), '-o', obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
  for j in jobs:
    o, _ = j.communicate()
    assert j.returncode == 0, o[-3000:]
  regular = [os.path.join(b.HERE, 'build', o) for o in sorted(os.listdir(os.path.join(b.HERE, 'build')))
             if o.endswith('.o') and not o.startswith(('model', 'ops'))]
  exe = os.path.join(out, 'spa3d_host_dryrun')
  drv = os.path.join(ROOT, 'tests', 'host', 'spa3d_host_dryrun.cpp')
  cmd = [b._hipcc(), '--offload-arch=gfx950', '-x', 'hip'] + san + ['-O1', '-I', os.path.join(ROOT, 'include'), drv, '-x', 'none'] + \
        [os.path.join(out, o) for o in ('model.o', 'model_f16.o', 'ops.o', 'ops_f16.o')] + regular + ['-o', exe]
  r = subprocess.run(cmd, capture_output=True, text=True)
  assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
  return exe


def test_host_orchestration_dry_runs_under_asan_and_ubsan(tmp_path):
  exe = _build()
  supp = tmp_path / 'lsan.supp'
  supp.write_text('leak:libamdhip64\nleak:libhsa-runtime64\nleak:libamd_comgr\nleak:librocprofiler\n')
  env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0:halt_on_error=1', UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1',
             LSAN_OPTIONS=f'suppressions={supp}:print_suppressions=0')
  for k in ('SPA3D_GEMM_IMPL', 'SPA3D_ATTN_IMPL', 'SPA3D_PRUNE', 'SPA3D_RO_SHARE', 'SPA3D_CHUNK', 'SPA3D_LOSS_SCALE'):
    env.pop(k, None)
  r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
  print(r.stdout[-3000:], r.stderr[-4000:])
  assert r.returncode == 0 and 'HOST_DRYRUN_OK' in r.stdout
  assert 'AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr and 'LeakSanitizer' not in r.stderr
