"""bench.py's own N-rank launcher (`python bench.py --gpus N`, as the driver invokes it) on a CPU-only host: --launch-check makes every
rank join a gloo group and exit before any GPU call.  BASELINE.json configs[3] (no reference counterpart: SURVEY 2, no multi-device code)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
  env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
  env.update(env_extra or {})
  return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_plain_invocation_spawns_n_ranks():
  r = _run(['--gpus', '2', '--launch-check'])
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, r.stdout  # rank 0 only
  j = json.loads(lines[0])
  assert j['launch_check'] is True and j['n_gpus'] == 2 and j['ranks_verified'] == 2 and j['requested_gpus'] == 2


def test_single_rank_launch_check():
  r = _run(['--gpus', '1', '--launch-check'])
  assert r.returncode == 0, r.stderr[-2000:]
  assert json.loads(r.stdout.strip().splitlines()[-1])['n_gpus'] == 1


def test_world_size_mismatch_is_refused():
  # a torchrun-style environment of the wrong size must not produce a number
  r = _run(['--gpus', '4', '--launch-check'], env_extra={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
  assert r.returncode == 2 and 'refusing' in r.stderr
  assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


def test_config5_argument_path_and_whole_step_hbm_fields():
  """BASELINE.json configs[4] is reachable from the driver's command (`bench.py --config 5 [--batch B]`), and the whole-step HBM figure comes
  from the committed PMC passes with their round tag (a warning field when they are older than the round this tree is built in)."""
  sys.path.insert(0, ROOT)
  import bench
  c5 = bench.CONFIGS[5]
  assert (c5['N'], c5['Q'], c5['T'], c5['dino'], c5['depth'], c5['precision']) == (8192, 2048, 300, 768, 1, 'fp16') and c5['B'] >= 1
  r = _run(['--gpus', '1', '--config', '5', '--batch', '2', '--launch-check'])
  assert r.returncode == 0, r.stderr[-2000:]
  r = _run(['--gpus', '1', '--config', '4', '--launch-check'])
  assert r.returncode != 0  # configs[3] is --gpus 8 of config 3, not a config of its own
  rows, src, tag, warn = bench.pmc_tables()
  assert rows and tag and src.startswith('profiles/' + tag)
  by = bench.pmc_step_bytes(rows)
  assert 1e12 < by < 2e13  # TB per step: the activations of every op round-trip through HBM
  assert (warn is None) == (int(tag[1:]) >= bench.current_round())
  assert 'embed' in bench.CLASS_INFO
