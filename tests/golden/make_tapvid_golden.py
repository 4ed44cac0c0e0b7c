"""Generates tests/golden/tapvid_adapter_golden.npz by RUNNING THE REFERENCE'S OWN `convert_predictions_to_tapvid3d_format`
(evaluate_tapvid3d.py:39-59) in this container.

`import evaluate_tapvid3d` fails here (jax / flax / tapnet / absl are not installed -- ordinary ModuleNotFoundError), but this function
depends on NumPy only, so the script parses /root/reference/evaluate_tapvid3d.py with `ast`, compiles exactly that one function
definition and calls it on synthetic predictions (a plain object with `.tracks` / `.visible_logits`, which is all it reads).  No
reference source text is stored in this repository: the file is read at generation time only; the fixture holds inputs and outputs.
Logits exactly 0, -0.0 and tiny values of both signs are included: "occluded" is `logit <= 0.0` (evaluate_tapvid3d.py:55).

`prepare_3d_batch` (data_loader.py:56-110) is NOT pinned this way: its last step wraps arrays with `jax.numpy.array`, and jax is absent
(no stand-in is substituted); 3dspa_code_amd/data.py restates it with the same NumPy RNG call sequence -- parity unpinned.

    python tests/golden/make_tapvid_golden.py
"""
import ast
import os
import types

import numpy as np

REF = '/root/reference/evaluate_tapvid3d.py'
NAME = 'convert_predictions_to_tapvid3d_format'


def load_reference_function():
  tree = ast.parse(open(REF).read())
  fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == NAME]
  assert len(fns) == 1
  ns = {'np': np}
  exec(compile(ast.Module(body=fns, type_ignores=[]), REF, 'exec'), ns)
  return ns[NAME]


def main():
  fn = load_reference_function()
  rng = np.random.default_rng(20260102)
  B, Q, T = 2, 7, 5
  tracks = rng.standard_normal((B, Q, T, 3)).astype(np.float32)
  logits = rng.standard_normal((B, Q, T, 1)).astype(np.float32)
  logits[0, 0, 0, 0] = 0.0
  logits[0, 1, 1, 0] = -0.0
  logits[0, 2, 2, 0] = 1e-30
  logits[0, 3, 3, 0] = -1e-30
  qp = rng.random((B, Q, 4)).astype(np.float32)
  pred_tracks, pred_occ = fn(types.SimpleNamespace(tracks=tracks, visible_logits=logits), qp)
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tapvid_adapter_golden.npz')
  np.savez_compressed(path, tracks=tracks, visible_logits=logits, query_points=qp, pred_tracks=pred_tracks, pred_occluded=pred_occ)
  print('wrote', path, os.path.getsize(path), 'bytes;', pred_tracks.shape, pred_occ.shape, pred_occ.dtype)


if __name__ == '__main__':
  main()
