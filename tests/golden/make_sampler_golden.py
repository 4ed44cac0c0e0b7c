"""Generates tests/golden/sampler_golden.npz by RUNNING THE REFERENCE'S OWN CODE for the three pure-NumPy feature
producers of inference.py (lift_2d_to_3d :287-336, sample_dino_features_for_tracks :339-395,
sample_depth_features_for_tracks :398-447) in this container.

`import inference` itself fails here (cv2 / jax / cotracker are not installed -- ordinary ModuleNotFoundError), but these
three functions depend on NumPy only, so the script parses /root/reference/inference.py with `ast`, compiles exactly
those three function definitions and calls them.  No reference source text is stored in this repository: the file is read
at generation time only; what is committed are the inputs and the outputs (the fixture).

NumPy here is 2.2 (NEP 50 promotion): `np.float32 * python_float` stays float32, so the reference's arithmetic is float32
end to end with float32 inputs; the fixture pins THAT behaviour (NumPy 1.x would promote to float64 -- noted in DESIGN.md).

    python tests/golden/make_sampler_golden.py
"""
import ast
import os

import numpy as np

REF = '/root/reference/inference.py'
WANT = ('lift_2d_to_3d', 'sample_dino_features_for_tracks', 'sample_depth_features_for_tracks')


def load_reference_functions():
  tree = ast.parse(open(REF).read())
  fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANT]
  assert len(fns) == len(WANT)
  mod = ast.Module(body=fns, type_ignores=[])
  ns = {'np': np}
  exec(compile(mod, REF, 'exec'), ns)
  return {k: ns[k] for k in WANT}


def main():
  ref = load_reference_functions()
  rng = np.random.default_rng(20260101)
  T, H, W, N, D, Hp, Wp = 5, 28, 42, 23, 40, 2, 3
  depth = (rng.random((T, H, W, 1)) * 5 + 0.5).astype(np.float32)
  dino = rng.standard_normal((T, Hp, Wp, D)).astype(np.float32)
  tracks = np.stack([rng.random((N, T)) * (W + 6) - 3, rng.random((N, T)) * (H + 6) - 3], -1).astype(np.float32)  # some points outside
  tracks[0, 0] = (0.0, 0.0)
  tracks[1, 1] = (W - 1.0, H - 1.0)
  tracks[2, 2] = (7.0, 3.0)          # integer coordinates: weights exactly 0
  tracks[3, 3] = (-2.5, H + 1.25)    # clamped on both axes, extrapolating weights
  out = {
      'depth': depth, 'dino': dino, 'tracks_2d': tracks, 'video_shape': np.array([T, H, W, 3]),
      'intrinsics': np.array([50.0, 45.0, 20.5, 13.25], dtype=np.float64),
      'numpy_version': np.array(np.__version__),
  }
  out['lift_default'] = ref['lift_2d_to_3d'](tracks, depth)
  out['lift_intr'] = ref['lift_2d_to_3d'](tracks, depth, tuple(float(v) for v in out['intrinsics']))
  out['dino_tracks'] = ref['sample_dino_features_for_tracks'](dino, tracks, (T, H, W, 3))
  out['depth_tracks'] = ref['sample_depth_features_for_tracks'](depth, tracks)
  for k in ('lift_default', 'lift_intr', 'dino_tracks', 'depth_tracks'):
    assert out[k].dtype == np.float32
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sampler_golden.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, os.path.getsize(path), 'bytes; numpy', np.__version__)


if __name__ == '__main__':
  main()
