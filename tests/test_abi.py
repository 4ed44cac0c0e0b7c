"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/spa3d.h declares,
and its parameter tree / workspace sizing / error behaviour are what the reference-API mirror relies on.
No compute call is made (no GPU here)."""
import ctypes as C
import os
import re

import pytest
import torch

from util import MINI, O, product_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def spa3d():
  import spa3d as s
  return s


def test_library_exports_every_declared_symbol(spa3d):
  hdr = open(os.path.join(ROOT, 'include', 'spa3d.h')).read()
  declared = set(re.findall(r'\b(spa3d_[a-z0-9_]+)\s*\(', hdr))
  declared -= {'spa3d_ctx'}
  lib = C.CDLL(spa3d._lib.LIB_PATH)
  missing = [n for n in sorted(declared) if not hasattr(lib, n)]
  assert not missing, f'libspa3d_hip.so does not export: {missing}'
  assert declared == set(spa3d._lib.exported_symbols()), 'ctypes binding table and header disagree'
  # and nothing undeclared leaks out of the library: its dynamic symbol table is exactly the header (the fp16 twins of the op entry points are hidden)
  import subprocess
  nm = subprocess.run(['nm', '-D', '--defined-only', spa3d._lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
  exported = {l.split()[-1] for l in nm.splitlines() if l.split()[-1].startswith('spa3d_')}
  assert exported == declared, f'exported but not declared: {sorted(exported - declared)}; declared but not exported: {sorted(declared - exported)}'
  assert spa3d._lib.load().spa3d_version().startswith(b'spa3d-hip')


def test_no_cpu_fallback_and_no_oracle_import_in_product():
  pkg = os.path.join(ROOT, '3dspa_code_amd')
  for fn in os.listdir(pkg):
    if fn.endswith('.py'):
      src = open(os.path.join(pkg, fn)).read()
      assert 'oracle' not in src.replace('no CPU fallback', ''), f'{fn} must not reference the oracle'


def test_cpu_tensors_are_refused_loudly(spa3d):
  cfg = O.Config(**MINI, use_dino=False, use_depth=False)
  model = product_model(spa3d, cfg, 'fp32')
  batch = O.synthetic_batch(2, 4, 3, 8)
  h, leaves, n = model._handle(0, 0)
  params = model.tree_from_flat(torch.zeros(n), 0, 0)
  with pytest.raises(spa3d._lib.Spa3dError):
    model.apply({'params': params}, batch)


def test_leaf_tree_matches_flax_names_and_shapes(spa3d):
  for dino, depth in ((0, 0), (768, 1), (768, 256)):
    m = spa3d.TrackAutoEncoder3D(precision='bf16')
    _, leaves, n = m._handle(dino, depth)
    cfg = O.Config(use_dino=dino > 0, use_depth=depth > 0)
    ref = O.tree_flatten(O.init_params(cfg, with_dino=dino > 0, with_depth=depth > 0, depth_dim=max(depth, 1)))
    got = {name: shape for name, shape, _ in leaves}
    assert set(got) == set(ref)
    for k, v in ref.items():
      assert tuple(v.shape) == tuple(got[k]), k
    # leaves are 256-byte aligned, non-overlapping, inside the flat buffer
    end = 0
    for name, shape, off in leaves:
      assert off % 64 == 0 and off >= end
      end = off + int(torch.tensor(shape).prod())
    assert end <= n


def test_param_tree_round_trip_and_errors(spa3d):
  cfg = O.Config(**MINI, use_dino=True, use_depth=True, dino_feature_dim=6, depth_feature_dim=2)
  model = product_model(spa3d, cfg, 'fp32')
  p = O.init_params(cfg, seed=0, depth_dim=2)
  flat = model.flat_from_tree(p, device='cpu')
  tree = model.tree_from_flat(flat, 6, 2)
  for k, v in O.tree_flatten(p).items():
    assert torch.equal(O.tree_flatten(tree)[k], v)
  assert model.flat_from_tree(tree) is flat  # zero-copy for a ParamTree
  bad = O.tree_unflatten({k: v for k, v in O.tree_flatten(p).items() if k != 'compressor/bias'})
  with pytest.raises(KeyError):
    model.flat_from_tree(bad, device='cpu')
  p['compressor']['bias'] = torch.zeros(3)
  with pytest.raises(ValueError):
    model.flat_from_tree(p, device='cpu')
  with pytest.raises(ValueError):
    spa3d.TrackAutoEncoder3D(precision='fp8')


def test_workspace_sizing_is_monotone_and_covers_baseline_configs(spa3d):
  lib = spa3d._lib.load()
  m = spa3d.TrackAutoEncoder3D(precision='bf16')
  h = m._handle(768, 1)[0]
  need = [lib.spa3d_workspace_bytes(h, 64, 2048, 512, 150, c, 1) for c in (1, 2, 4, 8)]
  assert all(b > a for a, b in zip(need, need[1:]))
  assert need[0] < 40 << 30, 'one sample of BASELINE configs[2] must fit comfortably in 288 GB'
  assert need[3] < 230 << 30
  fwd = lib.spa3d_workspace_bytes(h, 64, 2048, 512, 150, 1, 0)
  assert 0 < fwd < need[0]
  # the stress config (8192+2048 tracks, T=300) fits one sample at a time
  m2 = spa3d.TrackAutoEncoder3D(num_output_frames=300, precision='bf16')
  h2 = m2._handle(768, 1)[0]
  assert 0 < lib.spa3d_workspace_bytes(h2, 8, 8192, 2048, 300, 1, 1) < 250 << 30
  assert lib.spa3d_workspace_bytes(h, 0, 1, 1, 1, 1, 1) == -1


def test_create_rejects_bad_configs(spa3d):
  lib = spa3d._lib.load()
  m = spa3d.TrackAutoEncoder3D()
  m.num_heads = 7  # attention.py:147-148: num_heads must divide qkv_size
  with pytest.raises(spa3d._lib.Spa3dError):
    m._handle(0, 0)
  h = C.c_void_p()
  assert lib.spa3d_create(None, C.byref(h)) == 1


def test_lr_schedule_matches_oracle(spa3d):
  f = spa3d.create_learning_rate_schedule(1e-4, 10000, 1000000)
  for s in (0, 1, 9999, 10000, 10001, 500000, 1000000, 2000000):
    assert abs(f(s) - O.lr_schedule(s, 1e-4, 10000, 1000000)) < 1e-18


def test_options_plan_stats_and_gradient_segments(spa3d):
  """Round-3 additions to the boundary (no compute): per-handle options refuse unknown names, plan statistics start at zero, and the three
  gradient segments (spa3d_grad_segments: the order in which the flat gradient buffer becomes final in the last chunk's backward) tile the
  buffer at leaf boundaries with the Flax module groups on the documented sides."""
  lib = spa3d._lib.load()
  m = spa3d.TrackAutoEncoder3D(precision='bf16')
  h, leaves, n = m._handle(768, 1)
  assert lib.spa3d_set_option(h, b'prune', 0.0) == 0 and lib.spa3d_set_option(h, b'ro_share', 1.0) == 0
  assert lib.spa3d_set_option(h, b'no_such_option', 1.0) == 1 and b'unknown option' in lib.spa3d_last_error(h)
  assert lib.spa3d_set_option(h, b'loss_scale', 2.0) == 1  # fp16 handles only
  o = (C.c_double * 4)()
  assert lib.spa3d_plan_stats(h, o) == 0 and list(o) == [0.0, 0.0, 0.0, 0.0]
  b4 = (C.c_int64 * 4)()
  assert lib.spa3d_grad_segments(h, b4) == 0
  b = list(b4)
  assert b[0] == 0 and b[3] == n and 0 < b[1] < b[2] < n
  offs = {name: off for name, _, off in leaves}
  assert b[1] == min(off for name, off in offs.items() if name.startswith('tracks_to_latents/'))
  assert b[2] == min(off for name, off in offs.items() if name.startswith('track_readout_attn/'))
  for name, off in offs.items():
    top = name.split('/')[0]
    seg = 0 if off < b[1] else (1 if off < b[2] else 2)
    want = {'initializer': 0, 'input_readout_token': 0, 'track_token_projection': 0, 'dino_projection': 0, 'depth_projection': 0, 'input_track_transformer': 0,
            'tracks_to_latents': 1, 'compressor': 1, 'decompressor': 1, 'decompress_attn': 1, 'track_readout_attn': 2, 'query_encoder': 2, 'track_predictor': 2}[top]
    assert seg == want, name
  assert lib.spa3d_set_grad_events(h, None, None) == 0 and lib.spa3d_set_loss_scale_state(h, None) == 0
