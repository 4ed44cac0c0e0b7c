"""Poisoned-workspace mode ("NaN-poisoned workspaces"): spa3d_set_option(h, "poison", 1) fills everything a sample chunk may
bump-allocate with 0xFF bytes (NaN as bf16, fp16 and fp32) before the chunk runs.  A read of a row this call has not written -- the
rounded-up tails of the pruned GEMMs (rows_g = (rows + 7) & ~7, csrc/model.hip), a buffer that silently relied on the previous chunk's values or on
zero-initialised memory -- then shows up as NaN in an output or a gradient instead of as a plausible stale number.  Every case must give the
SAME forward values as the unpoisoned run, bit for bit (the forward has no float atomics), finite gradients, and gradients equal up to the
atomic-order noise of two runs.  Cases: ragged token pruning (c772: boundary_frame 150 / 97), shared readout rows (c772_q320), C = 4,
fp16 at T = 300, and a multi-chunk run (chunk = 1: the bump allocator reuses the same addresses chunk after chunk)."""
import os
import sys

import pytest
import torch

from util import O, batch_to, product_model, rel_err

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import make_t150_golden as G  # noqa: E402

pytestmark = pytest.mark.gpu


def _run(spa3d, cfg, p, batch, noise, precision, poison, chunk=0):
  model = product_model(spa3d, cfg, precision)
  gb = batch_to(batch, 'cuda')
  cast = {'bf16': torch.bfloat16, 'fp16': torch.float16}.get(precision)
  if cast is not None:
    for k in ('dino_features', 'depth_features'):
      if k in gb:
        gb[k] = gb[k].to(cast)
  gp = O.tree_map(lambda t: t.cuda(), p)
  dims = model._dims_from_params(gp)
  h = model._handle(*dims)[0]
  lib = spa3d._lib.load()
  spa3d._lib.check(lib.spa3d_set_option(h, b'poison', float(poison)), h)
  spa3d._lib.check(lib.spa3d_set_option(h, b'chunk', float(chunk)), h)
  ld, grads, preds = model.loss_and_grads({'params': gp}, gb, noise=noise.cuda(), return_predictions=True)
  lat = model.apply({'params': gp}, gb, method=model.encode)
  torch.cuda.synchronize()
  spa3d._lib.check(lib.spa3d_set_option(h, b'poison', 0.0), h)
  spa3d._lib.check(lib.spa3d_set_option(h, b'chunk', 0.0), h)
  return float(ld['total_loss']), preds.tracks.clone(), preds.visible_logits.clone(), lat.clone(), {k: v.clone() for k, v in O.tree_flatten(grads).items()}


@pytest.mark.parametrize('case,precision,chunk', [('c772', 'bf16', 0), ('c772', 'bf16', 1), ('c772_q320', 'bf16', 0), ('c4', 'bf16', 1),
                                                  ('c772_t300', 'fp16', 0), ('c772', 'fp32', 1)])
def test_poisoned_workspace_changes_nothing(case, precision, chunk):
  import spa3d
  cfg, p, batch, noise = G.make_inputs(case)
  a = _run(spa3d, cfg, p, batch, noise, precision, poison=0, chunk=chunk)
  b = _run(spa3d, cfg, p, batch, noise, precision, poison=1, chunk=chunk)
  assert all(bool(torch.isfinite(t).all()) for t in (b[1], b[2], b[3])) and all(bool(torch.isfinite(v).all()) for v in b[4].values())
  assert b[0] == a[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]), 'forward values changed under poison'
  worst = max((rel_err(b[4][k], a[4][k]), k) for k in a[4] if float(a[4][k].double().norm()) > 1e-12)
  print(f'{case} {precision} chunk={chunk}: poisoned vs clean, loss {b[0]} == {a[0]}; worst gradient leaf {worst}')
  assert worst[0] < (2e-3 if precision != 'fp32' else 1e-4)  # two runs differ by the order of the float-atomic dW / scale-gradient sums only
