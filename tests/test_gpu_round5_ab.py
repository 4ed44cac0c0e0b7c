"""In-model A/B of the round-5 large-register-tile kernels (dW: csrc/gemm_tnb.hip; NT: csrc/gemm_ntb.hip) against the 8-wave kernels they replace, on the full-size model of
tests/golden/make_t150_golden.py (T = 150, C = 772; backward of every Dense / DenseGeneral of /root/reference/attention.py:106-107,154-183 and of the input
embedding track_autoencoder_3d.py:123-149).  gemm_impl 9 puts every divisible dW and every eligible NT GEMM (plain / bias epilogue, contraction >= 768) on the new
kernels whatever its M, gemm_impl 3 keeps them on the 8-wave kernels.  The large-tile NT kernel rounds the same fp32 sums as the 8-wave kernels, so the forward
is required to be IDENTICAL, and the parameter gradients may differ by the fp32 summation order of the dW kernels only -- the bounds are tight (1e-5), not statistical.  The model exercises what the op-level test cannot reach: the fused column sums (bias gradients), the q | k | v segment
routing, the row remap of the embedding dW (token rows behind a readout row) and both tile orientations (384 x 256 and 256 x 384).  The profiler's per-launch
records prove that the new kernel ran in the one run and not in the other."""
import ctypes as C
import os
import sys

import pytest
import torch

from util import O, batch_to, product_model, rel_err

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import make_t150_golden as G  # noqa: E402

pytestmark = pytest.mark.gpu
TNB_FLAG = 1 << 20  # ProfRec tag[3] of a dW GEMM that ran on the large-tile kernel (csrc/gemm_fast.hip gemm_tn_bf16)
NTB_FLAG = 1 << 21  # ... of an NT GEMM on the large-tile NT kernel (csrc/gemm_ntb.hip gemm_ntb)


def _run(spa3d, cfg, p, batch, noise, precision, gemm_impl, tmp):
  model = product_model(spa3d, cfg, precision)
  gb = batch_to(batch, 'cuda')
  cast = {'bf16': torch.bfloat16, 'fp16': torch.float16}[precision]
  for k in ('dino_features', 'depth_features'):
    if k in gb:
      gb[k] = gb[k].to(cast)
  gp = O.tree_map(lambda t: t.cuda(), p)
  h = model._handle(*model._dims_from_params(gp))[0]
  lib = spa3d._lib.load()
  spa3d._lib.check(lib.spa3d_set_option(h, b'gemm_impl', float(gemm_impl)), h)
  lib.spa3d_prof_enable(h, 1)
  ld, grads, preds = model.loss_and_grads({'params': gp}, gb, noise=noise.cuda(), return_predictions=True)
  torch.cuda.synchronize()
  lib.spa3d_prof_dump.restype = C.c_int; lib.spa3d_prof_dump.argtypes = [C.c_void_p, C.c_char_p]
  path = os.path.join(tmp, f'prof_{precision}_{gemm_impl}.csv')
  assert lib.spa3d_prof_dump(h, path.encode()) == 0
  lib.spa3d_prof_enable(h, 0)
  spa3d._lib.check(lib.spa3d_set_option(h, b'gemm_impl', 0.0), h)
  tn, nt = [], []
  for line in open(path):
    f = line.strip().split(',')
    if int(f[0]) == 1:  # class TN: tags = M (reduction rows), N, Ki, flags
      tn.append((int(f[4]), int(f[5]), int(f[6]), int(f[7])))
    if int(f[0]) == 0:  # class NT: tags = M, N, K, flags
      nt.append((int(f[4]), int(f[5]), int(f[6]), int(f[7])))
  return float(ld['total_loss']), preds.tracks.clone(), {k: v.clone() for k, v in O.tree_flatten(grads).items()}, tn, nt


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_large_tile_dw_kernel_equals_the_8_wave_kernels_in_model(precision, tmp_path):
  import spa3d
  cfg, p, batch, noise = G.make_inputs('c772')
  new = _run(spa3d, cfg, p, batch, noise, precision, 9, str(tmp_path))
  old = _run(spa3d, cfg, p, batch, noise, precision, 3, str(tmp_path))
  big = [t for t in new[3] if t[3] & TNB_FLAG]
  assert not [t for t in old[3] if t[3] & TNB_FLAG], 'gemm_impl 3 still ran the large-tile kernel'
  shapes = {(t[1], t[2]) for t in big}
  print(f'{precision}: {len(big)} of {len(new[3])} dW launches on the large-tile kernel; (N, Ki) shapes: {sorted(shapes)}')
  # both orientations, both widths, the segmented q|k|v projection (N = 2304) and the remapped embedding dW (N = 384, Ki = 256 / 768) must be among them
  for need in ((2304, 384), (1536, 384), (384, 1536), (384, 768), (384, 256), (2304, 1280), (1280, 1536), (1536, 1280)):
    assert need in shapes, f'dW shape (N, Ki) = {need} did not reach the large-tile kernel: {sorted(shapes)}'
  nbig = [t for t in new[4] if t[3] & NTB_FLAG]
  assert not [t for t in old[4] if t[3] & NTB_FLAG], 'gemm_impl 3 still ran the large-tile NT kernel'
  nshapes = {(t[1], t[2]) for t in nbig}
  print(f'{precision}: {len(nbig)} of {len(new[4])} NT launches on the large-tile kernel; (N, K) shapes: {sorted(nshapes)}')
  # the dX GEMMs of MLP-in and q|k|v in the track encoder (256 x 384 tile), the readout stack's q|k|v forward and dX GEMMs (both tiles)
  for need in ((384, 1536), (384, 2304), (2304, 1280), (1280, 1536), (1280, 2304), (768, 1280)):
    assert need in nshapes, f'NT shape (N, K) = {need} did not reach the large-tile kernel: {sorted(nshapes)}'
  assert new[0] == old[0] and torch.equal(new[1], old[1]), 'the forward must not depend on the dW kernel'
  worst = max((rel_err(new[2][k], old[2][k]), k) for k in new[2] if float(old[2][k].double().norm()) > 0)
  print(f'{precision}: worst gradient leaf, large-tile vs 8-wave dW: {worst}')
  # fp32 accumulation of exact 16-bit products in both kernels; only the order of the fp32 sums (split-M atomics, 16-row quarters) differs
  assert worst[0] < 5e-5, worst  # (run-to-run noise of the fp32 atomics in the scale / bias gradients alone is ~1e-5: DESIGN.md 4a)
