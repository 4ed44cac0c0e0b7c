"""GPU parity of the sequence-resident MLP forward (csrc/mlp_fused.hip, /root/reference/attention.py:103-108) against an fp64 CPU
evaluation on the same 16-bit inputs, and against the two tiled GEMMs it replaces.
Tolerances: hpre / h / y are each ONE 16-bit rounding of an fp32-accumulated value: relative Frobenius error <= 3e-3 (bf16: 2^-9 per
element) / 4e-4 (fp16), and element-wise within 1.01 * 2^-8 (bf16) of the fp64 value."""
import ctypes as C
import math

import pytest
import torch

from util import O, rel_err

pytestmark = pytest.mark.gpu
BF16, F16 = 1, 2


@pytest.fixture(scope='module')
def lib():
  import spa3d
  return spa3d._lib.load()


def _s():
  return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _case(M, dtype, seed=0):
  g = torch.Generator().manual_seed(seed)
  dt = torch.bfloat16 if dtype == BF16 else torch.float16
  d, mlp = 384, 1536
  na = torch.randn(M, d, generator=g).to(dt)
  a = torch.randn(M, d, generator=g).to(dt)
  w_in = (torch.randn(d, mlp, generator=g) / math.sqrt(d)).to(dt)
  w_out = (torch.randn(mlp, d, generator=g) / math.sqrt(mlp)).to(dt)
  b_in = torch.randn(mlp, generator=g) * 0.5
  b_out = torch.randn(d, generator=g) * 0.5
  return na, a, w_in, w_out, b_in, b_out


def _run(lib, M, dtype, t):
  na, a, w_in, w_out, b_in, b_out = [x.cuda() for x in t]
  dt = na.dtype
  y = torch.full((M, 384), float('nan'), device='cuda', dtype=dt)
  h = torch.full((M, 1536), float('nan'), device='cuda', dtype=dt)
  hpre = torch.full((M, 1536), float('nan'), device='cuda', dtype=dt)
  ws = torch.empty(8 << 20, dtype=torch.uint8, device='cuda')
  rc = lib.spa3d_op_mlp_fused(na.data_ptr(), a.data_ptr(), w_in.data_ptr(), b_in.data_ptr(), w_out.data_ptr(), b_out.data_ptr(),
                              y.data_ptr(), h.data_ptr(), hpre.data_ptr(), M, 384, 1536, dtype, ws.data_ptr(), ws.numel(), _s())
  assert rc == 0
  torch.cuda.synchronize()
  return y.cpu(), h.cpu(), hpre.cpu()


@pytest.mark.parametrize('dtype', [BF16, F16])
@pytest.mark.parametrize('M', [128, 1000, 8, 40000, 256 * 128 + 72])
def test_mlp_fused_matches_fp64(lib, M, dtype):
  t = _case(M, dtype, seed=M)
  na, a, w_in, w_out, b_in, b_out = t
  y, h, hpre = _run(lib, M, dtype, t)
  assert torch.isfinite(y.float()).all() and torch.isfinite(h.float()).all() and torch.isfinite(hpre.float()).all()
  pre = na.double() @ w_in.double() + b_in.double()
  eps = 2.0 ** -8 if dtype == BF16 else 2.0 ** -11
  tol = 3e-3 if dtype == BF16 else 4e-4
  assert rel_err(hpre.float(), pre) < tol
  assert bool(((hpre.double() - pre).abs() <= 1.01 * eps * pre.abs() + 1e-5).all())
  gl = O.gelu_tanh(pre)
  assert rel_err(h.float(), gl) < tol
  assert bool(((h.double() - gl).abs() <= 1.01 * eps * gl.abs() + 2e-5).all())
  # the second product consumes the ROUNDED h (exactly what the unfused path reads back from HBM)
  yr = a.double() + h.double() @ w_out.double() + b_out.double()
  assert rel_err(y.float(), yr) < tol
  assert bool(((y.double() - yr).abs() <= 1.01 * eps * yr.abs() + 2e-5).all())


def test_mlp_fused_equals_the_two_gemms(lib):
  """Same bits as the MLP-in (dual output) + MLP-out (residual) GEMM pair except where fp32 summation order flips a 16-bit rounding."""
  M = 128 * 37 + 24
  t = _case(M, BF16, seed=5)
  na, a, w_in, w_out, b_in, b_out = [x.cuda() for x in t]
  y, h, hpre = _run(lib, M, BF16, t)
  ws = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
  h2 = torch.empty(M, 1536, device='cuda', dtype=torch.bfloat16)
  y2 = torch.empty(M, 384, device='cuda', dtype=torch.bfloat16)
  assert lib.spa3d_op_linear(na.data_ptr(), w_in.data_ptr(), b_in.data_ptr(), None, h2.data_ptr(), M, 1536, 384, 1, BF16, 2, ws.data_ptr(),
                             ws.numel(), _s()) == 0
  assert lib.spa3d_op_linear(h2.data_ptr(), w_out.data_ptr(), b_out.data_ptr(), a.data_ptr(), y2.data_ptr(), M, 384, 1536, 0, BF16, 2,
                             ws.data_ptr(), ws.numel(), _s()) == 0
  torch.cuda.synchronize()
  frac_h = float((h2.cpu().view(torch.int16) != h.view(torch.int16)).float().mean())
  frac_y = float((y2.cpu().view(torch.int16) != y.view(torch.int16)).float().mean())
  print(f'fused vs GEMM pair: h differs in {frac_h:.2e} of elements, y in {frac_y:.2e}')
  assert frac_h < 2e-2 and frac_y < 5e-2
  assert rel_err(y.float(), y2.float()) < 2e-3


def test_mlp_fused_rejects_other_widths(lib):
  x = torch.zeros(128, 512, device='cuda', dtype=torch.bfloat16)
  f = torch.zeros(2048, device='cuda')
  ws = torch.empty(8 << 20, dtype=torch.uint8, device='cuda')
  rc = lib.spa3d_op_mlp_fused(x.data_ptr(), x.data_ptr(), x.data_ptr(), f.data_ptr(), x.data_ptr(), f.data_ptr(), x.data_ptr(), x.data_ptr(),
                              x.data_ptr(), 128, 512, 2048, BF16, ws.data_ptr(), ws.numel(), _s())
  assert rc == 1


def test_mlp_fused_refuses_misaligned_pointers_and_reports_shape_before_workspace(lib):
  """every operand is accessed with 16-byte vector loads / stores or LDS-DMA -- a misaligned pointer from a C-ABI caller must come back as
  SPA3D_ERR_ARG (1), not fault on the GPU; and a wrong shape is an argument error even when the workspace is too small to say so first."""
  M = 256
  buf = torch.zeros(M * 1536 + 64, device='cuda', dtype=torch.bfloat16)
  w = torch.zeros(384 * 1536, device='cuda', dtype=torch.bfloat16)
  f = torch.zeros(2048, device='cuda')
  ws = torch.empty(32 << 20, dtype=torch.uint8, device='cuda')
  good = [buf.data_ptr(), buf.data_ptr(), w.data_ptr(), f.data_ptr(), w.data_ptr(), f.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr()]
  assert lib.spa3d_op_mlp_fused(*good, M, 384, 1536, BF16, ws.data_ptr(), ws.numel(), _s()) == 0
  for i in (0, 1, 2, 4, 6, 7, 8):  # na, a, w_in, w_out, y, h, hpre
    bad = list(good); bad[i] += 2
    assert lib.spa3d_op_mlp_fused(*bad, M, 384, 1536, BF16, ws.data_ptr(), ws.numel(), _s()) == 1, i
  tiny = torch.empty(1024, dtype=torch.uint8, device='cuda')
  assert lib.spa3d_op_mlp_fused(*good, M, 512, 2048, BF16, tiny.data_ptr(), tiny.numel(), _s()) == 1   # shape error, not ERR_WORKSPACE
  assert lib.spa3d_op_mlp_fused(*good, M, 384, 1536, BF16, tiny.data_ptr(), tiny.numel(), _s()) == 2   # right shape, workspace too small
  torch.cuda.synchronize()
