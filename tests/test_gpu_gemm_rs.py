"""GPU parity of the row-stationary K = 384 GEMM (csrc/gemm_rs.hip; the QKV / key-value projections of the d = 384 blocks, /root/reference/attention.py:154-173)
against an fp64 CPU evaluation on the same 16-bit inputs, and against the tiled kernel it replaces.  Through the C-ABI: spa3d_op_linear with impl 7
(= this kernel or an error).  Tolerance: C is ONE 16-bit rounding of an fp32-accumulated value: element-wise within 1.01 * 2^-8 (bf16) / 2^-11 (fp16) of the fp64 value."""
import ctypes as C
import math

import pytest
import torch

from util import rel_err

pytestmark = pytest.mark.gpu
BF16, F16 = 1, 2


@pytest.fixture(scope='module')
def lib():
  import spa3d
  return spa3d._lib.load()


def _s():
  return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _run(lib, A, B, bias, dtype, impl, pre=None):
  M, K = A.shape; N = B.shape[1]
  out = torch.full((M, N), float('nan'), device='cuda', dtype=A.dtype)
  ws = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
  rc = lib.spa3d_op_linear(A.data_ptr(), B.data_ptr(), bias.data_ptr() if bias is not None else None, pre.data_ptr() if pre is not None else None, out.data_ptr(), M, N, K,
                           2 if pre is not None else 0, dtype, impl,
                           ws.data_ptr(), ws.numel(), _s())
  torch.cuda.synchronize()
  return rc, out


@pytest.mark.parametrize('dtype', [BF16, F16])
@pytest.mark.parametrize('M,N,with_bias', [(256, 256, True), (300, 768, False), (8, 2304, True), (5000, 2304, False), (256 * 260 + 77, 768, True), (70000, 1536, True)])
def test_gemm_rs_matches_fp64(lib, M, N, with_bias, dtype):
  g = torch.Generator().manual_seed(M + N)
  dt = torch.bfloat16 if dtype == BF16 else torch.float16
  A = torch.randn(M, 384, generator=g).to(dt)
  B = (torch.randn(384, N, generator=g) / math.sqrt(384)).to(dt)
  bias = torch.randn(N, generator=g) * 0.5 if with_bias else None
  rc, out = _run(lib, A.cuda(), B.cuda(), bias.cuda() if with_bias else None, dtype, 7)
  assert rc == 0
  ref = A.double() @ B.double() + (bias.double() if with_bias else 0.0)
  o = out.cpu()
  assert torch.isfinite(o.float()).all()
  eps = 2.0 ** -8 if dtype == BF16 else 2.0 ** -11
  assert rel_err(o.float(), ref) < (3e-3 if dtype == BF16 else 4e-4)
  assert bool(((o.double() - ref).abs() <= 1.01 * eps * ref.abs() + 1e-5).all())


def test_gemm_rs_equals_the_tiled_kernel(lib):
  """Same bits as the tiled 8-phase kernel except where fp32 summation order flips a 16-bit rounding."""
  M, N = 256 * 70 + 40, 2304
  g = torch.Generator().manual_seed(3)
  A = torch.randn(M, 384, generator=g).bfloat16().cuda()
  B = (torch.randn(384, N, generator=g) / math.sqrt(384)).bfloat16().cuda()
  rc7, o7 = _run(lib, A, B, None, BF16, 7)
  rc2, o2 = _run(lib, A, B, None, BF16, 6)   # 6 = tiled kernels without the round-4 ones
  assert rc7 == 0 and rc2 == 0
  diff = (o7 != o2).float().mean().item()
  print(f'row-stationary vs tiled: {diff:.2e} of the elements differ')
  assert diff < 5e-3
  assert float((o7.float() - o2.float()).abs().max()) <= 2.0 ** -7 * float(o2.float().abs().max())


def test_gemm_rs_refuses_other_shapes(lib):
  A = torch.randn(512, 512).bfloat16().cuda(); B = torch.randn(512, 768).bfloat16().cuda()
  rc, _ = _run(lib, A, B, None, BF16, 7)
  assert rc != 0
  A = torch.randn(512, 384).bfloat16().cuda(); B = torch.randn(384, 192).bfloat16().cuda()
  rc, _ = _run(lib, A, B, None, BF16, 7)
  assert rc != 0


def _gelu_grad64(x):
  c = math.sqrt(2.0 / math.pi)
  u = c * (x + 0.044715 * x ** 3)
  t = torch.tanh(u)
  return 0.5 * (1.0 + t) + 0.5 * x * (1.0 - t * t) * c * (1.0 + 3.0 * 0.044715 * x * x)


@pytest.mark.parametrize('dtype', [BF16, F16])
@pytest.mark.parametrize('M,N', [(256, 256), (300, 1536), (256 * 130 + 9, 1536), (40000, 768)])
def test_gemm_rs_gelu_grad_epilogue_matches_fp64(lib, M, N, dtype):
  """The MLP backward's dh = (dy . W_out^T) o gelu'(hpre) (attention.py:106 backward): the product is formed in f32 and rounded once."""
  g = torch.Generator().manual_seed(7 * M + N)
  dt = torch.bfloat16 if dtype == BF16 else torch.float16
  A = torch.randn(M, 384, generator=g).to(dt)
  B = (torch.randn(384, N, generator=g) / math.sqrt(384)).to(dt)
  pre = (torch.randn(M, N, generator=g) * 1.5).to(dt)
  rc, out = _run(lib, A.cuda(), B.cuda(), None, dtype, 7, pre=pre.cuda())
  assert rc == 0
  ref = (A.double() @ B.double()) * _gelu_grad64(pre.double())
  o = out.cpu()
  assert torch.isfinite(o.float()).all()
  eps = 2.0 ** -8 if dtype == BF16 else 2.0 ** -11
  assert rel_err(o.float(), ref) < (3e-3 if dtype == BF16 else 4e-4)
  assert bool(((o.double() - ref).abs() <= 1.01 * eps * ref.abs() + 2e-5).all())
  rc6, o6 = _run(lib, A.cuda(), B.cuda(), None, dtype, 6, pre=pre.cuda())   # the tiled kernel's epilogue
  if rc6 == 0:
    diff = (out != o6).float().mean().item()
    print(f'row-stationary vs tiled gelu-grad epilogue: {diff:.2e} of the elements differ')
    assert diff < 1e-2
