"""GPU parity of every single-op C-ABI entry point (spa3d_op_*) against the CPU oracle.
fp32 ops use the exact-f32 MFMA path: tolerance 1e-4 absolute on O(1) values (north_star) -- in practice ~1e-6.
bf16 ops: inputs are rounded to bf16 first, the oracle runs in fp64 on those rounded inputs, and the tolerance
(2e-2 relative Frobenius, written per test) covers bf16 output rounding + bf16 intermediate P / normalised q,k."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from util import O, max_abs, rel_err

pytestmark = pytest.mark.gpu

F32, BF16, F16 = 0, 1, 2


@pytest.fixture(scope='module')
def lib():
  import spa3d
  return spa3d._lib.load()


def _s():
  return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ws(nbytes=256 << 20):
  return torch.empty(nbytes, dtype=torch.uint8, device='cuda')


def _dt(dtype):
  return {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}[dtype]


# ------------------------------------------------------------------------------------------------
def test_sin_embed_matches_oracle_and_kat(lib):
  g = torch.Generator().manual_seed(0)
  x = torch.rand(1000, 4, generator=g) * 2 - 0.5
  x[0] = 0.0
  out = torch.empty(1000, 256, device='cuda')
  assert lib.spa3d_op_sin_embed(x.cuda().data_ptr(), 1000, 4, 32, out.data_ptr(), F32, _s()) == 0
  ref = O.sinusoidal_embedding(x)  # fp32 argument arithmetic, fp32 sin
  # identical argument rounding; the two libm sinf differ by <= 2 ulp of 1.0
  assert max_abs(out, ref) < 5e-7
  # KAT: x = 0 -> [0]*32 + [sin(fl32(pi/2))]*32 per coordinate, layout "(coords d)"
  row = out[0].cpu().view(4, 64)
  assert torch.all(row[:, :32] == 0) and torch.allclose(row[:, 32:], torch.ones(4, 32), atol=1e-7)


@pytest.mark.parametrize('dtype', [F32, BF16])
@pytest.mark.parametrize('M,N,K,act,res', [(200, 96, 64, 0, False), (130, 600, 1280, 0, True), (77, 50, 33, 1, True),
                                           (1024, 384, 256, 1, False)])
def test_linear_generic(lib, dtype, M, N, K, act, res):
  g = torch.Generator().manual_seed(1)
  A = torch.randn(M, K, generator=g).to(_dt(dtype))
  B = (torch.randn(K, N, generator=g) / math.sqrt(K)).to(_dt(dtype))
  bias = torch.randn(N, generator=g)
  R = torch.randn(M, N, generator=g).to(_dt(dtype)) if res else None
  Cd = torch.empty(M, N, device='cuda', dtype=_dt(dtype))
  ws = _ws()
  Ad, Bd, bd = A.cuda(), B.cuda(), bias.cuda()
  Rd = R.cuda() if res else None
  rc = lib.spa3d_op_linear(Ad.data_ptr(), Bd.data_ptr(), bd.data_ptr(), Rd.data_ptr() if res else None, Cd.data_ptr(), M, N, K, act,
                           dtype, 1, ws.data_ptr(), ws.numel(), _s())
  assert rc == 0
  ref = A.double() @ B.double() + bias.double()
  if act:
    ref = O.gelu_tanh(ref)
  if res:
    ref = ref + R.double()
  if dtype == F32:
    assert max_abs(Cd, ref) < 1e-4
  else:
    assert rel_err(Cd.float(), ref) < 1e-2  # one bf16 rounding of the output (2^-9 relative)


@pytest.mark.parametrize('dtype', [F32, BF16])
@pytest.mark.parametrize('M,N,K', [(300, 96, 64), (5000, 130, 70)])
def test_linear_bwd_generic(lib, dtype, M, N, K):
  g = torch.Generator().manual_seed(2)
  A = torch.randn(M, K, generator=g).to(_dt(dtype))
  B = (torch.randn(K, N, generator=g) / math.sqrt(K)).to(_dt(dtype))
  dC = torch.randn(M, N, generator=g).to(_dt(dtype))
  dA = torch.empty(M, K, device='cuda', dtype=_dt(dtype))
  dB = torch.empty(K, N, device='cuda')
  db = torch.empty(N, device='cuda')
  ws = _ws()
  Ad, Bd, dCd = A.cuda(), B.cuda(), dC.cuda()
  rc = lib.spa3d_op_linear_bwd(Ad.data_ptr(), Bd.data_ptr(), dCd.data_ptr(), dA.data_ptr(), dB.data_ptr(), db.data_ptr(), M, N, K, dtype,
                               1, ws.data_ptr(), ws.numel(), _s())
  assert rc == 0
  rA = dC.double() @ B.double().T
  rB = A.double().T @ dC.double()
  rb = dC.double().sum(0)
  tol = 1e-5 if dtype == F32 else 1e-2
  assert rel_err(dA.float(), rA) < tol
  assert rel_err(dB, rB) < (1e-5 if dtype == F32 else 1e-5)  # fp32 accumulation of exact bf16 products
  assert rel_err(db, rb) < 1e-5


@pytest.mark.parametrize('dtype', [F32, BF16])
@pytest.mark.parametrize('d', [384, 512, 1152, 1280, 48])
def test_layernorm_fwd_bwd(lib, dtype, d):
  rows = 333
  g = torch.Generator().manual_seed(3)
  x = (torch.randn(rows, d, generator=g) * 2 + 0.5).to(_dt(dtype))
  scale = 1 + 0.1 * torch.randn(d, generator=g)
  dy = torch.randn(rows, d, generator=g).to(_dt(dtype))
  xd, sd, dyd = x.cuda(), scale.cuda(), dy.cuda()
  y = torch.empty_like(xd)
  stats = torch.empty(rows, 2, device='cuda')
  assert lib.spa3d_op_layernorm(xd.data_ptr(), sd.data_ptr(), y.data_ptr(), stats.data_ptr(), rows, d, dtype, _s()) == 0
  xr = x.double().requires_grad_(True)
  sr = scale.double().requires_grad_(True)
  yr = O.layer_norm(xr, sr)
  yr.backward(dy.double())
  dx = torch.empty_like(xd)
  ds = torch.zeros(d, device='cuda')
  assert lib.spa3d_op_layernorm_bwd(xd.data_ptr(), sd.data_ptr(), stats.data_ptr(), dyd.data_ptr(), dx.data_ptr(), ds.data_ptr(), rows, d,
                                    dtype, _s()) == 0
  if dtype == F32:
    assert max_abs(y, yr.detach()) < 1e-5
    assert max_abs(dx, xr.grad) < 1e-5
    assert rel_err(ds, sr.grad) < 1e-5
  else:
    assert rel_err(y.float(), yr.detach()) < 1e-2
    assert rel_err(dx.float(), xr.grad) < 1e-2
    assert rel_err(ds, sr.grad) < 1e-2


def _attn_ref(q, k, v, sq, sk, km, H, Dh):
  """oracle attention core on [nseq,S,H*Dh] tensors (attention.py:166-175)."""
  nseq, Sq, _ = q.shape
  Sk = k.shape[1]
  qh = O.rms_norm(q.view(nseq, Sq, H, Dh), sq)
  kh = O.rms_norm(k.view(nseq, Sk, H, Dh), sk)
  vh = v.view(nseq, Sk, H, Dh)
  mask = None if km is None else km[:, None, None, :].expand(nseq, H, Sq, Sk)
  return O.dot_product_attention(qh, kh, vh, mask).reshape(nseq, Sq, H * Dh)


@pytest.mark.parametrize('dtype', [F32, BF16])
@pytest.mark.parametrize('nseq,Sq,Sk,H,Dh,masked,packed', [(5, 25, 25, 8, 96, True, True), (3, 129, 129, 8, 96, False, True),
                                                          (2, 128, 200, 8, 96, False, False), (4, 9, 9, 2, 16, True, True),
                                                          (2, 151, 151, 8, 96, True, True)])
def test_attention_fwd_bwd_generic(lib, dtype, nseq, Sq, Sk, H, Dh, masked, packed):
  E = H * Dh
  g = torch.Generator().manual_seed(4)
  dt = _dt(dtype)
  if packed:  # q|k|v interleaved per token like the fused QKV projection output
    qkv = torch.randn(nseq, Sq, 3 * E, generator=g).to(dt)
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    ldq = ldk = ldv = 3 * E
    qkvd = qkv.cuda()
    qd, kd, vd = qkvd[..., :E], qkvd[..., E:2 * E], qkvd[..., 2 * E:]
  else:
    q = torch.randn(nseq, Sq, E, generator=g).to(dt)
    kv = torch.randn(nseq, Sk, 2 * E, generator=g).to(dt)
    k, v = kv[..., :E], kv[..., E:]
    ldq, ldk, ldv = E, 2 * E, 2 * E
    qd = q.cuda()
    kvd = kv.cuda()
    kd, vd = kvd[..., :E], kvd[..., E:]
  sq = 1 + 0.2 * torch.randn(Dh, generator=g)
  sk = 1 + 0.2 * torch.randn(Dh, generator=g)
  km = None
  if masked:
    km = (torch.rand(nseq, Sk, generator=g) < 0.8).float()
    km[:, 0] = 1.0
    km[0, 1:] = 0.0  # a sequence where only the readout key is visible
  d_o = torch.randn(nseq, Sq, E, generator=g).to(dt)
  sqd, skd = sq.cuda(), sk.cuda()
  kmd = km.cuda() if masked else None
  o = torch.empty(nseq, Sq, E, device='cuda', dtype=dt)
  ws = _ws(512 << 20)
  rc = lib.spa3d_op_attention(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), ldq, ldk, ldv, sqd.data_ptr(), skd.data_ptr(),
                              kmd.data_ptr() if masked else None, nseq, Sq, Sk, H, Dh, o.data_ptr(), None, dtype, 1, ws.data_ptr(),
                              ws.numel(), _s())
  assert rc == 0
  qr = q.double().contiguous().requires_grad_(True)
  kr = k.double().contiguous().requires_grad_(True)
  vr = v.double().contiguous().requires_grad_(True)
  sqr = sq.double().requires_grad_(True)
  skr = sk.double().requires_grad_(True)
  ref = _attn_ref(qr, kr, vr, sqr, skr, km, H, Dh)
  ref.backward(d_o.double())
  # backward
  dod = d_o.cuda()
  if packed:
    dqkv = torch.zeros(nseq, Sq, 3 * E, device='cuda', dtype=dt)
    dq, dk, dv = dqkv[..., :E], dqkv[..., E:2 * E], dqkv[..., 2 * E:]
  else:
    dq = torch.zeros(nseq, Sq, E, device='cuda', dtype=dt)
    dkv = torch.zeros(nseq, Sk, 2 * E, device='cuda', dtype=dt)
    dk, dv = dkv[..., :E], dkv[..., E:]
  dsq = torch.zeros(Dh, device='cuda')
  dsk = torch.zeros(Dh, device='cuda')
  rc = lib.spa3d_op_attention_bwd(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), ldq, ldk, ldv, sqd.data_ptr(), skd.data_ptr(),
                                  kmd.data_ptr() if masked else None, nseq, Sq, Sk, H, Dh, None, None, dod.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                                  dv.data_ptr(), dsq.data_ptr(), dsk.data_ptr(), dtype, 1, ws.data_ptr(), ws.numel(), _s())
  assert rc == 0
  if dtype == F32:
    assert max_abs(o, ref.detach()) < 1e-5
    for got, want in ((dq, qr.grad), (dk, kr.grad), (dv, vr.grad), (dsq, sqr.grad), (dsk, skr.grad)):
      assert rel_err(got.float(), want) < 1e-4
  else:
    assert rel_err(o.float(), ref.detach()) < 2e-2
    for got, want in ((dq, qr.grad), (dk, kr.grad), (dv, vr.grad), (dsq, sqr.grad), (dsk, skr.grad)):
      assert rel_err(got.float(), want) < 3e-2


def test_uniform_noise_matches_threefry_restatement(lib):
  from oracle import np_blocks as NB
  n = 2 * 8 * 16 + 1  # odd length exercises the padding rule
  out = torch.empty(n, device='cuda')
  assert lib.spa3d_uniform_noise(out.data_ptr(), n, 0, 0, _s()) == 0
  ref = NB.jax_uniform_legacy((n,))
  assert np.array_equal(out.cpu().numpy(), ref)


def test_adamw_step_matches_oracle(lib):
  g = torch.Generator().manual_seed(5)
  n = 100003
  p = torch.randn(n, generator=g)
  gr = torch.randn(n, generator=g) * 0.01
  m = torch.randn(n, generator=g) * 0.001
  v = torch.rand(n, generator=g) * 1e-4
  pd, gd, md, vd = p.cuda(), gr.cuda(), m.cuda(), v.cuda()
  scratch = torch.zeros(1024, device='cuda')
  assert lib.spa3d_adamw_step(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, 3e-4, 7, 1.0, 0.9, 0.999, 1e-8, 0.01,
                              scratch.data_ptr(), _s()) == 0
  P, G, M, V = {'a': p.double().clone()}, {'a': gr.double()}, {'a': m.double().clone()}, {'a': v.double().clone()}
  gn = O.adamw_step(P, G, M, V, step=7, lr=3e-4)
  assert abs(float(scratch[0]) - gn) / gn < 1e-5
  assert max_abs(pd, P['a']) < 1e-6 and max_abs(md, M['a']) < 1e-7 and max_abs(vd, V['a']) < 1e-9


@pytest.mark.parametrize('step,skipped', [(7, 3), (5, 5), (2, 9)])
def test_adamw_step_bias_correction_counts_applied_updates(lib, step, skipped):
  """scratch[3] = updates skipped so far (non-finite gradient norm).  The Adam bias correction then runs at step + 1 - skipped -- the number of
  updates actually applied to m and v -- clamped at 1 (include/spa3d.h, spa3d_adamw_step): the update must equal the oracle's (optax chain, train.py:239-242) at
  step - skipped, and at step 0 when the counter has run past the step count (a resumed scratch that disagrees with `step`: the clamp, not a division by zero)."""
  g = torch.Generator().manual_seed(6)
  n = 50021
  p = torch.randn(n, generator=g); gr = torch.randn(n, generator=g) * 0.01
  m = torch.randn(n, generator=g) * 0.001; v = torch.rand(n, generator=g) * 1e-4
  pd, gd, md, vd = p.cuda(), gr.cuda(), m.cuda(), v.cuda()
  scratch = torch.zeros(1024, device='cuda'); scratch[3] = float(skipped)
  assert lib.spa3d_adamw_step(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, 3e-4, step, 1.0, 0.9, 0.999, 1e-8, 0.01,
                              scratch.data_ptr(), _s()) == 0
  P, G, M, V = {'a': p.double().clone()}, {'a': gr.double()}, {'a': m.double().clone()}, {'a': v.double().clone()}
  O.adamw_step(P, G, M, V, step=max(step - skipped, 0), lr=3e-4)
  assert float(scratch[2]) == 0.0 and float(scratch[3]) == float(skipped)   # a finite step is applied and does not touch the counter
  assert max_abs(pd, P['a']) < 1e-6 and max_abs(md, M['a']) < 1e-7 and max_abs(vd, V['a']) < 1e-9


# ------------------------------------------------------------------------------------------------ tiled bf16 kernels (impl=2)
@pytest.mark.parametrize('M,N,K,act,res,bias', [(1000, 384, 256, 0, False, True), (4133, 2304, 384, 0, False, False),
                                                (777, 1536, 384, 1, False, True), (2050, 384, 1536, 0, True, True),
                                                (300, 600, 1280, 0, False, True), (129, 1280, 12352, 0, False, True)])
def test_linear_tiled_nt(lib, M, N, K, act, res, bias, impl=2):
  g = torch.Generator().manual_seed(11)
  A = torch.randn(M, K, generator=g).bfloat16()
  B = (torch.randn(K, N, generator=g) / math.sqrt(K)).bfloat16()
  bs = torch.randn(N, generator=g) if bias else None
  R = torch.randn(M, N, generator=g).bfloat16() if res else None
  Ad, Bd = A.cuda(), B.cuda()
  bd = bs.cuda() if bias else None
  Rd = R.cuda() if res else None
  Cd = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16)
  ws = _ws()
  rc = lib.spa3d_op_linear(Ad.data_ptr(), Bd.data_ptr(), bd.data_ptr() if bias else None, Rd.data_ptr() if res else None, Cd.data_ptr(),
                           M, N, K, act, BF16, impl, ws.data_ptr(), ws.numel(), _s())
  assert rc == 0
  ref = A.double() @ B.double()
  if bias:
    ref = ref + bs.double()
  if act:
    ref = O.gelu_tanh(ref)
  if res:
    ref = ref + R.double()
  assert not torch.isnan(Cd.float()).any()
  assert rel_err(Cd.float(), ref) < 4e-3  # bf16 output rounding only (2^-9); accumulation is fp32
  assert max_abs(Cd.float(), ref) < 0.05 * float(ref.abs().max())


@pytest.mark.parametrize('M,N,K', [(5000, 384, 256), (3333, 2304, 384), (20000, 128, 64), (1100, 600, 1280), (257, 96, 512)])
def test_linear_bwd_tiled(lib, M, N, K, impl=2):
  """dA = dC.B^T on the NT kernel (needs N % 64 == 0, else generic), dB = A^T.dC on the transposed-read TN kernel."""
  g = torch.Generator().manual_seed(12)
  A = torch.randn(M, K, generator=g).bfloat16()
  B = (torch.randn(K, N, generator=g) / math.sqrt(K)).bfloat16()
  dC = torch.randn(M, N, generator=g).bfloat16()
  Ad, Bd, dCd = A.cuda(), B.cuda(), dC.cuda()
  dA = torch.full((M, K), float('nan'), device='cuda', dtype=torch.bfloat16)
  dB = torch.full((K, N), float('nan'), device='cuda')
  ws = _ws()
  impl_a = impl if (N % 64 == 0 and M * K >= 128 * 128) else 0
  rc = lib.spa3d_op_linear_bwd(Ad.data_ptr(), Bd.data_ptr(), dCd.data_ptr(), dA.data_ptr(), None, None, M, N, K, BF16, impl_a,
                               ws.data_ptr(), ws.numel(), _s())
  assert rc == 0
  rc = lib.spa3d_op_linear_bwd(Ad.data_ptr(), Bd.data_ptr(), dCd.data_ptr(), None, dB.data_ptr(), None, M, N, K, BF16, impl,
                               ws.data_ptr(), ws.numel(), _s())
  assert rc == 0
  assert rel_err(dA.float(), dC.double() @ B.double().T) < 4e-3
  rB = A.double().T @ dC.double()
  assert rel_err(dB, rB) < 1e-5  # exact bf16 products, fp32 accumulate + fp32 atomics
  assert max_abs(dB, rB) < 1e-3 * float(rB.abs().max()) + 1e-4


@pytest.mark.parametrize('M,N,K', [(5000, 384, 256), (3333, 2304, 384), (20000, 128, 64), (1100, 600, 1280), (257, 96, 512),
                                   (9000, 768, 1280), (70001, 384, 768), (4097, 1536, 384), (640, 1280, 1536)])
def test_linear_bwd_tiled_8phase(lib, M, N, K):
  """dB = A^T.dC on the 8-phase TN kernels (256x256 / 128x384 / 384x128 tiles, ring of 16-row quarters), forced for any M (impl 3)"""
  test_linear_bwd_tiled(lib, M, N, K, impl=3)


@pytest.mark.parametrize('M,N,K', [(3333, 2304, 384), (9000, 768, 1280), (70001, 384, 768), (4097, 1536, 384), (640, 1280, 1536), (256, 768, 768),
                                   (100000, 256, 384), (288, 384, 256), (12320, 1280, 2304), (544, 1536, 1280)])
def test_linear_bwd_large_register_tile(lib, M, N, K):
  """dB = A^T.dC on the round-5 large-register-tile TN kernel (csrc/gemm_tnb.hip: 384 x 256 / 256 x 384 workgroup tiles, 384 accumulators per wave, the M % 32
  rows through the tail kernel), forced for any M (impl 9); both tile orientations, one and several tiles per dimension, M below one ring (6 quarters) and
  M with and without a tail.  Backward of a Dense, /root/reference/attention.py:106-107,154-183."""
  test_linear_bwd_tiled(lib, M, N, K, impl=9)


@pytest.mark.parametrize('M,N,K,bias', [(1000, 384, 256, True), (4133, 384, 1536, False), (256, 384, 64, False), (70001, 384, 768, True), (2050, 768, 2304, False),
                                        (777, 1280, 1536, True), (384, 256, 96, False), (100003, 256, 128, True), (5000, 1536, 1280, False), (255, 1152, 320, True),
                                        (1, 384, 64, True), (33, 256, 2304, False), (65537, 384, 1536, False)])
def test_linear_large_register_tile_nt(lib, M, N, K, bias):
  """C = A.W (+ bias) on the round-5 large-register-tile NT kernel (csrc/gemm_ntb.hip; impl 10 = that kernel or an error): both workgroup tiles (256 x 384 when
  384 | N, else 384 x 256), one and several n-tiles, fewer tiles than workgroups and several tiles per workgroup (the phase pipeline runs across tiles), M with a
  ragged last tile (rows past M are read clamped and their stores dropped by the buffer range check), K from two phases up.  attention.py:106-107,154-183."""
  test_linear_tiled_nt(lib, M, N, K, 0, False, bias, impl=10)


@pytest.mark.parametrize('M,N,K', [(5000, 1536, 384), (3333, 2304, 384), (70001, 768, 384), (1100, 1536, 1280), (257, 64, 768), (30000, 2304, 1280)])
def test_linear_bwd_dx_large_register_tile(lib, M, N, K):
  """dA = dC.B^T on the large-register-tile NT kernel (impl 10: B^T packed from the [K][N] weight with swapped strides) -- the dX GEMMs of MLP-in and of
  q|k|v in the track encoder (output 384 wide, contraction 1536 / 2304) and the 1280-wide ones of the readout stack."""
  g = torch.Generator().manual_seed(13)
  B = (torch.randn(K, N, generator=g) / math.sqrt(K)).bfloat16()
  dC = torch.randn(M, N, generator=g).bfloat16()
  A = torch.zeros(8, K).bfloat16()   # unused by the dA path
  dA = torch.full((M, K), float('nan'), device='cuda', dtype=torch.bfloat16)
  Bd, dCd, Ad = B.cuda(), dC.cuda(), A.cuda()
  ws = _ws()
  rc = lib.spa3d_op_linear_bwd(Ad.data_ptr(), Bd.data_ptr(), dCd.data_ptr(), dA.data_ptr(), None, None, M, N, K, BF16, 10, ws.data_ptr(), ws.numel(), _s())
  assert rc == 0
  ref = dC.double() @ B.double().T
  assert not torch.isnan(dA.float()).any()
  assert rel_err(dA.float(), ref) < 4e-3
  assert max_abs(dA.float(), ref) < 0.05 * float(ref.abs().max())


@pytest.mark.parametrize('bwd_mode', ['1', '2', '3'])
@pytest.mark.parametrize('nseq,S,H,masked', [(5, 25, 8, True), (3, 129, 8, False), (4, 151, 8, True), (2, 128, 8, False), (3, 40, 2, True),
                                             (17, 151, 8, True), (3, 301, 8, True), (2, 200, 8, False), (2, 320, 4, True), (9, 193, 2, True), (3, 176, 8, True), (2, 160, 4, False)])
def test_attention_fused_fwd_bwd(lib, nseq, S, H, masked, bwd_mode):
  """LDS-resident fused forward / backward (impl=2) vs the fp64 oracle; packed q|k|v rows as the QKV projection writes them.
  bwd_mode: 1 = four resident images + concurrent roles (S <= 160; the product dispatch, impl 2), 2 / 3 = split-pass with 4 / 8 waves (impl 3 / 4);
  S > 160 (BASELINE cfg#5: S = 301) always takes the 8-wave split-pass kernel.  nseq = 17 / 9 exercise the XCD-major problem map's identity tail."""
  if S > 160 and bwd_mode != '3':
    pytest.skip('S > 160 has one backward structure')
  bwd_impl = {'1': 2, '2': 3, '3': 4}[bwd_mode]
  Dh, E = 96, H * 96
  g = torch.Generator().manual_seed(21)
  qkv = torch.randn(nseq, S, 3 * E, generator=g).bfloat16()
  sq = 1 + 0.2 * torch.randn(Dh, generator=g)
  sk = 1 + 0.2 * torch.randn(Dh, generator=g)
  km = None
  if masked:
    km = (torch.rand(nseq, S, generator=g) < 0.8).float()
    km[:, 0] = 1.0
    km[0, 1:] = 0.0
  qkvd = qkv.cuda()
  o = torch.full((nseq, S, E), float('nan'), device='cuda', dtype=torch.bfloat16)
  lse = torch.zeros(nseq, H, S, 2, device='cuda')
  ws = _ws(64 << 20)
  sqd, skd = sq.cuda(), sk.cuda()
  kmd = km.cuda() if masked else None
  rc = lib.spa3d_op_attention(qkvd[..., :E].data_ptr(), qkvd[..., E:2 * E].data_ptr(), qkvd[..., 2 * E:].data_ptr(), 3 * E, 3 * E, 3 * E,
                              sqd.data_ptr(), skd.data_ptr(), kmd.data_ptr() if masked else None, nseq, S, S, H, Dh, o.data_ptr(),
                              lse.data_ptr(), BF16, 2, ws.data_ptr(), ws.numel(), _s())
  assert rc == 0
  qr = qkv[..., :E].double().contiguous().requires_grad_(True)
  kr = qkv[..., E:2 * E].double().contiguous().requires_grad_(True)
  vr = qkv[..., 2 * E:].double().contiguous().requires_grad_(True)
  sqr, skr = sq.double().requires_grad_(True), sk.double().requires_grad_(True)
  ref = _attn_ref(qr, kr, vr, sqr, skr, km, H, Dh)
  assert not torch.isnan(o.float()).any()
  e = rel_err(o.float(), ref.detach())
  print('fused attention fwd rel err', e)
  assert e < 2e-2
  # fused backward (needs the forward's o and lse)
  d_o = torch.randn(nseq, S, E, generator=g).bfloat16()
  ref.backward(d_o.double())
  dod = d_o.cuda()
  dqkv = torch.full((nseq, S, 3 * E), float('nan'), device='cuda', dtype=torch.bfloat16)
  dsq = torch.zeros(Dh, device='cuda')
  dsk = torch.zeros(Dh, device='cuda')
  rc = lib.spa3d_op_attention_bwd(qkvd[..., :E].data_ptr(), qkvd[..., E:2 * E].data_ptr(), qkvd[..., 2 * E:].data_ptr(), 3 * E, 3 * E,
                                  3 * E, sqd.data_ptr(), skd.data_ptr(), kmd.data_ptr() if masked else None, nseq, S, S, H, Dh,
                                  o.data_ptr(), lse.data_ptr(), dod.data_ptr(), dqkv[..., :E].data_ptr(), dqkv[..., E:2 * E].data_ptr(),
                                  dqkv[..., 2 * E:].data_ptr(), dsq.data_ptr(), dsk.data_ptr(), BF16, bwd_impl, ws.data_ptr(), ws.numel(), _s())
  assert rc == 0
  assert not torch.isnan(dqkv.float()).any()
  errs = [rel_err(dqkv[..., :E].float(), qr.grad), rel_err(dqkv[..., E:2 * E].float(), kr.grad), rel_err(dqkv[..., 2 * E:].float(), vr.grad),
          rel_err(dsq, sqr.grad), rel_err(dsk, skr.grad)]
  print('fused attention bwd rel errs dq dk dv dsq dsk', errs)
  assert max(errs) < 3e-2


@pytest.mark.parametrize('dtype', [BF16, F16])
@pytest.mark.parametrize('nseq,Sq,Sk,H,masked', [(3, 128, 2048, 8, False), (2, 128, 300, 8, True), (2, 100, 129, 8, True), (5, 16, 128, 2, False),
                                                 (1, 128, 8192, 8, False), (2, 37, 1000, 4, True), (3, 128, 8, 8, False), (2, 128, 13, 8, True), (2, 128, 100, 8, False)])
def test_attention_fused_cross(lib, dtype, nseq, Sq, Sk, H, masked):
  """Fused cross attention (impl=2, Sq != Sk: the 128 latents against the N track tokens, track_autoencoder_3d.py:95-100,201) vs the fp64
  oracle: keys in chunks of 128 with a split-softmax merge, dq^ partials per chunk summed before the RMSNorm backward.  Sk = 300 / 129 /
  1000: ragged last chunk (129: a one-key chunk); Sq = 100 / 37 / 16: partial query tiles; a fully masked first sequence and chunks whose
  keys are all masked exercise the finfo.min semantics across the merge."""
  Dh, E = 96, H * 96
  dt = _dt(dtype)
  g = torch.Generator().manual_seed(33)
  q = torch.randn(nseq, Sq, E, generator=g).to(dt)
  kv = torch.randn(nseq, Sk, 2 * E, generator=g).to(dt)
  sq = 1 + 0.2 * torch.randn(Dh, generator=g)
  sk = 1 + 0.2 * torch.randn(Dh, generator=g)
  km = None
  if masked:
    km = (torch.rand(nseq, Sk, generator=g) < 0.7).float()
    km[0, :] = 0.0            # every key masked: uniform attention over ALL keys
    if nseq > 1:
      km[1, :min(Sk // 2, 256)] = 0.0  # whole chunks masked, others not
      km[1, -1] = 1.0
  qd, kvd, sqd, skd = q.cuda(), kv.cuda(), sq.cuda(), sk.cuda()
  kmd = km.cuda() if masked else None
  o = torch.full((nseq, Sq, E), float('nan'), device='cuda', dtype=dt)
  lse = torch.zeros(nseq, H, Sq, 2, device='cuda')
  ws = _ws(512 << 20)
  rc = lib.spa3d_op_attention(qd.data_ptr(), kvd[..., :E].data_ptr(), kvd[..., E:].data_ptr(), E, 2 * E, 2 * E, sqd.data_ptr(), skd.data_ptr(),
                              kmd.data_ptr() if masked else None, nseq, Sq, Sk, H, Dh, o.data_ptr(), lse.data_ptr(), dtype, 2, ws.data_ptr(),
                              ws.numel(), _s())
  assert rc == 0
  qr = q.double().requires_grad_(True)
  kr = kv[..., :E].double().contiguous().requires_grad_(True)
  vr = kv[..., E:].double().contiguous().requires_grad_(True)
  sqr, skr = sq.double().requires_grad_(True), sk.double().requires_grad_(True)
  ref = _attn_ref(qr, kr, vr, sqr, skr, km, H, Dh)
  assert not torch.isnan(o.float()).any()
  e = rel_err(o.float(), ref.detach())
  print('fused cross attention fwd rel err', e)
  assert e < (2e-2 if dtype == BF16 else 4e-3)
  d_o = torch.randn(nseq, Sq, E, generator=g).to(dt)
  ref.backward(d_o.double())
  dod = d_o.cuda()
  dq = torch.full((nseq, Sq, E), float('nan'), device='cuda', dtype=dt)
  dkv = torch.full((nseq, Sk, 2 * E), float('nan'), device='cuda', dtype=dt)
  dsq = torch.zeros(Dh, device='cuda')
  dsk = torch.zeros(Dh, device='cuda')
  rc = lib.spa3d_op_attention_bwd(qd.data_ptr(), kvd[..., :E].data_ptr(), kvd[..., E:].data_ptr(), E, 2 * E, 2 * E, sqd.data_ptr(), skd.data_ptr(),
                                  kmd.data_ptr() if masked else None, nseq, Sq, Sk, H, Dh, o.data_ptr(), lse.data_ptr(), dod.data_ptr(),
                                  dq.data_ptr(), dkv[..., :E].data_ptr(), dkv[..., E:].data_ptr(), dsq.data_ptr(), dsk.data_ptr(), dtype, 2,
                                  ws.data_ptr(), ws.numel(), _s())
  assert rc == 0
  assert not torch.isnan(dq.float()).any() and not torch.isnan(dkv.float()).any()
  errs = [rel_err(dq.float(), qr.grad), rel_err(dkv[..., :E].float(), kr.grad), rel_err(dkv[..., E:].float(), vr.grad), rel_err(dsq, sqr.grad),
          rel_err(dsk, skr.grad)]
  print('fused cross attention bwd rel errs dq dk dv dsq dsk', errs)
  assert max(errs) < (3e-2 if dtype == BF16 else 6e-3)


@pytest.mark.parametrize('M,N,K,act,res,bias', [(1000, 384, 256, 0, False, True), (4133, 2304, 384, 0, False, False), (2050, 384, 1536, 0, True, True)])
def test_linear_tiled_nt_double_buffered(lib, M, N, K, act, res, bias):
  """the 2-buffer kernel with the LDS-staged epilogue also on the short-K shapes the single-buffer kernel normally takes (impl 5)"""
  test_linear_tiled_nt(lib, M, N, K, act, res, bias, impl=5)


@pytest.mark.parametrize('M,N,K,act,res,bias', [(4133, 2304, 384, 0, False, False), (1000, 1536, 384, 1, False, True),
                                                (2050, 768, 256, 0, True, True), (700, 1280, 1536, 0, False, True), (300, 256, 64, 0, False, False),
                                                (513, 512, 128, 0, False, True), (256, 256, 192, 0, True, False), (9000, 1280, 768, 0, True, True),
                                                (4133, 384, 768, 0, True, True), (900, 1152, 256, 1, False, True), (130, 384, 64, 0, False, False),
                                                (2050, 384, 1536, 0, True, True), (777, 384, 128, 0, False, True)])
def test_linear_tiled_nt_8phase(lib, M, N, K, act, res, bias):
  """the 8-phase kernels (256x256 when 256 | N, 128x384 when 384 | N; counted vmcnt, staggered wave rows), forced on for any M (impl 4: the 128x384
  shapes on the NON-persistent kernel; the persistent one is the default since round 3);
  K = 64 / 128 / 192 exercise the prologue and tail paths of the schedule (1, 2, 3 K-tiles)"""
  test_linear_tiled_nt(lib, M, N, K, act, res, bias, impl=4)


@pytest.mark.parametrize('M,N,K,act,res,bias', [(70001, 512, 256, 0, True, True), (70008, 512, 256, 0, True, True), (140000, 256, 128, 1, False, True), (66000, 768, 384, 0, False, False),
                                                (4133, 2304, 384, 0, True, False), (256 * 300, 512, 192, 0, True, True), (9000, 1280, 768, 1, False, True),
                                                (70008, 384, 256, 0, True, True), (66000, 1152, 384, 1, False, True), (140000, 384, 128, 0, False, False),
                                                (65544, 384, 768, 0, True, False), (128 * 700, 384, 1536, 0, False, True)])
def test_linear_tiled_nt_8phase_persistent(lib, M, N, K, act, res, bias):
  """persistent 8-phase kernels (256x256 and 128x384; impl 3): more tiles than CUs (cross-tile prefetch + counted store wait), ragged
  last M tile (drain path), exact multiples, residual / GELU epilogues"""
  test_linear_tiled_nt(lib, M, N, K, act, res, bias, impl=3)


@pytest.mark.parametrize('dtype,impl', [(F32, 1), (BF16, 1), (BF16, 2)])
def test_attention_fully_masked_sequence(lib, dtype, impl):
  """where(mask, logit, finfo.min): a sequence with NO visible key attends uniformly (forward) and passes no gradient to
  q/k through its logits (backward) -- generic and fused kernels."""
  nseq, S, H, Dh = 3, 40, 8, 96
  E = H * Dh
  g = torch.Generator().manual_seed(31)
  dt = _dt(dtype)
  qkv = torch.randn(nseq, S, 3 * E, generator=g).to(dt)
  sq = 1 + 0.2 * torch.randn(Dh, generator=g); sk = 1 + 0.2 * torch.randn(Dh, generator=g)
  km = (torch.rand(nseq, S, generator=g) < 0.7).float(); km[:, 0] = 1.0
  km[1] = 0.0  # fully masked sequence
  d_o = torch.randn(nseq, S, E, generator=g).to(dt)
  qkvd, sqd, skd, kmd, dod = qkv.cuda(), sq.cuda(), sk.cuda(), km.cuda(), d_o.cuda()
  o = torch.empty(nseq, S, E, device='cuda', dtype=dt); lse = torch.zeros(nseq, H, S, 2, device='cuda')
  ws = _ws(256 << 20)
  assert lib.spa3d_op_attention(qkvd[..., :E].data_ptr(), qkvd[..., E:2 * E].data_ptr(), qkvd[..., 2 * E:].data_ptr(), 3 * E, 3 * E, 3 * E,
                                sqd.data_ptr(), skd.data_ptr(), kmd.data_ptr(), nseq, S, S, H, Dh, o.data_ptr(), lse.data_ptr(), dtype, impl,
                                ws.data_ptr(), ws.numel(), _s()) == 0
  qr = qkv[..., :E].double().contiguous().requires_grad_(True)
  kr = qkv[..., E:2 * E].double().contiguous().requires_grad_(True)
  vr = qkv[..., 2 * E:].double().contiguous().requires_grad_(True)
  ref = _attn_ref(qr, kr, vr, sq.double(), sk.double(), km, H, Dh)
  ref.backward(d_o.double())
  assert float(qr.grad[1].abs().max()) == 0.0 and float(kr.grad[1].abs().max()) == 0.0  # the oracle's own statement of the rule
  dqkv = torch.full((nseq, S, 3 * E), float('nan'), device='cuda', dtype=dt)
  dsq = torch.zeros(Dh, device='cuda'); dsk = torch.zeros(Dh, device='cuda')
  assert lib.spa3d_op_attention_bwd(qkvd[..., :E].data_ptr(), qkvd[..., E:2 * E].data_ptr(), qkvd[..., 2 * E:].data_ptr(), 3 * E, 3 * E, 3 * E,
                                    sqd.data_ptr(), skd.data_ptr(), kmd.data_ptr(), nseq, S, S, H, Dh, o.data_ptr(), lse.data_ptr(),
                                    dod.data_ptr(), dqkv[..., :E].data_ptr(), dqkv[..., E:2 * E].data_ptr(), dqkv[..., 2 * E:].data_ptr(),
                                    dsq.data_ptr(), dsk.data_ptr(), dtype, impl, ws.data_ptr(), ws.numel(), _s()) == 0
  tol = 1e-4 if dtype == F32 else 3e-2
  assert rel_err(o.float(), ref.detach()) < tol
  assert float(dqkv[1, :, :2 * E].float().abs().max()) == 0.0
  assert rel_err(dqkv[..., :E].float(), qr.grad) < tol and rel_err(dqkv[..., E:2 * E].float(), kr.grad) < tol
  assert rel_err(dqkv[..., 2 * E:].float(), vr.grad) < tol
