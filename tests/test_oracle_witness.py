"""An outside witness for the oracle's third-party arithmetic (CPU).

`oracle/spa3d_oracle.py` restates the Flax / Optax operations the reference calls (`/root/reference/attention.py:76-78,103-107,166-175`,
`train.py:116,239-242`) by hand, and `tests/test_oracle_cross.py` compares it only with a second restatement by the same reader.  PyTorch ships its own,
independently written implementations of every one of those operations; this file checks the oracle's against them in fp64.  It does not pin parity with
the reference (nothing here executes JAX) -- it removes "two restatements, one reader" as a failure mode for the op-level arithmetic, and it writes down
the three places where the libraries legitimately differ:
  * Flax LayerNorm uses the fast variance E[x^2] - E[x]^2 clamped at 0, torch the two-pass form: identical in exact arithmetic, different rounding;
  * a fully masked attention row is UNIFORM in Flax (finfo.min fill, then softmax); torch fills with -inf and returns NaN (<= 2.4) or zeros (newer);
  * `clip_grad_norm_` divides by (norm + 1e-6), optax's `clip_by_global_norm` by the norm.
"""
import math

import torch
import torch.nn.functional as F

from util import O

torch.manual_seed(0)
F64 = torch.float64


def _rel(a, b):
  return float((a - b).norm() / (b.norm() + 1e-300))


def test_layer_norm_matches_torch_layer_norm():
  """attention.py:76-78,103-105: nn.LayerNorm(use_bias=False), Flax default epsilon 1e-6."""
  g = torch.Generator().manual_seed(1)
  for d in (32, 384, 1280):
    x = torch.randn(7, 5, d, generator=g, dtype=F64) * 3.0 + 0.7
    s = 1.0 + 0.2 * torch.randn(d, generator=g, dtype=F64)
    ref = F.layer_norm(x, (d,), weight=s, bias=None, eps=1e-6)
    assert _rel(O.layer_norm(x, s), ref) < 1e-13
  # where the fast variance matters: rows with |mean| >> spread.  In fp64 the two forms still agree to ~(mean/spread)^2 * 2^-53 ...
  x = 1000.0 + 1e-3 * torch.randn(4, 384, generator=g, dtype=F64)
  s = torch.ones(384, dtype=F64)
  ref = F.layer_norm(x, (384,), weight=s, bias=None, eps=1e-6)
  assert _rel(O.layer_norm(x, s), ref) < 1e-3  # (1e6)^2 * 1.1e-16 = 1.1e-4 relative error of the variance
  # ... in fp32 they do NOT (cancellation of E[x^2] - E[x]^2: variance 1e-6 against terms of 1e6 is below fp32 resolution): Flax's form -- the oracle's and the
  # HIP kernels' -- returns a clamped / noisy variance there.  Both are finite; the test documents the magnitude rather than hiding it.
  x32 = x.float()
  o32, t32 = O.layer_norm(x32, s.float()), F.layer_norm(x32, (384,), weight=s.float(), bias=None, eps=1e-6)
  assert torch.isfinite(o32).all() and torch.isfinite(t32).all()
  assert _rel(t32.double(), ref) < 0.2          # two-pass: close to the fp64 value (0.06 measured: the inputs themselves are rounded to fp32)
  assert float(o32.abs().max()) < 1.1e3         # fast variance clamped at 0: (x - mu) * rsqrt(eps) -- bounded, not the fp64 value
  # a constant row: variance exactly 0 in both forms -> output exactly 0
  c = torch.full((2, 96), 3.25, dtype=F64)
  assert float(O.layer_norm(c, torch.ones(96, dtype=F64)).abs().max()) == 0.0
  assert float(F.layer_norm(c, (96,), eps=1e-6).abs().max()) == 0.0


def test_rms_norm_matches_torch_rms_norm():
  """attention.py:166-167: nn.RMSNorm over the 96-wide head axis."""
  g = torch.Generator().manual_seed(2)
  x = torch.randn(3, 11, 8, 96, generator=g, dtype=F64) * 2.0
  s = 1.0 + 0.2 * torch.randn(96, generator=g, dtype=F64)
  assert _rel(O.rms_norm(x, s), F.rms_norm(x, (96,), weight=s, eps=1e-6)) < 1e-14


def test_gelu_matches_torch_tanh_gelu():
  """attention.py:106: nn.gelu (approximate=True)."""
  x = torch.linspace(-9.0, 9.0, 4001, dtype=F64)
  assert float((O.gelu_tanh(x) - F.gelu(x, approximate='tanh')).abs().max()) < 1e-14
  xs = torch.tensor([0.0, 1.0, -1.0, 3.0, -3.0], dtype=F64)
  assert float((O.gelu_tanh(xs) - F.gelu(xs, approximate='tanh')).abs().max()) < 1e-15


def test_dot_product_attention_matches_torch_sdpa():
  """attention.py:175: nn.dot_product_attention(query, key, value, mask=mask) -- [B, S, H, D] layout, 1/sqrt(D), boolean mask."""
  g = torch.Generator().manual_seed(3)
  B, Sq, Sk, H, D = 2, 9, 13, 4, 96
  q = torch.randn(B, Sq, H, D, generator=g, dtype=F64)
  k = torch.randn(B, Sk, H, D, generator=g, dtype=F64)
  v = torch.randn(B, Sk, H, D, generator=g, dtype=F64)
  tq, tk, tv = (t.permute(0, 2, 1, 3) for t in (q, k, v))  # torch: [B, H, S, D]
  ref = F.scaled_dot_product_attention(tq, tk, tv, scale=1.0 / math.sqrt(D)).permute(0, 2, 1, 3)
  assert _rel(O.dot_product_attention(q, k, v), ref) < 1e-13
  mask = torch.rand(B, 1, Sq, Sk, generator=g) < 0.7
  mask[..., 0] = True  # every row keeps one key
  ref = F.scaled_dot_product_attention(tq, tk, tv, attn_mask=mask, scale=1.0 / math.sqrt(D)).permute(0, 2, 1, 3)
  assert _rel(O.dot_product_attention(q, k, v, mask), ref) < 1e-13
  # the documented difference: a fully masked row
  mask[0, 0, 2, :] = False
  o = O.dot_product_attention(q, k, v, mask)
  t = F.scaled_dot_product_attention(tq, tk, tv, attn_mask=mask, scale=1.0 / math.sqrt(D)).permute(0, 2, 1, 3)
  assert torch.isnan(t[0, 2]).all() or float(t[0, 2].abs().max()) == 0.0     # torch: -inf fill -> NaN (<= 2.4) or zeros ("safe softmax", this image)
  assert _rel(o[0, 2], v[0].mean(dim=0)) < 1e-13                             # Flax (finfo.min fill): uniform over ALL keys
  keep = torch.ones(B, Sq, dtype=torch.bool); keep[0, 2] = False
  assert _rel(o[keep], t[keep]) < 1e-13                                      # every other row is untouched


def test_sigmoid_bce_matches_torch_bce_with_logits():
  """train.py:116: optax.sigmoid_binary_cross_entropy."""
  g = torch.Generator().manual_seed(4)
  logits = torch.cat([torch.randn(500, generator=g, dtype=F64) * 6.0, torch.tensor([0.0, 20.0, -20.0, 700.0, -700.0], dtype=F64)])
  labels = (torch.rand(505, generator=g) < 0.5).to(F64)
  ref = F.binary_cross_entropy_with_logits(logits, labels, reduction='none')
  assert float((O.sigmoid_binary_cross_entropy(logits, labels) - ref).abs().max()) < 1e-12
  assert torch.isfinite(O.sigmoid_binary_cross_entropy(logits, labels)).all()


def _adam_case(scale, steps=3):
  g = torch.Generator().manual_seed(5)
  shapes = {'a/kernel': (17, 5), 'a/bias': (5,), 'b/scale': (9,)}
  p0 = {k: torch.randn(*s, generator=g, dtype=F64) for k, s in shapes.items()}
  grads = [{k: scale * torch.randn(*s, generator=g, dtype=F64) for k, s in shapes.items()} for _ in range(steps)]
  lr = 3e-3
  # oracle
  po = {k: v.clone() for k, v in p0.items()}
  m = {k: torch.zeros_like(v) for k, v in p0.items()}; vv = {k: torch.zeros_like(v) for k, v in p0.items()}
  norms = [O.adamw_step(po, grads[t], m, vv, t, lr) for t in range(steps)]
  # torch: clip_grad_norm_(1.0) then AdamW(eps=1e-8, weight_decay=0.01) -- every leaf decayed, as optax.adamw without a mask (train.py:239-242)
  pt = {k: torch.nn.Parameter(v.clone()) for k, v in p0.items()}
  opt = torch.optim.AdamW(list(pt.values()), lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
  tn = []
  for t in range(steps):
    for k in pt:
      pt[k].grad = grads[t][k].clone()
    tn.append(float(torch.nn.utils.clip_grad_norm_(list(pt.values()), 1.0)))
    opt.step()
  return po, {k: v.detach() for k, v in pt.items()}, norms, tn


def test_adamw_matches_torch_adamw_without_clipping():
  """Gradient norm < 1: optax's clip is the identity and torch's coefficient is clamped to 1 -- three steps must agree to rounding."""
  po, pt, norms, tn = _adam_case(scale=0.02)
  assert max(norms) < 1.0
  for k in po:
    assert _rel(po[k], pt[k]) < 1e-13, k
  assert max(abs(a - b) for a, b in zip(norms, tn)) < 1e-12


def test_adamw_matches_torch_adamw_with_clipping_up_to_the_documented_1e6():
  """Gradient norm > 1: optax scales by 1/norm, torch by 1/(norm + 1e-6).  Adam's update is invariant to a common gradient scale up to eps = 1e-8 against
  sqrt(v_hat) ~ 1e-1, so the parameters still agree far below the 1e-6 of the clip coefficient."""
  po, pt, norms, tn = _adam_case(scale=5.0)
  assert min(norms) > 1.0
  for k in po:
    assert _rel(po[k], pt[k]) < 1e-9, k
  assert max(abs(a - b) for a, b in zip(norms, tn)) < 1e-10
