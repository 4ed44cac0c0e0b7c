"""Second, independent CPU restatement (NumPy float64, einsum-literal) of the shared blocks.
TEST INFRASTRUCTURE ONLY -- cross-checks oracle/spa3d_oracle.py.

Follows /root/reference/attention.py:56-185 and track_autoencoder.py:18-38 directly; written
without looking at the torch oracle's helper decomposition (explicit loops over heads / rows
where that is the clearest literal reading).  PARITY UNPINNED (see spa3d_oracle.py header).
Also holds Threefry-2x32 and the JAX `uniform` bit recipe, KAT-checked against
the public Random123 vectors.
"""
from __future__ import annotations

import math

import numpy as np


def sin_embed(x, nf=32):
  scales = np.asarray([2 ** (i / 3) for i in range(nf)], dtype=np.float32)
  x = np.asarray(x, dtype=np.float32)
  v = (x[..., None] * scales).astype(np.float32)
  w = (v + np.float32(0.5 * np.pi)).astype(np.float32)
  out = np.sin(np.concatenate([v, w], axis=-1).astype(np.float64))
  return out.reshape(x.shape[:-1] + (x.shape[-1] * 2 * nf,))


def ln(x, s):
  mu = x.mean(-1, keepdims=True)
  var = np.maximum((x * x).mean(-1, keepdims=True) - mu * mu, 0.0)
  return (x - mu) / np.sqrt(var + 1e-6) * s


def rms(x, s):
  return x / np.sqrt((x * x).mean(-1, keepdims=True) + 1e-6) * s


def gelu(x):
  return 0.5 * x * (1 + np.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))


def attention(p, xq, xkv, keymask=None):
  """xq [S,d], xkv [Sk,dk]; keymask [Sk] of 0/1 applied to every query row."""
  Wq, Wk, Wv, Wo = (np.asarray(p[n]['kernel'], np.float64) for n in ('dense_query', 'dense_key', 'dense_value', 'dense_out'))
  H, Dh = Wq.shape[1], Wq.shape[2]
  out = np.zeros((xq.shape[0], Wo.shape[2]))
  for h in range(H):
    q = rms(xq @ Wq[:, h, :], np.asarray(p['norm_query']['scale'], np.float64))
    k = rms(xkv @ Wk[:, h, :], np.asarray(p['norm_key']['scale'], np.float64))
    v = xkv @ Wv[:, h, :]
    logits = (q / math.sqrt(Dh)) @ k.T
    if keymask is not None:
      logits = np.where(keymask[None, :] != 0, logits, np.finfo(np.float64).min)
    logits = logits - logits.max(-1, keepdims=True)
    w = np.exp(logits)
    w /= w.sum(-1, keepdims=True)
    out += (w @ v) @ Wo[h]
  return out + np.asarray(p['dense_out']['bias'], np.float64)


def block(p, x, kv=None, keymask=None):
  nq = ln(x, np.asarray(p['norm_q']['scale'], np.float64))
  a = x + attention(p['self_att'], nq, nq, keymask)
  if kv is not None:
    a = a + attention(p['cross_att'], nq, kv, None)
  na = ln(a, np.asarray(p['norm_attn']['scale'], np.float64))
  h = gelu(na @ np.asarray(p['MLP_in']['kernel'], np.float64) + np.asarray(p['MLP_in']['bias'], np.float64))
  return a + h @ np.asarray(p['MLP_out']['kernel'], np.float64) + np.asarray(p['MLP_out']['bias'], np.float64)


def transformer(p, x, kv=None, keymask=None):
  i = 0
  while f'layer_{i}' in p:
    x = block(p[f'layer_{i}'], x, kv, keymask)
    i += 1
  return ln(x, np.asarray(p['norm_encoder']['scale'], np.float64))


# ---------------------------------------------------------------------------------------------
# Threefry-2x32 (20 rounds) and jax.random.uniform bit recipe -- SURVEY Appendix C
# ---------------------------------------------------------------------------------------------

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def threefry2x32(key, ctr):
  """key=(k0,k1), ctr=(x0,x1) arrays/ints of uint32 -> (y0,y1)."""
  M = np.uint64(0xFFFFFFFF)
  k0, k1 = np.uint64(key[0]), np.uint64(key[1])
  ks = (k0, k1, (k0 ^ k1 ^ np.uint64(0x1BD11BDA)) & M)
  x0 = (np.asarray(ctr[0], dtype=np.uint64) + ks[0]) & M
  x1 = (np.asarray(ctr[1], dtype=np.uint64) + ks[1]) & M
  for i in range(5):
    for r in _ROT[i % 2]:
      x0 = (x0 + x1) & M
      x1 = ((x1 << np.uint64(r)) | (x1 >> np.uint64(32 - r))) & M
      x1 = x1 ^ x0
    x0 = (x0 + ks[(i + 1) % 3]) & M
    x1 = (x1 + ks[(i + 2) % 3] + np.uint64(i + 1)) & M
  return x0.astype(np.uint32), x1.astype(np.uint32)


def jax_uniform_legacy(shape, key=(0, 0)):
  """jax.random.uniform(PRNGKey(0), shape) under jax_threefry_partitionable=False.
  NOT pinned by any reference artefact; only the Threefry block function is KAT-checked."""
  n = int(np.prod(shape))
  npad = n + (n % 2)
  cnt = np.arange(npad, dtype=np.uint32)
  y0, y1 = threefry2x32(key, (cnt[:npad // 2], cnt[npad // 2:]))
  bits = np.concatenate([y0, y1])[:n]
  f = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
  return f.reshape(shape)
