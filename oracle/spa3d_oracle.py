"""CPU oracle for the 3DSPA TrackAutoEncoder3D hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (plain PyTorch ops, any float dtype) of the reference
graph in /root/reference/{attention.py, track_autoencoder.py, track_autoencoder_3d.py,
train.py}.  It exists only so that tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg can check / time against it.  The product path
(`3dspa_code_amd`) never imports it.

PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors, its
dependencies (jax/flax/optax) are not installed here, and its 3D forward pass is not
executable as written.  This oracle therefore follows the reference
source line by line with the repairs R2-R6 listed in SURVEY.md section 0.2, is
cross-checked against an independent NumPy restatement (oracle/np_blocks.py), by
closed-form known-answer tests and by finite differences (tests/test_oracle_*.py).

Third-party semantics restated from documented Flax/JAX/Optax behaviour:
  nn.Dense            y = x @ kernel[in,out] + bias
  nn.DenseGeneral     kernel [in,H,Dh] (features=(H,Dh)); kernel [H,Dh,D]+bias (axis=(-2,-1))
  nn.LayerNorm        eps=1e-6, var = max(0, E[x^2]-E[x]^2), no bias
  nn.RMSNorm          eps=1e-6, y = x * rsqrt(mean(x^2)+eps) * scale
  nn.gelu             tanh approximation
  nn.dot_product_attention   q/sqrt(d); where(mask, logits, finfo.min); softmax; PV
  optax.sigmoid_binary_cross_entropy, clip_by_global_norm, adamw, schedules
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Dict, Optional

import numpy as np
import torch

Params = Dict[str, Any]

# --------------------------------------------------------------------------------------
# hyper-parameters (track_autoencoder_3d.py:53-67)
# --------------------------------------------------------------------------------------


@dataclass
class Config:
  num_output_frames: int = 150
  num_latent_tokens: int = 128
  latent_token_dim: int = 96
  num_frequencies: int = 32
  track_scale_factor: float = 1.0
  time_scale_factor: float = 150.0
  track_token_dim: int = 384
  encoder_latent_dim: int = 512
  decoder_num_channels: int = 1280
  dino_feature_dim: int = 768
  depth_feature_dim: int = 256
  use_dino: bool = True
  use_depth: bool = True
  decoder_scan_chunk_size: Optional[int] = None
  # transformer sizes (track_autoencoder_3d.py:89-112)
  num_heads: int = 8
  qkv_size: int = 768
  enc_mlp: int = 1536
  enc_layers: int = 3
  t2l_mlp: int = 2048
  t2l_layers: int = 4
  dec_mlp: int = 2048
  dec_layers: int = 4
  ro_mlp: int = 1536
  ro_layers: int = 4


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------


def sin_scales(num_frequencies: int = 32) -> np.ndarray:
  """track_autoencoder.py:28 -- python doubles 2**(i/3) cast to float32 by jnp.asarray."""
  return np.asarray([2 ** (i / 3) for i in range(num_frequencies)], dtype=np.float32)


def sinusoidal_embedding(inputs: torch.Tensor, num_frequencies: int = 32) -> torch.Tensor:
  """SinusoidalEmbedding.__call__ (track_autoencoder.py:23-38).

  The reference runs in float32: v = fl32(x * s); [v, fl32(v + fl32(pi/2))] -> sin.
  The argument arithmetic is ALWAYS done in float32 here (phase errors are amplified by
  scales up to 2^(31/3) = 1290); only the sin itself is evaluated in the working dtype.
  Never calls cos (track_autoencoder.py:36).
  """
  wd = inputs.dtype
  x32 = inputs.to(torch.float32)
  scales = torch.from_numpy(sin_scales(num_frequencies))
  v = x32[..., None] * scales  # einsum("...,b->...b")        ta:30
  half_pi = torch.tensor(0.5 * math.pi, dtype=torch.float32)  # 0.5*jnp.pi -> weak f32
  arg = torch.cat([v, v + half_pi], dim=-1)  # ta:36
  out = torch.sin(arg.to(wd))
  return out.reshape(*inputs.shape[:-1], inputs.shape[-1] * 2 * num_frequencies)  # ta:37


def layer_norm(x: torch.Tensor, scale: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
  """flax nn.LayerNorm(use_bias=False): fast variance, clamped at 0."""
  mu = x.mean(-1, keepdim=True)
  var = torch.clamp((x * x).mean(-1, keepdim=True) - mu * mu, min=0.0)
  return (x - mu) * torch.rsqrt(var + eps) * scale


def rms_norm(x: torch.Tensor, scale: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
  """flax nn.RMSNorm over the last axis."""
  ms = (x * x).mean(-1, keepdim=True)
  return x * torch.rsqrt(ms + eps) * scale


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
  """flax nn.gelu default approximate=True."""
  c = math.sqrt(2.0 / math.pi)
  return 0.5 * x * (1.0 + torch.tanh(c * (x + 0.044715 * x * x * x)))


def dense(p: Params, x: torch.Tensor) -> torch.Tensor:
  return x @ p['kernel'] + p['bias']


def dot_product_attention(q, k, v, mask=None):
  """flax.linen.dot_product_attention: q,k,v [..., S, H, D]; mask [..., H|1, Sq|1, Sk]."""
  depth = q.shape[-1]
  q = q / math.sqrt(depth)
  logits = torch.einsum('...qhd,...khd->...hqk', q, k)
  if mask is not None:
    big_neg = torch.finfo(logits.dtype).min
    logits = torch.where(mask != 0, logits, torch.full_like(logits, big_neg))
  w = torch.softmax(logits, dim=-1)
  return torch.einsum('...hqk,...khd->...qhd', w, v)


def mhdp_attention(p: Params, inputs_q, inputs_kv, mask=None):
  """ImprovedMHDPAttention.__call__ (attention.py:125-185)."""
  query = torch.einsum('...d,dhk->...hk', inputs_q, p['dense_query']['kernel'])  # :154
  key = torch.einsum('...d,dhk->...hk', inputs_kv, p['dense_key']['kernel'])  # :159
  query = rms_norm(query, p['norm_query']['scale'])  # :166
  key = rms_norm(key, p['norm_key']['scale'])  # :167
  value = torch.einsum('...d,dhk->...hk', inputs_kv, p['dense_value']['kernel'])  # :169
  x = dot_product_attention(query, key, value, mask)  # :175
  out = torch.einsum('...hk,hkd->...d', x, p['dense_out']['kernel']) + p['dense_out']['bias']
  return out  # :178-185


def transformer_block(p: Params, queries, inputs_kv=None, qq_mask=None, qk_mask=None):
  """ImprovedTransformerBlock.__call__ (attention.py:67-108)."""
  normed_queries = layer_norm(queries, p['norm_q']['scale'])  # :76-78
  attn_out = queries
  attn_out = attn_out + mhdp_attention(p['self_att'], normed_queries, normed_queries, qq_mask)
  if inputs_kv is not None:  # :92-100 -- K/V from UN-normalised inputs_kv
    attn_out = attn_out + mhdp_attention(p['cross_att'], normed_queries, inputs_kv, qk_mask)
  normed_attn_out = layer_norm(attn_out, p['norm_attn']['scale'])  # :103-105
  h = gelu_tanh(dense(p['MLP_in'], normed_attn_out))  # :106
  return attn_out + dense(p['MLP_out'], h)  # :107-108


def transformer(p: Params, queries, inputs_kv=None, qq_mask=None, qk_mask=None):
  """ImprovedTransformer.__call__ (attention.py:23-53)."""
  num_layers = sum(1 for k in p if k.startswith('layer_'))
  for i in range(num_layers):
    # mask rank fix-up (attention.py:32-35): add the head axis once
    if qk_mask is not None and qk_mask.dim() == inputs_kv.dim():
      qk_mask = qk_mask[..., None, :, :]
    if qq_mask is not None and qq_mask.dim() == queries.dim():
      qq_mask = qq_mask[..., None, :, :]
    queries = transformer_block(p[f'layer_{i}'], queries, inputs_kv, qq_mask, qk_mask)
  return layer_norm(queries, p['norm_encoder']['scale'])  # :49-51


# --------------------------------------------------------------------------------------
# results / context containers (track_autoencoder.py:72-114)
# --------------------------------------------------------------------------------------


@dataclass
class Results:
  tracks: torch.Tensor  # [B,Q,T,3]
  visible_logits: torch.Tensor  # [B,Q,T,1]
  certain_logits: torch.Tensor  # [B,Q,T,1]

  @property
  def visible(self):  # ta:93-95
    return (self.visible_logits > 0).to(torch.float32)

  @property
  def certain(self):  # ta:97-99
    return (self.certain_logits > 0).to(torch.float32)

  @property
  def visible_and_certain(self):  # ta:101-105
    return ((torch.sigmoid(self.visible_logits) * torch.sigmoid(self.certain_logits)) > 0.5).to(
        torch.float32)


@dataclass
class DecoderContext:
  decoder_query: torch.Tensor  # [B,Q,192]
  query_frame: torch.Tensor  # int [B,Q]
  boundary_frame: torch.Tensor


# --------------------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------------------


class TrackAutoEncoder3D:
  """Restatement of track_autoencoder_3d.TrackAutoEncoder3D (3d:43-357) with repairs R2-R5."""

  def __init__(self, cfg: Config | None = None, **kw):
    self.cfg = cfg or Config(**kw)

  # ---- 3d:117-121
  def encode_point_identities(self, query_points):
    return sinusoidal_embedding(query_points / self.cfg.track_scale_factor, self.cfg.num_frequencies)

  # ---- 3d:123-149 (R4/R5: dino/depth project to track_token_dim)
  def embed_track_pos_visible(self, p, tracks, visible, dino_features=None, depth_features=None):
    T = tracks.shape[-2]
    # jnp.arange(T)/T : int32 / python int -> float32 true division               3d:126
    fr_id = (torch.arange(T, dtype=torch.float32) / T).to(tracks.dtype)
    fr_id = fr_id[None, None, :, None].expand(visible.shape)
    tracks_with_time = torch.cat([tracks, fr_id], dim=-1)  # 3d:131
    emb = sinusoidal_embedding(tracks_with_time / self.cfg.track_scale_factor, self.cfg.num_frequencies)
    out = dense(p['track_token_projection'], emb)  # 3d:137
    if self.cfg.use_dino and dino_features is not None:
      out = out + dense(p['dino_projection'], dino_features)  # 3d:140-142
    if self.cfg.use_depth and depth_features is not None:
      out = out + dense(p['depth_projection'], depth_features)  # 3d:145-147
    return out

  # ---- key mask (R2/R3): km[...,0]=1 ; km[...,1+t] = visible[t] & (t < boundary_frame)
  @staticmethod
  def key_mask(visible, restart):
    T = visible.shape[2]
    time = torch.arange(T)
    partition = time[None, None, :] < restart[:, None, None]  # 3d:167-168
    vis = visible[..., 0] != 0  # 3d:169
    km = partition & vis
    ones = torch.ones_like(km[..., :1])  # "Readout token is always visible" 3d:176
    return torch.cat([ones, km], dim=-1)  # [B,N,T+1]

  # ---- 3d:151-188
  def encode_tracks(self, p, tracks, visible, restart, dino_features=None, depth_features=None):
    emb = self.embed_track_pos_visible(p, tracks, visible, dino_features, depth_features)
    B, N = emb.shape[:2]
    readout = p['input_readout_token']['state_init'].expand(B, N, 1, emb.shape[-1])  # 3d:161-162
    track_tokens = torch.cat([readout, emb], dim=-2)  # 3d:163-165
    km = self.key_mask(visible, restart)  # [B,N,T+1]
    qq_mask = km[:, :, None, :].expand(B, N, km.shape[-1], km.shape[-1])  # same for every query row
    track_tokens = transformer(p['input_track_transformer'], track_tokens, qq_mask=qq_mask)
    return track_tokens[..., 0, :]  # 3d:187-188

  # ---- 3d:190-204
  def encode(self, p, inputs):
    support_track_tokens = self.encode_tracks(
        p, inputs['support_tracks'], inputs['support_tracks_visible'], inputs['boundary_frame'],
        inputs.get('dino_features'), inputs.get('depth_features'))
    B = inputs['support_tracks'].shape[0]
    latents = p['initializer']['state_init'].expand(B, *p['initializer']['state_init'].shape)
    latents = transformer(p['tracks_to_latents'], latents, support_track_tokens)  # 3d:201
    return dense(p['compressor'], latents)  # 3d:203

  # ---- 3d:206-233
  def get_decoder_context(self, inputs):
    if 'query_points' in inputs:
      decoder_query = inputs['query_points'][..., 1:]
      query_frame = torch.round(inputs['query_points'][..., 0]).to(torch.int32)  # half-even, as jnp.round
    else:
      grid = torch.arange(32, dtype=torch.float32) / 32.0 + 1.0 / 64.0
      qx, qy = torch.meshgrid(grid, grid, indexing='xy')  # jnp.meshgrid default 'xy'
      qz = torch.zeros_like(qx)
      decoder_query = torch.stack([qx, qy, qz], dim=-1).reshape(-1, 3)
      lead = inputs['support_tracks'].shape[:-3]
      decoder_query = decoder_query.expand(*lead, *decoder_query.shape).to(inputs['support_tracks'].dtype)
      query_frame = torch.zeros(decoder_query.shape[:-1], dtype=torch.int32)
    return DecoderContext(self.encode_point_identities(decoder_query), query_frame, inputs['boundary_frame'])

  # ---- 3d:235-246, literal eye-einsum form (the window gather is checked against this in the KATs)
  def append_time_feat(self, latents, query_frame):
    C = latents.shape[-1]
    assert C == self.cfg.decoder_num_channels - 128  # 3d:237
    d = torch.arange(128)[:, None]
    c = torch.arange(C)[None, :]
    eye = (c == d + 5 * query_frame[..., None, None].to(torch.int64)).to(latents.dtype)  # jnp.eye(128,C,5*idx)
    to_append = torch.einsum('...nc,...dc->...nd', latents, eye)
    return torch.cat([latents, to_append], dim=-1)

  # ---- 3d:248-307
  def decode(self, p, latents, ctx: DecoderContext, discretize=True, noise=None):
    cfg = self.cfg
    latents = torch.clamp(latents, -1.0, 1.0)  # 3d:251
    if discretize:
      latents_disc = torch.round(latents * 128.0) / 128.0  # 3d:253
      if noise is None:
        raise ValueError('oracle decode(discretize=True) needs the uniform noise tensor explicitly '
                         '(jax.random.uniform(PRNGKey(0)) is not pinned, SURVEY App. C)')
      latents_disc = latents_disc + noise.to(latents.dtype) / 128.0 - 1.0 / 256.0  # 3d:254-258
      latents = latents - (latents - latents_disc).detach()  # 3d:260
    latents = dense(p['decompressor'], latents)  # 3d:262
    latents = transformer(p['decompress_attn'], latents)  # 3d:263
    tfeat = torch.floor(ctx.query_frame[..., None].to(latents.dtype) / cfg.time_scale_factor)  # `//` 3d:268-269
    queries = torch.cat([ctx.decoder_query, tfeat], dim=-1)  # 3d:265-272
    pce = dense(p['query_encoder'],
                sinusoidal_embedding(queries / cfg.track_scale_factor, cfg.num_frequencies))  # 3d:273-275
    Q = pce.shape[-2]
    latents = latents[:, None].expand(latents.shape[0], Q, *latents.shape[1:])  # tile 3d:276-280
    latents = self.append_time_feat(latents, ctx.query_frame)  # 3d:281
    latents = torch.cat([pce[..., None, :], latents], dim=2)  # 3d:282-284
    out = transformer(p['track_readout_attn'], latents)  # 3d:285
    out = dense(p['track_predictor'], out[..., 0, :])  # 3d:286-287
    T = cfg.num_output_frames
    tracks = torch.stack([out[..., :T], out[..., T:2 * T], out[..., 2 * T:3 * T]], dim=-1)  # 3d:291-298
    visible_logits = out[..., 3 * T:, None]  # 3d:299
    return Results(tracks, visible_logits, torch.zeros_like(visible_logits))  # 3d:301-307

  # ---- 3d:309-357 (scan-chunked decode is numerically identical; chunk loop restated)
  def __call__(self, p, inputs, discretize=True, noise=None):
    latents = self.encode(p, inputs)
    h = self.cfg.decoder_scan_chunk_size
    if h is None or 'query_points' not in inputs:
      return self.decode(p, latents, self.get_decoder_context(inputs), discretize, noise)
    outs = []
    Q = inputs['query_points'].shape[-2]
    for q0 in range(0, Q, h):
      sub = dict(inputs)
      sub['query_points'] = inputs['query_points'][..., q0:q0 + h, :]
      outs.append(self.decode(p, latents, self.get_decoder_context(sub), discretize, noise))
    return Results(torch.cat([o.tracks for o in outs], 1), torch.cat([o.visible_logits for o in outs], 1),
                   torch.cat([o.certain_logits for o in outs], 1))

  apply = __call__


# --------------------------------------------------------------------------------------
# the 2-D TRAJAN twin (track_autoencoder.py:117-390) -- runnable as written upstream, restated with the SAME blocks
# --------------------------------------------------------------------------------------


def config_2d(**kw) -> Config:
  """hyper-parameters of TrackAutoEncoder (ta:120-135, 149-172)"""
  base = dict(latent_token_dim=64, track_token_dim=256, encoder_latent_dim=512, decoder_num_channels=1024, use_dino=False,
              use_depth=False, num_heads=8, qkv_size=512, enc_mlp=1024, enc_layers=2, t2l_mlp=2048, t2l_layers=6, dec_mlp=2048,
              dec_layers=3, ro_mlp=1024, ro_layers=4)
  base.update(kw)
  return Config(**base)


class TrackAutoEncoder2D(TrackAutoEncoder3D):
  """track_autoencoder.TrackAutoEncoder: (x,y) tracks, key mask without a readout key, visible-mean pooling, certainty head."""

  def embed_track_pos_visible(self, p, tracks, visible, dino_features=None, depth_features=None):  # ta:183-203
    T = tracks.shape[-2]
    fr_id = (torch.arange(T, dtype=torch.float32) / T).to(tracks.dtype)
    fr_id = fr_id[None, None, :, None].expand(visible.shape)
    return sinusoidal_embedding(torch.cat([tracks, fr_id], dim=-1) / self.cfg.track_scale_factor, self.cfg.num_frequencies)

  def encode_tracks(self, p, tracks, visible, restart, dino_features=None, depth_features=None):  # ta:205-232
    tok = dense(p['track_token_projection'], self.embed_track_pos_visible(p, tracks, visible))
    B, N, T = tok.shape[:3]
    partition = torch.arange(T)[None, None, :] < restart[:, None, None]
    vis = visible[..., 0] != 0
    km = partition & vis  # [B,N,T]; "ones_like(visible[..., newaxis]) * visible[..., newaxis, :]" is a key-only mask
    tok = transformer(p['input_track_transformer'], tok, qq_mask=km[:, :, None, :].expand(B, N, T, T))
    v = vis.to(tok.dtype)[..., None]
    return (tok * v).sum(-2) / torch.clamp(v.sum(-2), min=1.0)

  def get_decoder_context(self, inputs):  # ta:248-273
    if 'query_points' in inputs:
      decoder_query = inputs['query_points'][..., 1:]
      query_frame = torch.round(inputs['query_points'][..., 0]).to(torch.int32)
    else:
      grid = torch.arange(32, dtype=torch.float32) / 32.0 + 1.0 / 64.0
      qx, qy = torch.meshgrid(grid, grid, indexing='xy')
      decoder_query = torch.stack([qx, qy], dim=-1).reshape(-1, 2)
      lead = inputs['support_tracks'].shape[:-3]
      decoder_query = decoder_query.expand(*lead, *decoder_query.shape).to(inputs['support_tracks'].dtype)
      query_frame = torch.zeros(decoder_query.shape[:-1], dtype=torch.int32)
    return DecoderContext(self.encode_point_identities(decoder_query), query_frame, inputs['boundary_frame'])

  def decode(self, p, latents, ctx, discretize=True, noise=None):  # ta:290-350
    r3 = super().decode(p, latents, ctx, discretize, noise)  # identical up to the head split; rebuild it from the raw head
    T = self.cfg.num_output_frames
    # super() stacked blocks 0,1,2 as coordinates and returned block 3 as visible logits; TRAJAN: blocks 0,1 coords, 2 visible, 3 certain
    tracks = r3.tracks[..., :2]
    visible_logits = r3.tracks[..., 2:3]
    certain_logits = r3.visible_logits
    return Results(tracks, visible_logits, certain_logits)


def init_params_2d(cfg: Config, seed=0, dtype=torch.float32, perturb=0.0) -> Params:
  p = init_params(cfg, seed=seed, dtype=dtype, with_dino=False, with_depth=False, perturb=perturb)
  gen = torch.Generator().manual_seed(seed + 1000)
  nf = cfg.num_frequencies
  d, dd = cfg.track_token_dim, cfg.decoder_num_channels
  del p['input_readout_token']  # declared in setup() but never called upstream -> no Flax parameter
  p['track_token_projection'] = _dense_p(gen, 3 * 2 * nf, d, dtype)
  p['query_encoder'] = _dense_p(gen, (2 * 2 * nf + 1) * 2 * nf, dd, dtype)
  if perturb > 0:
    for k in ('track_token_projection', 'query_encoder'):
      p[k]['bias'] += perturb * torch.randn(p[k]['bias'].shape, generator=gen, dtype=torch.float64).to(dtype)
  return p


def synthetic_batch_2d(B, N, Q, T, seed=1234, dtype=torch.float32):
  b = synthetic_batch(B, N, Q, T, seed=seed, dtype=dtype)
  b['support_tracks'] = b['support_tracks'][..., :2].contiguous()
  b['query_tracks'] = b['query_tracks'][..., :2].contiguous()
  b['query_points'] = b['query_points'][..., :3].contiguous()
  return b


compute_loss_2d = None  # assigned below: same arithmetic as compute_loss_3d (train.py:60-93)


# --------------------------------------------------------------------------------------
# loss (train.py:96-129)
# --------------------------------------------------------------------------------------


def sigmoid_binary_cross_entropy(logits, labels):
  """optax: -y*log_sigmoid(l) - (1-y)*log_sigmoid(-l)."""
  ls = torch.nn.functional.logsigmoid
  return -labels * ls(logits) - (1.0 - labels) * ls(-logits)


def compute_loss_3d(predictions: Results, targets, l1_weight=5000.0, bce_weight=1e-8, denom=None):
  """train.py:96-129.  `denom` (not in the reference): an externally supplied max(sum(visible),1) -- the batch-GLOBAL
  count when the batch is sharded over data-parallel ranks; None = this batch's own, as the reference computes it."""
  tt, tv = targets['query_tracks'], targets['query_tracks_visible']
  vm = tv.to(predictions.tracks.dtype)
  pos = (torch.abs(predictions.tracks - tt) * vm).sum(dim=(-2, -1))
  denom = torch.clamp(vm.sum(), min=1.0) if denom is None else denom
  pos = pos.sum() / denom
  vis = sigmoid_binary_cross_entropy(predictions.visible_logits, vm).sum() / denom
  return {'total_loss': l1_weight * pos + bce_weight * vis, 'position_loss': pos, 'visible_loss': vis}


compute_loss_2d = compute_loss_3d  # train.py:60-93 is the same code on [B,Q,T,2]


# --------------------------------------------------------------------------------------
# parameter tree with Flax default initialisers (values are seed-derived here;
# JAX's RNG stream cannot be reproduced and no released checkpoint is reachable)
# --------------------------------------------------------------------------------------


def _lecun(gen, fan_in, shape, dtype):
  # variance_scaling(1.0, 'fan_in', 'truncated_normal'): N(0, 1/fan_in) truncated at +-2 sigma,
  # std corrected by 0.87962566103423978
  std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
  t = torch.empty(shape, dtype=torch.float64)
  torch.nn.init.trunc_normal_(t, 0.0, 1.0, -2.0, 2.0, generator=gen)
  return (t * std).to(dtype)


def _dense_p(gen, fin, fout, dtype):
  return {'kernel': _lecun(gen, fin, (fin, fout), dtype), 'bias': torch.zeros(fout, dtype=dtype)}


def _attn_p(gen, dq, dkv, H, Dh, dtype):
  return {
      'dense_query': {'kernel': _lecun(gen, dq, (dq, H, Dh), dtype)},
      'dense_key': {'kernel': _lecun(gen, dkv, (dkv, H, Dh), dtype)},
      'dense_value': {'kernel': _lecun(gen, dkv, (dkv, H, Dh), dtype)},
      'norm_query': {'scale': torch.ones(Dh, dtype=dtype)},
      'norm_key': {'scale': torch.ones(Dh, dtype=dtype)},
      'dense_out': {'kernel': _lecun(gen, H * Dh, (H, Dh, dq), dtype), 'bias': torch.zeros(dq, dtype=dtype)},
  }


def _xf_p(gen, d, mlp, L, H, Dh, dtype, kv=None):
  p = {}
  for i in range(L):
    b = {'norm_q': {'scale': torch.ones(d, dtype=dtype)}, 'self_att': _attn_p(gen, d, d, H, Dh, dtype)}
    if kv is not None:
      b['cross_att'] = _attn_p(gen, d, kv, H, Dh, dtype)
    b['norm_attn'] = {'scale': torch.ones(d, dtype=dtype)}
    b['MLP_in'] = _dense_p(gen, d, mlp, dtype)
    b['MLP_out'] = _dense_p(gen, mlp, d, dtype)
    p[f'layer_{i}'] = b
  p['norm_encoder'] = {'scale': torch.ones(d, dtype=dtype)}
  return p


def init_params(cfg: Config, seed=0, dtype=torch.float32, with_dino=True, with_depth=True, depth_dim=1,
                perturb=0.0) -> Params:
  """model.init(...)['params'] equivalent.  `perturb`>0 adds N(0,perturb) to zero/one-initialised leaves
  (biases, norm scales) so tests exercise them."""
  gen = torch.Generator().manual_seed(seed)
  H, Dh = cfg.num_heads, cfg.qkv_size // cfg.num_heads
  d, dl, dd = cfg.track_token_dim, cfg.encoder_latent_dim, cfg.decoder_num_channels
  nin = 4 * 2 * cfg.num_frequencies
  p: Params = {}
  p['initializer'] = {'state_init': torch.randn(cfg.num_latent_tokens, dl, generator=gen, dtype=torch.float64).to(dtype)}
  p['input_readout_token'] = {'state_init': torch.randn(1, d, generator=gen, dtype=torch.float64).to(dtype)}
  p['track_token_projection'] = _dense_p(gen, nin, d, dtype)
  if cfg.use_dino and with_dino:
    p['dino_projection'] = _dense_p(gen, cfg.dino_feature_dim, d, dtype)  # R4
  if cfg.use_depth and with_depth:
    p['depth_projection'] = _dense_p(gen, depth_dim, d, dtype)  # R5
  p['input_track_transformer'] = _xf_p(gen, d, cfg.enc_mlp, cfg.enc_layers, H, Dh, dtype)
  p['tracks_to_latents'] = _xf_p(gen, dl, cfg.t2l_mlp, cfg.t2l_layers, H, Dh, dtype, kv=d)
  p['compressor'] = _dense_p(gen, dl, cfg.latent_token_dim, dtype)
  p['decompressor'] = _dense_p(gen, cfg.latent_token_dim, dd - 128, dtype)
  p['decompress_attn'] = _xf_p(gen, dd - 128, cfg.dec_mlp, cfg.dec_layers, H, Dh, dtype)
  p['track_readout_attn'] = _xf_p(gen, dd, cfg.ro_mlp, cfg.ro_layers, H, Dh, dtype)
  qin = (3 * 2 * cfg.num_frequencies + 1) * 2 * cfg.num_frequencies
  p['query_encoder'] = _dense_p(gen, qin, dd, dtype)
  p['track_predictor'] = _dense_p(gen, dd, 4 * cfg.num_output_frames, dtype)
  if perturb > 0:
    def rec(t):
      for k, v in t.items():
        if isinstance(v, dict):
          rec(v)
        elif k in ('bias', 'scale'):
          v += perturb * torch.randn(v.shape, generator=gen, dtype=torch.float64).to(dtype)
    rec(p)
  return p


def tree_map(fn, t):
  return {k: (tree_map(fn, v) if isinstance(v, dict) else fn(v)) for k, v in t.items()}


def tree_flatten(t, prefix=''):
  out = {}
  for k, v in t.items():
    key = f'{prefix}/{k}' if prefix else k
    if isinstance(v, dict):
      out.update(tree_flatten(v, key))
    else:
      out[key] = v
  return out


def tree_unflatten(flat):
  out: Params = {}
  for key, v in flat.items():
    parts = key.split('/')
    d = out
    for q in parts[:-1]:
      d = d.setdefault(q, {})
    d[parts[-1]] = v
  return out


# --------------------------------------------------------------------------------------
# synthetic batch (SURVEY 8(d) / BASELINE.md section 2)
# --------------------------------------------------------------------------------------


def synthetic_batch(B, N, Q, T, seed=1234, dino_dim=0, depth_dim=0, dtype=torch.float32):
  g = torch.Generator().manual_seed(seed)

  def walk(n):
    x0 = torch.rand(B, n, 1, 3, generator=g)
    steps = 0.01 * torch.randn(B, n, T, 3, generator=g)
    return torch.clamp(x0 + torch.cumsum(steps, dim=2), 0.0, 1.0)

  sup = walk(N)
  sup_vis = (torch.rand(B, N, T, 1, generator=g) < 0.9).float()
  qt = walk(Q)
  qt_vis = (torch.rand(B, Q, T, 1, generator=g) < 0.9).float()
  tq = torch.randint(0, T, (B, Q), generator=g)
  xyz = torch.gather(qt, 2, tq[:, :, None, None].expand(B, Q, 1, 3))[:, :, 0]
  qp = torch.cat([tq[..., None].float(), xyz], dim=-1)  # (t,x,y,z) data_loader.py:77-85
  batch = {
      'support_tracks': sup.to(dtype), 'support_tracks_visible': sup_vis.to(dtype),
      'query_points': qp.to(dtype), 'boundary_frame': torch.full((B,), T, dtype=torch.int32),
      'query_tracks': qt.to(dtype), 'query_tracks_visible': qt_vis.to(dtype),
  }
  if depth_dim:
    batch['depth_features'] = sup[..., 2:3].expand(B, N, T, depth_dim).contiguous().to(dtype)
  if dino_dim:
    batch['dino_features'] = torch.randn(B, N, T, dino_dim, generator=g).to(dtype)
  return batch


# --------------------------------------------------------------------------------------
# intended optimizer step (train.py:41-57, 161-165, 239-242; repair R6)
# --------------------------------------------------------------------------------------


def lr_schedule(step, base_lr, warmup_steps, total_steps):
  """optax.join_schedules([linear_schedule(0,base,warmup), cosine_decay_schedule(base,total-warmup)], [warmup])."""
  if step < warmup_steps:
    return base_lr * step / warmup_steps
  s = min(step - warmup_steps, total_steps - warmup_steps)
  return base_lr * 0.5 * (1.0 + math.cos(math.pi * s / (total_steps - warmup_steps)))


def adamw_step(params_flat, grads_flat, m, v, step, lr, clip=1.0, wd=0.01, b1=0.9, b2=0.999, eps=1e-8):
  """optax.chain(clip_by_global_norm(1.0), adamw(lr, weight_decay=0.01)); `step` = count BEFORE this update.
  In place on dict-of-tensors; returns the global grad norm."""
  gn = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads_flat.values()))
  sc = 1.0 if gn < clip else clip / gn
  t = step + 1
  for k, p in params_flat.items():
    g = grads_flat[k] * sc
    m[k].mul_(b1).add_(g, alpha=1 - b1)
    v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
    mh = m[k] / (1 - b1 ** t)
    vh = v[k] / (1 - b2 ** t)
    p.sub_(lr * (mh / (vh.sqrt() + eps) + wd * p))
  return gn


def loss_and_grads(model: TrackAutoEncoder3D, params: Params, batch, discretize=True, noise=None, denom=None):
  """jax.value_and_grad(loss_fn)(params) equivalent via torch autograd on the restated graph."""
  flat = tree_flatten(params)
  leaves = {k: v.detach().clone().requires_grad_(True) for k, v in flat.items()}
  preds = model(tree_unflatten(leaves), batch, discretize=discretize, noise=noise)
  ld = compute_loss_3d(preds, batch, denom=denom)
  ld['total_loss'].backward()
  grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
  return {k: v.detach() for k, v in ld.items()}, preds, grads
