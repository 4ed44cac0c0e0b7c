"""Host-side mirror of the reference's Flax module protocol for the 3DSPA hot path.

Same names, argument meaning and error behaviour as the reference call sites
(train.py:137-146,194-199,221-233; evaluate_tapvid3d.py:70-77; inference.py:594-623):

    model  = TrackAutoEncoder3D(num_output_frames=..., use_dino=..., use_depth=...)
    params = model.init(rng, batch)['params']            # nested dict, Flax names/shapes
    preds  = model.apply({'params': params}, batch)      # TrackAutoEncoderResults
    losses = compute_loss_3d(preds, batch)               # {'total_loss','position_loss','visible_loss'}
    model.apply({'params': p}, batch, method=model.encode) etc.

Everything numeric happens in libspa3d_hip.so (hand-written HIP for gfx950) through the C-ABI in
include/spa3d.h; PyTorch only owns device memory and the stream.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math
from typing import Any, Dict, Optional

import torch

from . import _lib


# ------------------------------------------------------------------------------------------------
# result containers (track_autoencoder.py:72-114)
# ------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class TrackAutoEncoderResults:
  tracks: torch.Tensor  # [B,Q,T,3]
  visible_logits: torch.Tensor  # [B,Q,T,1]
  certain_logits: torch.Tensor  # [B,Q,T,1]  (identically zero, track_autoencoder_3d.py:301)

  @property
  def visible(self):  # ta:93-95
    return (self.visible_logits > 0).to(torch.float32)

  @property
  def certain(self):  # ta:97-99
    return (self.certain_logits > 0).to(torch.float32)

  @property
  def visible_and_certain(self):  # ta:101-105
    return ((torch.sigmoid(self.visible_logits) * torch.sigmoid(self.certain_logits)) > 0.5).to(torch.float32)


@dataclasses.dataclass
class TrackAutoEncoderDecoderContext:
  """get_decoder_context output.  The sinusoidal query identity itself is produced inside the fused
  decode kernels; the context carries what decode needs to rebuild it: the query points."""
  query_points: torch.Tensor  # [B,Q,4] (t,x,y,z)
  query_frame: torch.Tensor  # int32 [B,Q] = round(t)
  boundary_frame: Optional[torch.Tensor]

  @property
  def decoder_query(self):  # [B,Q,192] -- materialised on demand (ta:28-37)
    return sinusoidal_embedding(self.query_points[..., 1:].contiguous())


class ParamTree(dict):
  """Nested dict of parameter views; the root carries the flat fp32 buffer the views alias."""
  flat: Optional[torch.Tensor] = None


def sinusoidal_embedding(x: torch.Tensor, num_frequencies: int = 32) -> torch.Tensor:
  """SinusoidalEmbedding (track_autoencoder.py:18-38) on the GPU kernel."""
  lib = _lib.load()
  x = x.to(torch.float32).contiguous()
  rows = x.numel() // x.shape[-1]
  out = torch.empty(*x.shape[:-1], x.shape[-1] * 2 * num_frequencies, device=x.device, dtype=torch.float32)
  _lib.check(lib.spa3d_op_sin_embed(x.data_ptr(), rows, x.shape[-1], num_frequencies, out.data_ptr(), _lib.F32, _stream(x)),
             what='spa3d_op_sin_embed')
  return out


def _stream(t: torch.Tensor):
  return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _require_cuda(t: torch.Tensor, name: str):
  if not t.is_cuda:
    raise _lib.Spa3dError(f'{name} must live on the GPU: the 3DSPA hot path is HIP-only (no CPU fallback)')


# precision name -> (spa3d_config::precision, dtype of the DINO / depth feature planes handed to the library)
_PRECISIONS = {'fp32': (_lib.F32, torch.float32), 'bf16': (_lib.BF16, torch.bfloat16), 'fp16': (_lib.F16, torch.float16)}


# ------------------------------------------------------------------------------------------------
# the model
# ------------------------------------------------------------------------------------------------
class TrackAutoEncoder3D:
  """Drop-in for track_autoencoder_3d.TrackAutoEncoder3D (fields of 3d:53-67)."""

  def __init__(self, num_output_frames: int = 150, num_latent_tokens: int = 128, latent_token_dim: int = 96,
               num_frequencies: int = 32, track_scale_factor: float = 1.0, time_scale_factor: float = 150.0,
               track_token_dim: int = 384, encoder_latent_dim: int = 512, decoder_num_channels: int = 1280,
               dino_feature_dim: int = 768, depth_feature_dim: int = 256, use_dino: bool = True, use_depth: bool = True,
               decoder_scan_chunk_size: Optional[int] = None, precision: str = 'bf16',
               workspace_fraction: float = 0.80):
    if precision not in _PRECISIONS:
      raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}, got {precision!r}")
    self.num_output_frames = num_output_frames
    self.num_latent_tokens = num_latent_tokens
    self.latent_token_dim = latent_token_dim
    self.num_frequencies = num_frequencies
    self.track_scale_factor = track_scale_factor
    self.time_scale_factor = time_scale_factor
    self.track_token_dim = track_token_dim
    self.encoder_latent_dim = encoder_latent_dim
    self.decoder_num_channels = decoder_num_channels
    self.dino_feature_dim = dino_feature_dim
    self.depth_feature_dim = depth_feature_dim
    self.use_dino = use_dino
    self.use_depth = use_depth
    self.decoder_scan_chunk_size = decoder_scan_chunk_size  # numerically a no-op (3d:312-349); chunking is internal
    self.precision = precision
    self.workspace_fraction = workspace_fraction
    # transformer sizes of setup() (3d:89-112)
    self.num_heads, self.qkv_size = 8, 96 * 8
    self.enc_mlp, self.enc_layers = 1536, 3
    self.t2l_mlp, self.t2l_layers = 2048, 4
    self.dec_mlp, self.dec_layers = 2048, 4
    self.ro_mlp, self.ro_layers = 1536, 4
    if self.qkv_size % self.num_heads:  # attention.py:147-148
      raise ValueError(f'self.num_heads={self.num_heads} must divide self.qk_size={self.qkv_size}.')
    self._handles: Dict[Any, Any] = {}
    self._ws: Optional[torch.Tensor] = None
    self._kind, self._nc = 0, 3  # 0: 3DSPA (x,y,z); the 2-D TRAJAN subclass sets (1, 2)
    self._ws_cap = 0  # size of a budget-limited workspace (0: none / large enough for every request so far)

  # -------------------------------------------------------------------------------- handles / layout
  @property
  def act_dtype(self):
    return _PRECISIONS[self.precision][1]

  def _handle(self, dino_dim: int, depth_dim: int):
    key = (dino_dim, depth_dim)
    if key not in self._handles:
      lib = _lib.load()
      cfg = _lib.Config(self.num_output_frames, self.num_latent_tokens, self.latent_token_dim, self.num_frequencies,
                        self.track_scale_factor, self.time_scale_factor, self.track_token_dim, self.encoder_latent_dim,
                        self.decoder_num_channels, dino_dim, depth_dim, self.num_heads, self.qkv_size, self.enc_mlp,
                        self.enc_layers, self.t2l_mlp, self.t2l_layers, self.dec_mlp, self.dec_layers, self.ro_mlp,
                        self.ro_layers, _PRECISIONS[self.precision][0], self._kind)
      h = C.c_void_p()
      _lib.check(lib.spa3d_create(C.byref(cfg), C.byref(h)), what='spa3d_create')
      leaves = []
      name = C.create_string_buffer(160)
      nd = C.c_int32()
      shape = (C.c_int64 * 4)()
      off = C.c_int64()
      for i in range(lib.spa3d_num_leaves(h)):
        _lib.check(lib.spa3d_leaf_info(h, i, name, C.byref(nd), shape, C.byref(off)), h, 'spa3d_leaf_info')
        leaves.append((name.value.decode(), tuple(shape[k] for k in range(nd.value)), off.value))
      self._handles[key] = (h, leaves, lib.spa3d_param_elems(h))
    return self._handles[key]

  def _dims_from_batch(self, batch):
    dino = batch['dino_features'].shape[-1] if (self.use_dino and batch.get('dino_features') is not None) else 0
    depth = batch['depth_features'].shape[-1] if (self.use_depth and batch.get('depth_features') is not None) else 0
    return dino, depth

  @staticmethod
  def _dims_from_params(params):
    dino = params['dino_projection']['kernel'].shape[0] if 'dino_projection' in params else 0
    depth = params['depth_projection']['kernel'].shape[0] if 'depth_projection' in params else 0
    return dino, depth

  def tree_from_flat(self, flat: torch.Tensor, dino_dim: int, depth_dim: int) -> ParamTree:
    _, leaves, n = self._handle(dino_dim, depth_dim)
    assert flat.numel() == n and flat.dtype == torch.float32
    root = ParamTree()
    for name, shape, off in leaves:
      d = root
      parts = name.split('/')
      for q in parts[:-1]:
        d = d.setdefault(q, {})
      d[parts[-1]] = flat[off:off + math.prod(shape)].view(shape)
    root.flat = flat
    return root

  def flat_from_tree(self, params, device=None) -> torch.Tensor:
    """Returns the flat fp32 buffer behind `params` (zero-copy for a ParamTree, packed copy otherwise)."""
    if isinstance(params, ParamTree) and params.flat is not None:
      return params.flat
    dino, depth = self._dims_from_params(params)
    _, leaves, n = self._handle(dino, depth)
    first = params['initializer']['state_init']
    device = device or (first.device if isinstance(first, torch.Tensor) else 'cuda')
    flat = torch.zeros(n, dtype=torch.float32, device=device)
    for name, shape, off in leaves:
      d = params
      for q in name.split('/'):
        if q not in d:
          raise KeyError(f'parameter tree is missing {name!r}')  # cf. inference.py:608-619
        d = d[q]
      t = torch.as_tensor(d)
      if tuple(t.shape) != tuple(shape):
        raise ValueError(f'shape mismatch for {name}: expected {shape}, got {tuple(t.shape)}')
      flat[off:off + t.numel()] = t.to(device=device, dtype=torch.float32).reshape(-1)
    return flat

  # -------------------------------------------------------------------------------- init (train.py:233)
  def init(self, rng, batch, device=None):
    """model.init(rng, dummy_batch) -> {'params': tree}.  Flax default initialisers: lecun-normal (truncated)
    kernels, zero biases, unit norm scales, normal(1) `state_init`.  `rng`: int seed or torch.Generator.
    As in Flax, dino/depth projection leaves exist only if the batch carries those keys (3d:140,145)."""
    dino, depth = self._dims_from_batch(batch)
    _, leaves, n = self._handle(dino, depth)
    if device is None:
      st = batch.get('support_tracks')
      device = st.device if isinstance(st, torch.Tensor) and st.is_cuda else 'cuda'
    gen = rng if isinstance(rng, torch.Generator) else torch.Generator().manual_seed(int(rng))
    host = torch.zeros(n, dtype=torch.float32)
    for name, shape, off in leaves:
      numel = math.prod(shape)
      leaf = name.rsplit('/', 1)[-1]
      if leaf == 'kernel':
        fan_in = math.prod(shape[:-1]) if name.endswith('dense_out/kernel') else shape[0]
        std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
        t = torch.empty(numel, dtype=torch.float32)
        torch.nn.init.trunc_normal_(t, 0.0, 1.0, -2.0, 2.0, generator=gen)
        host[off:off + numel] = t * std
      elif leaf == 'scale':
        host[off:off + numel] = 1.0
      elif leaf == 'state_init':
        host[off:off + numel] = torch.randn(numel, generator=gen)
      # bias: zeros
    flat = host.to(device)
    return {'params': self.tree_from_flat(flat, dino, depth)}

  # -------------------------------------------------------------------------------- batch marshalling
  def _f32(self, t, name):
    _require_cuda(t, name)
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()

  def _marshal(self, inputs, dino_dim, depth_dim, need_support=True, need_query=True, targets=False, discretize=True,
               noise=None, query_points=None):
    keep = []  # keeps converted tensors alive until the call was enqueued
    b = _lib.Batch()
    st = inputs.get('support_tracks')
    if need_support:
      if st is None or inputs.get('support_tracks_visible') is None or inputs.get('boundary_frame') is None:
        raise KeyError('inputs need support_tracks, support_tracks_visible and boundary_frame')
      st = self._f32(st, 'support_tracks')
      vis = self._f32(inputs['support_tracks_visible'], 'support_tracks_visible')
      if st.dim() != 4 or st.shape[-1] != self._nc:
        raise ValueError(f'support_tracks must be [B,N,T,{self._nc}], got {tuple(st.shape)}')
      if tuple(vis.shape[:3]) != tuple(st.shape[:3]):
        raise ValueError('support_tracks_visible must be [B,N,T,1]')
      bf = inputs['boundary_frame']
      _require_cuda(bf, 'boundary_frame')
      bf = bf.to(torch.int32).contiguous()
      B, N, T = st.shape[:3]
      keep += [st, vis, bf]
      b.B, b.N, b.T = B, N, T
      b.support_tracks, b.support_tracks_visible, b.boundary_frame = st.data_ptr(), vis.data_ptr(), bf.data_ptr()
      for key, dim, field in (('dino_features', dino_dim, 'dino_features'), ('depth_features', depth_dim, 'depth_features')):
        if dim > 0:
          f = inputs.get(key)
          if f is None:
            raise KeyError(f'parameters contain a {key[:-9]}_projection but the batch has no {key!r}')
          _require_cuda(f, key)
          if tuple(f.shape) != (B, N, T, dim):
            raise ValueError(f'{key} must be {(B, N, T, dim)}, got {tuple(f.shape)}')
          f = f if (f.dtype == self.act_dtype and f.is_contiguous()) else f.to(self.act_dtype).contiguous()
          keep.append(f)
          setattr(b, field, f.data_ptr())
    if need_query:
      qp = query_points if query_points is not None else inputs.get('query_points')
      if qp is None:
        qp = self.default_query_grid(inputs['support_tracks'])
      qp = self._f32(qp, 'query_points')
      if qp.dim() != 3 or qp.shape[-1] != self._nc + 1:
        raise ValueError(f'query_points must be [B,Q,{self._nc + 1}], got {tuple(qp.shape)}')
      if need_support and qp.shape[0] != b.B:
        raise ValueError('query_points batch dimension does not match support_tracks')
      keep.append(qp)
      b.B = qp.shape[0]
      b.Q = qp.shape[1]
      b.query_points = qp.data_ptr()
    else:
      b.Q = 1
    b.discretize = 1 if discretize else 0
    if noise is not None:
      nz = self._f32(noise, 'noise')
      if tuple(nz.shape) != (b.B, self.num_latent_tokens, self.latent_token_dim):
        raise ValueError('noise must be [B, num_latent_tokens, latent_token_dim]')
      keep.append(nz)
      b.noise = nz.data_ptr()
    if targets:
      To = self.num_output_frames
      qt = self._f32(inputs['query_tracks'], 'query_tracks')
      qv = self._f32(inputs['query_tracks_visible'], 'query_tracks_visible')
      if tuple(qt.shape) != (b.B, b.Q, To, self._nc) or tuple(qv.shape[:3]) != (b.B, b.Q, To):
        raise ValueError(f'query_tracks must be {(b.B, b.Q, To, self._nc)} and query_tracks_visible {(b.B, b.Q, To, 1)}')
      keep += [qt, qv]
      b.query_tracks, b.query_tracks_visible = qt.data_ptr(), qv.data_ptr()
    return b, keep

  def default_query_grid(self, support_tracks):
    """32x32 grid with z=0 and frame 0 when no query_points are given (3d:215-226)."""
    dev = support_tracks.device
    g = torch.arange(32, dtype=torch.float32, device=dev) / 32.0 + 1.0 / 64.0
    qx, qy = torch.meshgrid(g, g, indexing='xy')
    cols = [torch.zeros_like(qx), qx, qy] + ([torch.zeros_like(qx)] if self._nc == 3 else [])  # z = 0 in 3-D (3d:218)
    q = torch.stack(cols, dim=-1).reshape(-1, self._nc + 1)
    return q[None].expand(support_tracks.shape[0], -1, -1).contiguous()

  # -------------------------------------------------------------------------------- workspace
  def _workspace(self, h, B, N, Q, T, train, device):
    lib = _lib.load()
    want = lib.spa3d_workspace_bytes(h, B, max(N, 1), Q, max(T, 1), B, 1 if train else 0)
    floor = lib.spa3d_workspace_bytes(h, B, max(N, 1), Q, max(T, 1), 1, 1 if train else 0)
    have = self._ws.numel() if (self._ws is not None and self._ws.device == torch.device(device)) else 0
    if have >= want or (self._ws_cap > 0 and have >= max(floor, self._ws_cap)):
      return self._ws  # big enough for the whole batch, or a budget cap was hit before and this is already that large
    free, _total = torch.cuda.mem_get_info(device)
    budget = int((free + have) * self.workspace_fraction)
    size = min(want, max(budget, floor))
    if size > have:
      self._ws = None
      torch.cuda.empty_cache()
      self._ws = torch.empty(size, dtype=torch.uint8, device=device)
      self._ws_cap = size if size < want else 0
    return self._ws

  # -------------------------------------------------------------------------------- public methods
  def encode(self, variables, inputs):
    """TrackAutoEncoder3D.encode (3d:190-204) -> latents [B, 128, 96] (fp32)."""
    params = variables['params'] if 'params' in variables and isinstance(variables['params'], dict) else variables
    dino, depth = self._dims_from_params(params)
    h, _, _ = self._handle(dino, depth)
    b, keep = self._marshal(inputs, dino, depth, need_query=False)
    flat = self.flat_from_tree(params)
    dev = flat.device
    lat = torch.empty(b.B, self.num_latent_tokens, self.latent_token_dim, dtype=torch.float32, device=dev)
    ws = self._workspace(h, b.B, b.N, 1, b.T, False, dev)
    _lib.check(_lib.load().spa3d_encode(h, flat.data_ptr(), C.byref(b), lat.data_ptr(), ws.data_ptr(), ws.numel(), _stream(flat)),
               h, 'spa3d_encode')
    return lat

  def get_decoder_context(self, inputs):
    """3d:206-233."""
    qp = inputs.get('query_points')
    if qp is None:
      qp = self.default_query_grid(inputs['support_tracks'])
    qp = qp.to(torch.float32)
    return TrackAutoEncoderDecoderContext(qp, torch.round(qp[..., 0]).to(torch.int32), inputs.get('boundary_frame'))

  def decode(self, variables, latents, decoder_context, discretize: bool = True, noise=None):
    """TrackAutoEncoder3D.decode (3d:248-307)."""
    params = variables['params'] if 'params' in variables and isinstance(variables['params'], dict) else variables
    dino, depth = self._dims_from_params(params)
    h, _, _ = self._handle(dino, depth)
    flat = self.flat_from_tree(params)
    dev = flat.device
    b, keep = self._marshal({}, dino, depth, need_support=False, discretize=discretize, noise=noise,
                            query_points=decoder_context.query_points)
    lat = self._f32(latents, 'latents')
    if tuple(lat.shape) != (b.B, self.num_latent_tokens, self.latent_token_dim):
      raise ValueError('latents must be [B, num_latent_tokens, latent_token_dim]')
    res, out = self._alloc_outputs(b.B, b.Q, dev)
    ws = self._workspace(h, b.B, 1, b.Q, 1, False, dev)
    _lib.check(_lib.load().spa3d_decode(h, flat.data_ptr(), C.byref(b), lat.data_ptr(), C.byref(out), ws.data_ptr(), ws.numel(),
                                        _stream(flat)), h, 'spa3d_decode')
    return res

  def _alloc_outputs(self, B, Q, dev):
    To = self.num_output_frames
    res = TrackAutoEncoderResults(
        torch.empty(B, Q, To, self._nc, dtype=torch.float32, device=dev), torch.empty(B, Q, To, 1, dtype=torch.float32, device=dev),
        torch.empty(B, Q, To, 1, dtype=torch.float32, device=dev))
    out = _lib.Outputs(res.tracks.data_ptr(), res.visible_logits.data_ptr(), res.certain_logits.data_ptr(), None)
    return res, out

  def __call__(self, variables, inputs, discretize: bool = True, noise=None):
    params = variables['params'] if 'params' in variables and isinstance(variables['params'], dict) else variables
    dino, depth = self._dims_from_params(params)
    h, _, _ = self._handle(dino, depth)
    flat = self.flat_from_tree(params)
    dev = flat.device
    b, keep = self._marshal(inputs, dino, depth, discretize=discretize, noise=noise)
    res, out = self._alloc_outputs(b.B, b.Q, dev)
    ws = self._workspace(h, b.B, b.N, b.Q, b.T, False, dev)
    _lib.check(_lib.load().spa3d_forward(h, flat.data_ptr(), C.byref(b), C.byref(out), ws.data_ptr(), ws.numel(), _stream(flat)),
               h, 'spa3d_forward')
    return res

  def apply(self, variables, *args, rngs=None, method=None, **kw):
    """Flax-style apply: model.apply({'params': p}, batch[, rngs=...][, method=model.encode])."""
    if method is None:
      return self(variables, *args, **kw)
    fn = getattr(self, method) if isinstance(method, str) else method
    name = getattr(fn, '__name__', '')
    if name == 'get_decoder_context':
      return self.get_decoder_context(*args, **kw)
    return fn(variables, *args, **kw)

  # -------------------------------------------------------------------------------- value_and_grad
  def loss_and_grads(self, variables, batch, grads_flat: Optional[torch.Tensor] = None, accumulate: bool = False,
                     denom: float = 0.0, discretize: bool = True, noise=None, return_predictions: bool = False):
    """jax.value_and_grad(loss_fn)(params) of train.py:134-162 in one call.  Returns (loss_dict, grads_tree, preds|None).
    `denom`: the batch-GLOBAL sum(query_tracks_visible) under data parallelism (train.py:111-113)."""
    params = variables['params'] if 'params' in variables and isinstance(variables['params'], dict) else variables
    dino, depth = self._dims_from_params(params)
    h, _, n = self._handle(dino, depth)
    flat = self.flat_from_tree(params)
    dev = flat.device
    b, keep = self._marshal(batch, dino, depth, targets=True, discretize=discretize, noise=noise)
    if grads_flat is None:
      grads_flat = torch.empty(n, dtype=torch.float32, device=dev)
      accumulate = False
    loss3 = torch.empty(4, dtype=torch.float32, device=dev)
    res, out, outp = None, None, None
    if return_predictions:
      res, out = self._alloc_outputs(b.B, b.Q, dev)
      outp = C.byref(out)
    ws = self._workspace(h, b.B, b.N, b.Q, b.T, True, dev)
    _lib.check(_lib.load().spa3d_loss_and_grads(h, flat.data_ptr(), C.byref(b), float(denom), grads_flat.data_ptr(),
                                                1 if accumulate else 0, loss3.data_ptr(), outp, ws.data_ptr(), ws.numel(),
                                                _stream(flat)), h, 'spa3d_loss_and_grads')
    ld = {'total_loss': loss3[0], 'position_loss': loss3[1], 'visible_loss': loss3[2]}
    return ld, self.tree_from_flat(grads_flat, dino, depth), res


def compute_loss_3d(predictions: TrackAutoEncoderResults, targets, l1_weight: float = 5000.0, bce_weight: float = 1e-8,
                    denom: float = 0.0):
  """train.py:96-129 on the GPU kernels.  Needs any model handle only for its output-frame count."""
  lib = _lib.load()
  tr = predictions.tracks
  _require_cuda(tr, 'predictions.tracks')
  B, Q, To = tr.shape[:3]
  h = _loss_handle(To, 1 if tr.shape[-1] == 2 else 0)
  b = _lib.Batch()
  b.B, b.Q = B, Q
  qt = targets['query_tracks'].to(torch.float32).contiguous()
  qv = targets['query_tracks_visible'].to(torch.float32).contiguous()
  if tuple(qt.shape) != tuple(tr.shape):
    raise ValueError(f'query_tracks {tuple(qt.shape)} does not match predictions {tuple(tr.shape)}')
  b.query_tracks, b.query_tracks_visible = qt.data_ptr(), qv.data_ptr()
  t32 = tr.to(torch.float32).contiguous()
  vl = predictions.visible_logits.to(torch.float32).contiguous()
  out = _lib.Outputs(t32.data_ptr(), vl.data_ptr(), None, None)
  loss = torch.empty(12, dtype=torch.float32, device=tr.device)
  _lib.check(lib.spa3d_loss(h, C.byref(b), C.byref(out), float(denom), loss.data_ptr(), _stream(tr)), h, 'spa3d_loss')
  pos, vis = loss[1], loss[2]
  return {'total_loss': l1_weight * pos + bce_weight * vis, 'position_loss': pos, 'visible_loss': vis}


_LOSS_HANDLES: Dict[Any, Any] = {}


def _loss_handle(num_output_frames, kind=0):
  if (num_output_frames, kind) not in _LOSS_HANDLES:
    cls = TrackAutoEncoder if kind else TrackAutoEncoder3D
    m = cls(num_output_frames=num_output_frames, precision='fp32') if kind else cls(num_output_frames=num_output_frames, use_dino=False,
                                                                                    use_depth=False, precision='fp32')
    _LOSS_HANDLES[(num_output_frames, kind)] = m._handle(0, 0)[0]
  return _LOSS_HANDLES[(num_output_frames, kind)]


def compute_loss_2d(predictions: TrackAutoEncoderResults, targets, l1_weight: float = 5000.0, bce_weight: float = 1e-8, denom: float = 0.0):
  """train.py:60-93 (same arithmetic as compute_loss_3d on 2 coordinates)."""
  return compute_loss_3d(predictions, targets, l1_weight, bce_weight, denom)


class TrackAutoEncoder(TrackAutoEncoder3D):
  """Drop-in for the 2-D TRAJAN twin track_autoencoder.TrackAutoEncoder (track_autoencoder.py:117-390): (x,y) tracks,
  no readout token (frame tokens are mean-pooled over visible frames, :230-232), a real certainty head (:344), no DINO /
  depth inputs.  Same kernels as the 3-D model; its 64-wide heads take the generic attention composition."""

  def __init__(self, num_output_frames: int = 150, num_latent_tokens: int = 128, latent_token_dim: int = 64, num_frequencies: int = 32,
               track_scale_factor: float = 1.0, time_scale_factor: float = 150.0, track_token_dim: int = 256,
               encoder_latent_dim: int = 512, decoder_num_channels: int = 1024, decoder_scan_chunk_size: Optional[int] = None,
               precision: str = 'bf16', workspace_fraction: float = 0.80):
    super().__init__(num_output_frames=num_output_frames, num_latent_tokens=num_latent_tokens, latent_token_dim=latent_token_dim,
                     num_frequencies=num_frequencies, track_scale_factor=track_scale_factor, time_scale_factor=time_scale_factor,
                     track_token_dim=track_token_dim, encoder_latent_dim=encoder_latent_dim, decoder_num_channels=decoder_num_channels,
                     dino_feature_dim=0, depth_feature_dim=0, use_dino=False, use_depth=False,
                     decoder_scan_chunk_size=decoder_scan_chunk_size, precision=precision, workspace_fraction=workspace_fraction)
    self._kind, self._nc = 1, 2
    # transformer sizes of setup() (track_autoencoder.py:149-172)
    self.num_heads, self.qkv_size = 8, 64 * 8
    self.enc_mlp, self.enc_layers = 1024, 2
    self.t2l_mlp, self.t2l_layers = 2048, 6
    self.dec_mlp, self.dec_layers = 2048, 3
    self.ro_mlp, self.ro_layers = 1024, 4


PROF_CLASSES = ('gemm_nt_bf16 (tiled MFMA, Y=X.W / dX=dY.W^T)', 'gemm_tn_bf16 (tiled MFMA, dW=X^T.dY)', 'gemm_generic (strided MFMA)',
                'attention_fused_fwd', 'attention_fused_bwd', 'layernorm_fwd', 'layernorm_bwd', 'attention_single_query (pruned last block)',
                'embed (sin features + token / dino / depth projections + readout row + prune gather; its GEMMs also count in gemm_nt_bf16)')


def profile_summary(model, handle, peak_flops: float = 2.5e15):
  """Reads the live HIP-event timings (spa3d_prof_*): one row per instrumented kernel class with its launches, device ms,
  algorithmic FLOPs and algorithmic bytes (bench.py prices each class against its own roofline)."""
  lib = _lib.load()
  rows = []
  for cls, name in enumerate(PROF_CLASSES):
    o = (C.c_double * 4)()
    _lib.check(lib.spa3d_prof_read(handle, cls, o), handle, 'spa3d_prof_read')
    rows.append({'kernel': name, 'launches': int(o[0]), 'ms': o[1], 'flops': o[2], 'bytes': o[3]})
  if not any(r['launches'] for r in rows):
    return None
  return {'classes': rows, 'peak': peak_flops}
