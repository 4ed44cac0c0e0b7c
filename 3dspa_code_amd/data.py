"""Host-side steps either side of the hot path, with the reference's names:

  prepare_3d_batch                         data_loader.py:56-110   support/query split + query-point sampling
  convert_predictions_to_tapvid3d_format   evaluate_tapvid3d.py:39-59
  load_checkpoint / save_checkpoint        inference.py:450-508, evaluate_tapvid3d.py:247-285 / train.py:389-393 (a stub upstream)

Plain NumPy / torch glue: no arithmetic worth a kernel.  The metric arithmetic of TAPVid-3D stays in the un-vendored
`tapnet` package, as upstream."""
from __future__ import annotations

import os
from typing import Any, Dict, Optional

import numpy as np
import torch


def prepare_3d_batch(example, num_support_tracks: int = 2048, num_query_tracks: int = 2048, num_frames: int = 150, use_dino: bool = True,
                     use_depth: bool = True, device='cuda', feature_dtype=torch.bfloat16):
  """data_loader.py:56-110.  Same RNG call sequence on NumPy's global generator as the reference
  (`np.random.permutation(num_total)` then one `np.random.randint(0, num_frames)` per query track), so a seeded
  reference run and a seeded run of this function pick the same tracks and frames."""
  tracks_3d = np.asarray(example['tracks_3d'])  # [N, T, 3]
  visible = np.asarray(example['visible'])  # [N, T, 1]
  num_total = tracks_3d.shape[0]
  indices = np.random.permutation(num_total)
  support_indices = indices[:num_support_tracks]
  query_indices = indices[num_support_tracks:num_support_tracks + num_query_tracks]
  if len(query_indices) < num_query_tracks:  # the reference would raise IndexError inside its sampling loop
    raise IndexError(f'example has {num_total} tracks, need {num_support_tracks} support + {num_query_tracks} query')
  query_tracks = tracks_3d[query_indices]
  ts = np.array([np.random.randint(0, num_frames) for _ in range(num_query_tracks)])
  xyz = query_tracks[np.arange(num_query_tracks), ts]
  query_points = np.concatenate([ts[:, None].astype(np.float64), xyz.astype(np.float64)], axis=1)  # np.array of python lists -> float64

  def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(device=device, dtype=dtype)[None]

  batch = {
      'support_tracks': dev(tracks_3d[support_indices]),
      'support_tracks_visible': dev(visible[support_indices]),
      'query_points': dev(query_points),
      'query_tracks': dev(query_tracks),
      'query_tracks_visible': dev(visible[query_indices]),
      'boundary_frame': torch.tensor([num_frames], dtype=torch.int32, device=device),
  }
  if use_dino and 'dino_features' in example:
    batch['dino_features'] = dev(np.asarray(example['dino_features'])[support_indices], feature_dtype)
  if use_depth and 'depth_features' in example:
    batch['depth_features'] = dev(np.asarray(example['depth_features'])[support_indices], feature_dtype)
  return batch


def convert_predictions_to_tapvid3d_format(predictions, query_points=None):
  """evaluate_tapvid3d.py:39-59: [B,Q,T,3] -> pred_tracks [T,Q,3], pred_occluded [T,Q] (True = occluded, logit <= 0)."""
  pred_tracks = predictions.tracks.detach().float().cpu().numpy()[0]
  logits = predictions.visible_logits.detach().float().cpu().numpy()[0, :, :, 0]
  return np.transpose(pred_tracks, (1, 0, 2)), np.transpose(logits <= 0.0, (1, 0))


# ------------------------------------------------------------------------------------------------ checkpoints
def _flatten(tree, prefix=''):
  out = {}
  for k, v in tree.items():
    key = f'{prefix}/{k}' if prefix else k
    if isinstance(v, dict):
      out.update(_flatten(v, key))
    else:
      out[key] = v
  return out


def _unflatten_params(flat: Dict[str, Any]) -> Dict[str, Any]:
  """inference.py:450-461."""
  result: Dict[str, Any] = {}
  for key, value in flat.items():
    parts = key.split('/')
    d = result
    for part in parts[:-1]:
      d = d.setdefault(part, {})
    d[parts[-1]] = value
  return result


def save_checkpoint(path: str, params, state=None):
  """The save the reference leaves as a log line (train.py:389-393).  Flat 'a/b/c' keys -- the third layout its own
  loader accepts (inference.py:483-485) -- so the file loads in the reference too; optimizer moments and the step go
  under 'opt_m/...', 'opt_v/...', 'step' when a TrainState is given (resume)."""
  flat = {k: v.detach().float().cpu().numpy() for k, v in _flatten(params).items()}
  if state is not None:
    names = list(_flatten(state.params).keys())
    m_tree = state.model.tree_from_flat(state.m, *state.model._dims_from_params(state.params))
    v_tree = state.model.tree_from_flat(state.v, *state.model._dims_from_params(state.params))
    flat.update({f'opt_m/{k}': t.detach().cpu().numpy() for k, t in _flatten(m_tree).items()})
    flat.update({f'opt_v/{k}': t.detach().cpu().numpy() for k, t in _flatten(v_tree).items()})
    flat['step'] = np.array(state.step, dtype=np.int64)
    # skipped-step counter, dynamic loss-scale multiplier and its good-step counter (spa3d_adamw_step scratch[3..5]) belong to the optimizer state
    flat['opt_scratch'] = state.scratch[3:6].detach().cpu().numpy()
    assert names
  if not path.endswith('.npz'):
    path = path + '.npz'
  os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
  np.savez(path, **flat)
  return path


def load_checkpoint(checkpoint_path: str, model=None, allow_pickle: bool = False):
  """inference.py:464-508 for `.npz` files.  Flat-key files (and files written by save_checkpoint) load with
  allow_pickle=False; the two pickled layouts ('params' / 'optimizer' object arrays) execute code from the file when
  unpickled, so they need an explicit allow_pickle=True from the caller.  Returns the nested parameter dict (optimizer
  entries, if present, under the keys 'opt_m', 'opt_v', 'step' are stripped -- use load_train_state for those)."""
  if not os.path.exists(checkpoint_path):
    raise FileNotFoundError(f'Checkpoint not found: {checkpoint_path}')
  if not checkpoint_path.endswith('.npz'):  # Flax format (evaluate_tapvid3d.py:278-285, inference.py:490-503)
    state_dict = restore_flax_checkpoint(checkpoint_path)
    if 'params' in state_dict:
      return state_dict['params']
    if 'optimizer' in state_dict and isinstance(state_dict['optimizer'], dict) and 'target' in state_dict['optimizer']:
      return state_dict['optimizer']['target']
    return state_dict
  data = np.load(checkpoint_path, allow_pickle=allow_pickle)
  if 'params' in data.files:
    p = data['params']
    params = p.item() if p.ndim == 0 else dict(p)
  elif 'optimizer' in data.files:
    opt = data['optimizer']
    opt = opt.item() if opt.ndim == 0 else dict(opt)
    params = opt.get('target', opt) if isinstance(opt, dict) else opt
  else:
    flat = {k: np.array(data[k]) for k in data.files if not (k.startswith('opt_m/') or k.startswith('opt_v/') or k in ('step', 'opt_scratch'))}
    params = _unflatten_params(flat)
  return params


# ------------------------------------------------------------------------------------------------ Flax msgpack checkpoints
# `flax.training.checkpoints.restore_checkpoint(path, target=None)` (evaluate_tapvid3d.py:278, inference.py:490) restated: Flax is not
# installed here and the reference holds no such file, so this follows Flax's documented serialization format -- PARITY UNPINNED:
#   file      = msgpack(state_dict)                                  nested dicts with str keys
#   ndarray   = ExtType(1, msgpack((shape, dtype_name, raw C-order bytes)))
#   complex   = ExtType(2, msgpack((real, imag)))          np scalar = ExtType(3, same tuple as ndarray)
#   arrays over 2**30 bytes are stored as {'__msgpack_chunked_array__': True, 'shape': [...], 'chunks': {'0': ndarray, '1': ...}}
#   a checkpoint DIRECTORY holds files `<prefix><step>`; the largest step wins (natural ordering).
_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3


def _nd_from_bytes(data: bytes):
  import msgpack
  shape, dtype_name, buf = msgpack.unpackb(data, raw=True)
  name = dtype_name.decode() if isinstance(dtype_name, bytes) else dtype_name
  if name == 'bfloat16':  # not a NumPy dtype: widen exactly to float32
    u16 = np.frombuffer(buf, dtype=np.uint16).astype(np.uint32) << 16
    return u16.view(np.float32).reshape(tuple(shape))
  return np.frombuffer(buf, dtype=np.dtype(name)).reshape(tuple(shape)).copy()


def _ext_hook(code, data):
  import msgpack
  if code == _EXT_NDARRAY:
    return _nd_from_bytes(data)
  if code == _EXT_COMPLEX:
    re_, im_ = msgpack.unpackb(data)
    return complex(re_, im_)
  if code == _EXT_NPSCALAR:
    return _nd_from_bytes(data)[()]
  return msgpack.ExtType(code, data)


def _unchunk(tree):
  if isinstance(tree, dict):
    if tree.get('__msgpack_chunked_array__'):
      # Flax writes both 'chunks' and 'shape' through _tuple_to_dict: {'0': .., '1': ..} with str keys (a plain list is accepted too).
      # Only the unchunked case has a round-trip here (msgpack_serialize refuses > 2**30-byte arrays); this branch is exercised by a
      # hand-built fixture (tests/test_data_host.py); parity unpinned: no Flax here, no msgpack file in the reference.
      def seq(v):
        return [v[str(i)] for i in range(len(v))] if isinstance(v, dict) else list(v)
      chunks = seq(tree['chunks'])
      return np.concatenate([np.asarray(c).reshape(-1) for c in chunks]).reshape(tuple(int(x) for x in seq(tree['shape'])))
    return {k: _unchunk(v) for k, v in tree.items()}
  return tree


def msgpack_restore(raw: bytes):
  """flax.serialization.msgpack_restore: bytes -> nested dict of NumPy arrays / scalars."""
  import msgpack
  return _unchunk(msgpack.unpackb(raw, ext_hook=_ext_hook, raw=False, strict_map_key=False))


def msgpack_serialize(tree) -> bytes:
  """flax.serialization.msgpack_serialize for nested dicts of arrays / Python scalars (arrays under 2**30 bytes)."""
  import msgpack

  def enc(o):
    if isinstance(o, torch.Tensor):
      o = o.detach().cpu().numpy()
    if isinstance(o, np.ndarray):
      if o.nbytes > 2**30:
        raise ValueError('chunked arrays are not written (every 3DSPA leaf is far below 2**30 bytes)')
      return msgpack.ExtType(_EXT_NDARRAY, msgpack.packb((list(o.shape), o.dtype.name, o.tobytes('C')), use_bin_type=True))
    if isinstance(o, np.generic):
      return msgpack.ExtType(_EXT_NPSCALAR, msgpack.packb(([], o.dtype.name, o.tobytes()), use_bin_type=True))
    if isinstance(o, complex):
      return msgpack.ExtType(_EXT_COMPLEX, msgpack.packb((o.real, o.imag)))
    raise TypeError(f'cannot serialize {type(o)}')

  return msgpack.packb(tree, default=enc, use_bin_type=True, strict_types=True)


def restore_flax_checkpoint(ckpt_dir: str, prefix: str = 'checkpoint_'):
  """checkpoints.restore_checkpoint(ckpt_dir, target=None): a file, or the newest `<prefix><step>` file of a directory."""
  import re
  path = ckpt_dir
  if os.path.isdir(ckpt_dir):
    def step_of(fn):
      m = re.fullmatch(re.escape(prefix) + r'(\d+(?:\.\d+)?)', fn)
      return float(m.group(1)) if m else None
    cands = [(step_of(f), f) for f in os.listdir(ckpt_dir)]
    cands = [c for c in cands if c[0] is not None]
    if not cands:
      raise ValueError(f'Checkpoint at {ckpt_dir} is empty or invalid')  # evaluate_tapvid3d.py:279-280
    path = os.path.join(ckpt_dir, max(cands)[1])
  with open(path, 'rb') as f:
    return msgpack_restore(f.read())


def _group_src(group, src: int = 0) -> int:
  """dist.broadcast takes a GLOBAL rank; `src` is a rank of `group` (a sub-group's rank 0 is generally not global rank 0)."""
  import torch.distributed as dist
  return dist.get_global_rank(group, src) if group is not None else src


def load_train_state(checkpoint_path: str, state, rank0_only: bool = False):
  """Resume: parameters, Adam moments and step back into a TrainState (in place).  Under data parallelism every replica ends
  with rank 0's state: with `rank0_only` only rank 0 reads the file and the buffers + step are broadcast."""
  import torch.distributed as dist
  multi = dist.is_available() and dist.is_initialized() and state.world > 1
  if multi and rank0_only and state.rank != 0:
    step = torch.zeros(1, dtype=torch.int64, device=state.flat.device)
    state.sync_from_rank0()
    dist.broadcast(step, src=_group_src(state.pg), group=state.pg)
    state.step = int(step.item())
    return state
  data = np.load(checkpoint_path, allow_pickle=False)
  model = state.model
  dims = model._dims_from_params(state.params)
  _, leaves, _ = model._handle(*dims)
  for name, shape, off in leaves:
    n = int(np.prod(shape))
    for buf, prefix in ((state.flat, ''), (state.m, 'opt_m/'), (state.v, 'opt_v/')):
      key = prefix + name
      if key not in data.files:
        raise KeyError(f'checkpoint is missing {key!r}')  # cf. check_params_structure, inference.py:608-619
      a = data[key]
      if tuple(a.shape) != tuple(shape):
        raise ValueError(f'shape mismatch for {key}: expected {shape}, got {a.shape}')
      buf[off:off + n] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).reshape(-1).to(buf.device)
  state.step = int(data['step'])
  if 'opt_scratch' in data.files:  # absent in checkpoints written before round 4: counters start at 0, multiplier at 1
    state.scratch[3:6] = torch.from_numpy(np.asarray(data['opt_scratch'], dtype=np.float32)).to(state.scratch.device)
  if multi:
    state.sync_from_rank0()
    if rank0_only:
      dist.broadcast(torch.tensor([state.step], dtype=torch.int64, device=state.flat.device), src=_group_src(state.pg), group=state.pg)
  return state
