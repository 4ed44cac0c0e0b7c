"""Train-step glue for the 3DSPA hot path: the *intended* step of train.py:132-187 (repair R6):
value_and_grad -> clip_by_global_norm(1.0) -> adamw(lr_schedule, wd=0.01) -> apply_updates, data-parallel over
one process per GPU with RCCL (torch.distributed backend "nccl") gradient all-reduce over xGMI.

Only host glue lives here (schedule arithmetic, the two collectives); forward, loss, backward, clipping and
AdamW run in libspa3d_hip.so."""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib
from .model import TrackAutoEncoder3D, _stream


def create_learning_rate_schedule(base_lr: float, warmup_steps: int, total_steps: int):
  """train.py:41-57: optax.join_schedules([linear_schedule(0->base, warmup), cosine_decay(base, total-warmup)], [warmup])."""

  def schedule(step: int) -> float:
    if step < warmup_steps:
      return base_lr * step / max(warmup_steps, 1)
    decay = max(total_steps - warmup_steps, 1)
    s = min(step - warmup_steps, decay)
    return base_lr * 0.5 * (1.0 + math.cos(math.pi * s / decay))

  return schedule


def _collectives_on(group=None, force=False) -> bool:
  """True when the step's collectives must run: more than one rank, or `force` (world size 1: the reductions are identities, but the
  RCCL path -- communicator, device buffers, bucketed async all-reduce -- is exercised end to end; tests/test_gpu_rccl.py)."""
  if not (dist.is_available() and dist.is_initialized()):
    return False
  return force or dist.get_world_size(group) > 1


def global_visible_count(visible: torch.Tensor, group=None, force=False, check_equal_batches: bool = False) -> float:
  """max(sum(query_tracks_visible) over ALL ranks, 1): both loss terms divide by the batch-global visible count
  (train.py:111-113,119-121), so under data parallelism it is one scalar all-reduce BEFORE the backward.  The C-ABI takes the
  denominator by value, so this is one 4-byte device->host read per step (the only host sync of the multi-GPU step).
  `check_equal_batches`: the same collective also carries (B_local, B_local^2), so EVERY rank checks EVERY step that all ranks hold the same
  local batch (world * sum(B^2) == (sum B)^2) and all of them raise together -- a check issued by only some ranks (e.g. on a cache miss)
  would leave the others in the next collective and hang the job instead."""
  b = float(visible.shape[0])
  s = torch.stack([visible.to(torch.float32).sum(), visible.new_tensor(b, dtype=torch.float32), visible.new_tensor(b * b, dtype=torch.float32)])
  if _collectives_on(group, force):
    dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    h = s.tolist()
    if check_equal_batches:
      world = dist.get_world_size(group)
      if world * h[2] != h[1] * h[1]:
        raise ValueError(f'local batch sizes differ across ranks (this rank: {int(b)}, mean {h[1] / world:g}); the global discretisation '
                         'noise is sliced by equal per-rank batches and the gradient SUM assumes them')
    return max(h[0], 1.0)
  return max(float(s[0].item()), 1.0)


def allreduce_flat_(flat: torch.Tensor, bucket_elems: int, group=None, extra=(), force=False):
  """In-place SUM all-reduce of a flat buffer in a few large buckets (RCCL ring: per-link xGMI bound, so few and
  large), all in flight together; `extra` small tensors ride along.  Returns when every bucket has been reduced."""
  if not _collectives_on(group, force):
    return
  works = []
  n = flat.numel()
  for s in range(0, n, bucket_elems):
    works.append(dist.all_reduce(flat[s:s + bucket_elems], op=dist.ReduceOp.SUM, group=group, async_op=True))
  for t in extra:
    works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True))
  for w in works:
    w.wait()


def allreduce_segments_overlapped_(flat: torch.Tensor, bounds, events, side: 'torch.cuda.Stream', bucket_elems: int, group=None, extra=()):
  """SUM all-reduce of the flat gradient buffer in the order its segments become final during the last sample chunk's backward
  (include/spa3d.h, spa3d_grad_segments): segment [b2, n) once `events[0]` has fired, [b1, b2) once `events[1]` has, both on the side stream
  `side` and therefore UNDER the rest of the backward that is still running on the launch stream; [0, b1) and the `extra` scalars on the
  launch stream behind everything.  The events were recorded by spa3d_loss_and_grads on the launch stream before it returned (the call
  enqueues; it does not wait), so waiting for them here is well ordered.  Returns when every reduction has been enqueued and the launch
  stream has been made to wait for the side stream."""
  main = torch.cuda.current_stream(flat.device)
  works = []
  for ev, (lo, hi) in zip(events, ((bounds[2], bounds[3]), (bounds[1], bounds[2]))):
    side.wait_event(ev)
    with torch.cuda.stream(side):
      for s_ in range(lo, hi, bucket_elems):
        works.append(dist.all_reduce(flat[s_:min(s_ + bucket_elems, hi)], op=dist.ReduceOp.SUM, group=group, async_op=True))
  for s_ in range(0, bounds[1], bucket_elems):
    works.append(dist.all_reduce(flat[s_:min(s_ + bucket_elems, bounds[1])], op=dist.ReduceOp.SUM, group=group, async_op=True))
  for t in extra:
    works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True))
  for w in works:
    w.wait()  # stream-level wait (NCCL): the launch stream continues behind the reductions
  main.wait_stream(side)


def broadcast_state_(tensors, src: int = 0, group=None, force=False):
  """Replicas must start from rank `src`'s parameters and Adam moments whatever each rank initialised or loaded
  (a checkpoint read on rank 0 only, different seeds): one broadcast per buffer at construction / resume."""
  if not _collectives_on(group, force):
    return
  gsrc = dist.get_global_rank(group, src) if group is not None else src  # dist.broadcast takes a GLOBAL rank; `src` is a rank of `group`
  for t in tensors:
    dist.broadcast(t, src=gsrc, group=group)


def _hip_uniform_noise(n: int, device) -> torch.Tensor:
  out = torch.empty(n, dtype=torch.float32, device=device)
  _lib.check(_lib.load().spa3d_uniform_noise(out.data_ptr(), n, 0, 0, _stream(out)), what='spa3d_uniform_noise')
  return out


class TrainState:
  """create_model_state + train_step of train.py:132-187,217-260 for model_type='3dspa'.

  `compute`, `adamw` and `noise_fn` default to the HIP library (there is no CPU fallback); they are injectable so that the
  host logic of the data-parallel step -- denominator pre-reduce, rank slice of the global discretisation noise, bucketed
  SUM all-reduce, clip + AdamW on reduced gradients, rank-0 broadcast -- can be driven at world size 2 over gloo on CPU
  (tests/test_dp_gloo.py) with a stand-in compute.
    compute(params_tree, batch, grads_flat, denom, discretize, noise) -> {'total_loss','position_loss','visible_loss'}
    adamw(flat, grads, m, v, lr, step, clip, b1, b2, eps, wd, scratch) -> None  (scratch[0] := global grad norm)
    noise_fn(n, device) -> float32[n] = jax.random.uniform(PRNGKey(0), [n]) (track_autoencoder_3d.py:254-257)
    evaluate(params_tree, batch, denom, discretize, noise) -> (loss dict, predictions)   (eval_step, train.py:189-213)"""

  def __init__(self, model: TrackAutoEncoder3D, params, learning_rate: float = 1e-4, warmup_steps: int = 10000,
               total_steps: int = 1000000, weight_decay: float = 0.01, clip_norm: float = 1.0, b1: float = 0.9,
               b2: float = 0.999, eps: float = 1e-8, process_group: Optional[dist.ProcessGroup] = None,
               grad_bucket_bytes: int = 128 << 20, compute=None, adamw=None, noise_fn=None, force_collectives: bool = False,
               evaluate=None, overlap_allreduce: bool = True):
    self.model = model
    self.params = params if hasattr(params, 'flat') and params.flat is not None else None
    flat = model.flat_from_tree(params)
    if self.params is None:
      dino, depth = model._dims_from_params(params)
      self.params = model.tree_from_flat(flat, dino, depth)
    self.flat = flat
    self.m = torch.zeros_like(flat)
    self.v = torch.zeros_like(flat)
    self.grads = torch.zeros_like(flat)
    self.scratch = torch.zeros(1024, dtype=torch.float32, device=flat.device)
    self.step = 0
    self.schedule = create_learning_rate_schedule(learning_rate, warmup_steps, total_steps)
    self.wd, self.clip, self.b1, self.b2, self.eps = weight_decay, clip_norm, b1, b2, eps
    self.pg = process_group
    on = dist.is_available() and dist.is_initialized()
    self.world = dist.get_world_size(process_group) if on else 1
    self.rank = dist.get_rank(process_group) if on else 0
    self.bucket_elems = max(1, grad_bucket_bytes // 4)
    self.force = bool(force_collectives) and on  # run the collectives at world size 1 too (RCCL smoke test)
    self._compute = compute or self._hip_compute
    self._adamw = adamw or self._hip_adamw
    self._noise_fn = noise_fn or _hip_uniform_noise
    self._evaluate = evaluate or self._hip_evaluate
    self._noise_cache = {}
    # gradient all-reduce under the backward: needs the HIP compute (it records the segment events) and a GPU; fp16 too -- the library
    # unscales a finished segment (a power of two: exact) right before it records the segment's event
    self._overlap = None
    import os
    if os.environ.get('SPA3D_DP_OVERLAP', '1') == '0':  # A/B switch for scaling runs
      overlap_allreduce = False
    if overlap_allreduce and compute is None and on and flat.is_cuda and (self.world > 1 or self.force):
      lib = _lib.load()
      h = model._handle(*model._dims_from_params(self.params))[0]
      b4 = (C.c_int64 * 4)()
      _lib.check(lib.spa3d_grad_segments(h, b4), h, 'spa3d_grad_segments')
      evs = (torch.cuda.Event(), torch.cuda.Event())
      for e in evs:
        e.record()  # instantiates the underlying hipEvent_t
      # last writer wins on the handle (one state trains a model at a time): a state replaced here keeps its own events but they are no
      # longer recorded, which train_step detects through spa3d_grad_events_recorded and answers with the stream-ordered all-reduce
      _lib.check(lib.spa3d_set_grad_events(h, C.c_void_p(evs[0].cuda_event), C.c_void_p(evs[1].cuda_event)), h, 'spa3d_set_grad_events')
      self._overlap = (tuple(int(x) for x in b4), evs, torch.cuda.Stream(device=flat.device))
      self._ev_gen = self._events_recorded()
    if model.precision == 'fp16' and compute is None and adamw is None:
      # dynamic loss scale: spa3d_adamw_step skips a step whose gradient norm is inf/NaN and halves the multiplier kept in scratch[4],
      # which the next spa3d_loss_and_grads on this handle applies (include/spa3d.h)
      h = model._handle(*model._dims_from_params(self.params))[0]
      _lib.check(_lib.load().spa3d_set_loss_scale_state(h, self.scratch.data_ptr() + 16), h, 'spa3d_set_loss_scale_state')
      self._scale_state_on = True
    self.sync_from_rank0()

  def _events_recorded(self):
    h = self.model._handle(*self.model._dims_from_params(self.params))[0]
    g4 = (C.c_int64 * 4)()
    _lib.check(_lib.load().spa3d_grad_events_recorded(h, g4), h, 'spa3d_grad_events_recorded')
    mine = self._overlap is not None and int(g4[2]) == int(self._overlap[1][0].cuda_event) and int(g4[3]) == int(self._overlap[1][1].cuda_event)
    return int(g4[0]), int(g4[1]), mine

  def close(self):
    """Detach what THIS state registered on the model's handle (gradient-segment events, the loss-scale state in `scratch`): the handle
    outlives the state (it belongs to the model), the events and the scratch buffer do not.  Owner-aware (spa3d_detach compares pointers):
    a state created later on the same model -- a resume, `state = TrainState(model, ...)` rebinding, whose predecessor's __del__ runs
    AFTER the new constructor -- keeps its registration."""
    try:
      lib = _lib.load()
      h = self.model._handle(*self.model._dims_from_params(self.params))[0]
      ov = getattr(self, '_overlap', None)
      scale = self.scratch.data_ptr() + 16 if getattr(self, '_scale_state_on', False) else None
      if ov is not None or scale is not None:
        e0 = C.c_void_p(ov[1][0].cuda_event) if ov is not None else None
        e1 = C.c_void_p(ov[1][1].cuda_event) if ov is not None else None
        lib.spa3d_detach(h, e0, e1, C.c_void_p(scale) if scale is not None else None)
      self._overlap = None
      self._scale_state_on = False
    except Exception:
      pass

  def __del__(self):
    self.close()

  def sync_from_rank0(self):
    """parameters + Adam moments + optimizer counters of every replica := rank 0's (also after load_train_state on rank 0 only).
    scratch[3:6] = skipped-update count (enters the Adam bias correction), loss-scale multiplier, its good-step counter: a replica that kept its own
    would apply a different update after a resume and drift apart silently."""
    broadcast_state_((self.flat, self.m, self.v, self.scratch[3:6]), 0, self.pg, self.force)

  # ---- default (HIP) compute and optimizer
  def _hip_compute(self, params, batch, grads_flat, denom, discretize, noise):
    ld, _, _ = self.model.loss_and_grads(params, batch, grads_flat=grads_flat, accumulate=False, denom=denom,
                                         discretize=discretize, noise=noise)
    return ld

  def _hip_adamw(self, flat, grads, m, v, lr, step, clip, b1, b2, eps, wd, scratch):
    _lib.check(_lib.load().spa3d_adamw_step(flat.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr(), flat.numel(), lr, step,
                                            clip, b1, b2, eps, wd, scratch.data_ptr(), _stream(flat)), what='spa3d_adamw_step')

  def _hip_evaluate(self, params, batch, denom, discretize, noise):
    from .model import compute_loss_2d, compute_loss_3d
    preds = self.model.apply({'params': params}, batch, discretize=discretize, noise=noise)
    loss_fn = compute_loss_2d if preds.tracks.shape[-1] == 2 else compute_loss_3d
    return loss_fn(preds, batch, denom=denom), preds

  def rank_noise(self, b_local: int):
    """The reference draws uniform(PRNGKey(0), [B_global, L, Ld]) over the GLOBAL batch (3d:254-258); rank r owns rows
    r*B_local .. of that tensor, not a draw of its own over [B_local, L, Ld].  Fixed key => drawn once and cached."""
    key = (b_local, self.world, self.rank)
    if key not in self._noise_cache:  # (equal local batches are checked every step inside global_visible_count's all-reduce)
      L, Ld = self.model.num_latent_tokens, self.model.latent_token_dim
      full = self._noise_fn(b_local * self.world * L * Ld, self.flat.device).view(self.world, b_local, L, Ld)
      self._noise_cache = {key: full[self.rank].clone()}
    return self._noise_cache[key]

  def train_step(self, batch, discretize: bool = True, noise=None):
    multi = self.world > 1 or self.force
    denom = global_visible_count(batch['query_tracks_visible'], self.pg, self.force, check_equal_batches=True) if multi else 0.0
    if multi and discretize and noise is None:
      noise = self.rank_noise(batch['query_tracks_visible'].shape[0])
    ld = self._compute(self.params, batch, self.grads, denom, discretize, noise)
    l3 = torch.stack([torch.as_tensor(ld[k], dtype=torch.float32, device=self.flat.device).reshape(())
                      for k in ('total_loss', 'position_loss', 'visible_loss')])
    # the per-rank gradients and loss terms already carry the global 1/denominator -> plain SUM over ranks
    use_overlap = False
    if self._overlap is not None and multi:
      gen = self._events_recorded()  # did THIS call record both segment events?  (another state may own the handle's events by now)
      use_overlap = gen[2] and gen[0] == self._ev_gen[0] + 1 and gen[1] == self._ev_gen[1] + 1
      self._ev_gen = gen
    if use_overlap:
      bounds, evs, side = self._overlap
      allreduce_segments_overlapped_(self.grads, bounds, evs, side, self.bucket_elems, self.pg, extra=(l3,))
    else:
      allreduce_flat_(self.grads, self.bucket_elems, self.pg, extra=(l3,), force=self.force)
    lr = self.schedule(self.step)
    # `step` counts CALLS.  A skipped update (non-finite gradient norm; scratch[3] counts them on the device) leaves m and v untouched, and
    # spa3d_adamw_step takes the Adam bias correction at step - skipped, so the optimizer count does not run ahead of the moments; the
    # learning-rate schedule is host arithmetic on `step` and does advance through a skip (reading the counter back would cost a sync).
    self._adamw(self.flat, self.grads, self.m, self.v, lr, self.step, self.clip, self.b1, self.b2, self.eps, self.wd, self.scratch)
    self.step += 1
    # metric keys of train.py:180-185 (device scalars)
    # 'train/skipped' = 1 when the update was skipped (non-finite gradient norm: fp16 overflow), 'train/loss_scale_mult' the dynamic multiplier
    return {'train/loss': l3[0], 'train/position_loss': l3[1], 'train/visible_loss': l3[2], 'train/learning_rate': lr,
            'train/grad_norm': self.scratch[0], 'train/skipped': self.scratch[2], 'train/loss_scale_mult': self.scratch[4]}

  def skipped_steps(self) -> int:
    """Updates skipped so far because the gradient norm was not finite (device counter scratch[3]; reading it synchronises)."""
    return int(self.scratch[3].item())

  def check_finite(self):
    """For a training loop's logging interval (one device read): in fp16 a skipped step is the loss-scale mechanism at work; in bf16 / fp32 there is
    no scale to lower, so a skipped LAST step means the gradients themselves are NaN / inf and every later step will be skipped too -- where the
    reference's optax update would have propagated the NaN into the parameters.  Raises FloatingPointError in that case instead of training on silently."""
    if float(self.scratch[2].item()) != 0.0 and getattr(self.model, 'precision', 'fp16') != 'fp16':
      raise FloatingPointError(f'non-finite gradient norm at step {self.step - 1} ({self.skipped_steps()} skipped updates so far)')

  def eval_step(self, batch, discretize: bool = True, noise=None):
    """train.py:189-213: forward pass + compute_loss on the current parameters, no update.  Returns (metrics, predictions) with the
    reference's metric keys 'eval/loss', 'eval/position_loss', 'eval/visible_loss' (device scalars).  Under data parallelism the two
    loss terms are normalised by the GLOBAL visible count and summed over ranks, exactly as the training loss is, and the rank
    takes its slice of the global discretisation noise; predictions are the rank's own."""
    multi = self.world > 1 or self.force
    denom = global_visible_count(batch['query_tracks_visible'], self.pg, self.force, check_equal_batches=True) if multi else 0.0
    if multi and discretize and noise is None:
      noise = self.rank_noise(batch['query_tracks_visible'].shape[0])
    ld, preds = self._evaluate(self.params, batch, denom, discretize, noise)
    l3 = torch.stack([torch.as_tensor(ld[k], dtype=torch.float32, device=self.flat.device).reshape(())
                      for k in ('total_loss', 'position_loss', 'visible_loss')])
    if _collectives_on(self.pg, self.force):
      dist.all_reduce(l3, op=dist.ReduceOp.SUM, group=self.pg)
    return {'eval/loss': l3[0], 'eval/position_loss': l3[1], 'eval/visible_loss': l3[2]}, preds
