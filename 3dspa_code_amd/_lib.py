"""ctypes binding of libspa3d_hip.so (include/spa3d.h).  Fails loudly when the library is missing:
there is no CPU fallback for the product path."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libspa3d_hip.so')

F32, BF16, F16 = 0, 1, 2  # spa3d_config::precision / op dtype (include/spa3d.h)


class Config(C.Structure):
  _fields_ = [
      ('num_output_frames', C.c_int32), ('num_latent_tokens', C.c_int32), ('latent_token_dim', C.c_int32),
      ('num_frequencies', C.c_int32), ('track_scale_factor', C.c_float), ('time_scale_factor', C.c_float),
      ('track_token_dim', C.c_int32), ('encoder_latent_dim', C.c_int32), ('decoder_num_channels', C.c_int32),
      ('dino_feature_dim', C.c_int32), ('depth_feature_dim', C.c_int32), ('num_heads', C.c_int32),
      ('qkv_size', C.c_int32), ('enc_mlp', C.c_int32), ('enc_layers', C.c_int32), ('t2l_mlp', C.c_int32),
      ('t2l_layers', C.c_int32), ('dec_mlp', C.c_int32), ('dec_layers', C.c_int32), ('ro_mlp', C.c_int32),
      ('ro_layers', C.c_int32), ('precision', C.c_int32), ('model_kind', C.c_int32),
  ]


class Batch(C.Structure):
  _fields_ = [
      ('B', C.c_int32), ('N', C.c_int32), ('Q', C.c_int32), ('T', C.c_int32),
      ('support_tracks', C.c_void_p), ('support_tracks_visible', C.c_void_p), ('query_points', C.c_void_p),
      ('boundary_frame', C.c_void_p), ('dino_features', C.c_void_p), ('depth_features', C.c_void_p),
      ('noise', C.c_void_p), ('discretize', C.c_int32), ('query_tracks', C.c_void_p),
      ('query_tracks_visible', C.c_void_p),
  ]


class Outputs(C.Structure):
  _fields_ = [('tracks', C.c_void_p), ('visible_logits', C.c_void_p), ('certain_logits', C.c_void_p),
              ('latents', C.c_void_p)]


_SIGS = {
    'spa3d_version': (C.c_char_p, []),
    'spa3d_create': (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    'spa3d_destroy': (C.c_int, [C.c_void_p]),
    'spa3d_last_error': (C.c_char_p, [C.c_void_p]),
    'spa3d_param_elems': (C.c_int64, [C.c_void_p]),
    'spa3d_num_leaves': (C.c_int32, [C.c_void_p]),
    'spa3d_leaf_info': (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int64)]),
    'spa3d_workspace_bytes': (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'spa3d_encode': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'spa3d_decode': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Batch), C.c_void_p, C.POINTER(Outputs), C.c_void_p,
                               C.c_int64, C.c_void_p]),
    'spa3d_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Batch), C.POINTER(Outputs), C.c_void_p, C.c_int64,
                                C.c_void_p]),
    'spa3d_loss': (C.c_int, [C.c_void_p, C.POINTER(Batch), C.POINTER(Outputs), C.c_float, C.c_void_p, C.c_void_p]),
    'spa3d_loss_and_grads': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Batch), C.c_float, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.POINTER(Outputs), C.c_void_p, C.c_int64, C.c_void_p]),
    'spa3d_adamw_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_int64,
                                   C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    'spa3d_set_option': (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    'spa3d_set_loss_scale_state': (C.c_int, [C.c_void_p, C.c_void_p]),
    'spa3d_grad_segments': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    'spa3d_set_grad_events': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'spa3d_grad_events_recorded': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    'spa3d_detach': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'spa3d_plan_stats': (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    'spa3d_prof_enable': (C.c_int, [C.c_void_p, C.c_int32]),
    'spa3d_prof_read': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]),
    'spa3d_prof_dump': (C.c_int, [C.c_void_p, C.c_char_p]),
    'spa3d_uniform_noise': (C.c_int, [C.c_void_p, C.c_int64, C.c_uint32, C.c_uint32, C.c_void_p]),
    'spa3d_op_sin_embed': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    'spa3d_op_linear': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    'spa3d_op_linear_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    'spa3d_op_mlp_fused': (C.c_int, [C.c_void_p] * 9 + [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    'spa3d_op_layernorm': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                     C.c_void_p]),
    'spa3d_op_layernorm_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    'spa3d_op_qkv_attention': (C.c_int, [C.c_void_p, C.c_int64] + [C.c_void_p] * 7 + [C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 3
                               + [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    'spa3d_op_attention': (C.c_int, [C.c_void_p] * 3 + [C.c_int64] * 3 + [C.c_void_p] * 3 + [C.c_int64] + [C.c_int32] * 4
                           + [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    'spa3d_op_attention_bwd': (C.c_int, [C.c_void_p] * 3 + [C.c_int64] * 3 + [C.c_void_p] * 3 + [C.c_int64]
                               + [C.c_int32] * 4 + [C.c_void_p] * 8 + [C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                                                       C.c_void_p]),
    'spa3d_op_sample_dino': (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 7 + [C.c_void_p, C.c_int32, C.c_void_p]),
    'spa3d_op_sample_depth_features': (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p]),
    'spa3d_op_lift_2d_to_3d': (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.POINTER(C.c_double), C.c_void_p, C.c_void_p]),
}

_lib = None


def exported_symbols():
  return sorted(_SIGS)


def load():
  """dlopen the library and bind every symbol include/spa3d.h declares."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise ImportError(
        f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
        '(hipcc --offload-arch=gfx950).  The 3DSPA hot path has no CPU fallback.')
  lib = C.CDLL(LIB_PATH)
  for name, (res, args) in _SIGS.items():
    fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
    fn.restype = res
    fn.argtypes = args
  _lib = lib
  return lib


class Spa3dError(RuntimeError):
  pass


def check(rc, handle=None, what=''):
  if rc != 0:
    msg = ''
    if handle is not None:
      msg = load().spa3d_last_error(handle).decode()
    raise Spa3dError(f'{what} failed with status {rc}: {msg}')
