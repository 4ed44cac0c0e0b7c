"""3dspa_code_amd -- MI355X-native (gfx950) 3DSPA TrackAutoEncoder3D train-step hot path.

The directory name is not a Python identifier; import it as `import spa3d` (repo-root shim) or
`importlib.import_module('3dspa_code_amd')`."""
from . import _lib
from .model import (ParamTree, TrackAutoEncoder, TrackAutoEncoder3D, compute_loss_2d, TrackAutoEncoderDecoderContext, TrackAutoEncoderResults, compute_loss_3d,
                    profile_summary, sinusoidal_embedding)
from .data import convert_predictions_to_tapvid3d_format, load_checkpoint, load_train_state, prepare_3d_batch, save_checkpoint
from .features import lift_2d_to_3d, sample_depth_features_for_tracks, sample_dino_features_for_tracks
from .train import TrainState, allreduce_flat_, create_learning_rate_schedule, global_visible_count

__all__ = ['TrackAutoEncoder3D', 'TrackAutoEncoder', 'compute_loss_2d', 'TrackAutoEncoderResults', 'TrackAutoEncoderDecoderContext', 'ParamTree', 'compute_loss_3d',
           'sinusoidal_embedding', 'profile_summary', 'lift_2d_to_3d', 'sample_dino_features_for_tracks',
           'sample_depth_features_for_tracks', 'prepare_3d_batch', 'convert_predictions_to_tapvid3d_format', 'load_checkpoint',
           'save_checkpoint', 'load_train_state', 'TrainState', 'create_learning_rate_schedule', 'global_visible_count', 'allreduce_flat_', '_lib']
