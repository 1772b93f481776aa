"""MI355X-native expected-free-energy rollout engine -- drop-in for the EFE hot path of
zfountas/deep-active-inference-mc (`ActiveInferenceModel.calculate_G*`, MCTS node expansion).

Import name: `daimc_amd` (the directory name `deep-active-inference-mc_amd` is not a Python identifier;
`/daimc_amd.py` at the repo root registers this package under that name).
"""
from .model import ActiveInferenceModel, ModelTop, ModelMid, ModelDown  # noqa: F401
from .mcts import (Node, MCTS_Params, active_inference_mcts, calc_threshold, normalization,  # noqa: F401
                   BatchedMCTS, active_inference_mcts_batch)
from .util import softmax_multi_with_log, plan_actions_batch, make_batch_dsprites_active_inference  # noqa: F401
from .env import Game, synthetic_sprite_bank  # noqa: F401
from .parallel import episode_shard, gather_action_posteriors  # noqa: F401
