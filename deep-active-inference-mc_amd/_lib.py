"""ctypes binding of include/efe_engine.h.  There is NO fallback: if the HIP library is missing the
import fails loudly (the product path never routes through the CPU oracle)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('EFE_LIB_PATH') or os.path.join(HERE, 'libefe_mi355x.so')    # override: A/B of kernel builds

EXPORTS = ['efe_create', 'efe_destroy', 'efe_last_error', 'efe_abi_version', 'efe_set_weight', 'efe_commit_weights',
           'efe_set_option', 'efe_transition', 'efe_decoder', 'efe_encoder', 'efe_habit', 'efe_calculate_g',
           'efe_rollout', 'efe_trajectory', 'efe_simulate', 'efe_action_posterior', 'efe_last_call_macs',
           'efe_prof_enable', 'efe_prof_classes', 'efe_prof_read', 'efe_env_reset', 'efe_env_step', 'efe_env_render',
           'efe_check_reward', 'efe_reparameterize', 'efe_mcts_select', 'efe_mcts_expand', 'efe_mcts_backprop', 'efe_mcts_stop']


class EfeMctsTree(C.Structure):
    _fields_ = [('W', C.c_void_p), ('N', C.c_void_p), ('Qpi', C.c_void_p), ('child', C.c_void_p), ('S', C.c_void_p),
                ('E', C.c_int32), ('cap', C.c_int32), ('A', C.c_int32), ('s_dim', C.c_int32)]


class EfeNoise(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('stage', C.c_uint32), ('pass_', C.c_uint32), ('sample', C.c_uint32),
                ('row_offset', C.c_uint32)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f'{LIB_PATH} not built: run `python -c "import __graft_entry__ as g; g.build()"` '
                          '(hipcc --offload-arch=gfx950); there is no CPU fallback')
    lib = C.CDLL(LIB_PATH)
    p, i, f32p = C.c_void_p, C.c_int, C.c_void_p
    nzp = C.POINTER(EfeNoise)
    lib.efe_create.argtypes = [C.POINTER(p), i]; lib.efe_create.restype = i
    lib.efe_destroy.argtypes = [p]; lib.efe_destroy.restype = None
    lib.efe_last_error.argtypes = [p]; lib.efe_last_error.restype = C.c_char_p
    lib.efe_abi_version.argtypes = []; lib.efe_abi_version.restype = i
    lib.efe_set_weight.argtypes = [p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), i]; lib.efe_set_weight.restype = i
    lib.efe_commit_weights.argtypes = [p]; lib.efe_commit_weights.restype = i
    lib.efe_set_option.argtypes = [p, C.c_char_p, C.c_int64]; lib.efe_set_option.restype = i
    lib.efe_transition.argtypes = [p, f32p, f32p, i, nzp, f32p, f32p, f32p, f32p, p]; lib.efe_transition.restype = i
    lib.efe_decoder.argtypes = [p, f32p, i, nzp, f32p, p]; lib.efe_decoder.restype = i
    lib.efe_encoder.argtypes = [p, f32p, i, nzp, f32p, f32p, f32p, f32p, p]; lib.efe_encoder.restype = i
    lib.efe_habit.argtypes = [p, f32p, i, f32p, f32p, f32p, p]; lib.efe_habit.restype = i
    lib.efe_calculate_g.argtypes = [p, f32p, f32p, i, i, i, nzp, f32p, f32p, f32p, f32p, f32p, f32p, f32p, p]
    lib.efe_calculate_g.restype = i
    lib.efe_rollout.argtypes = [p, f32p, f32p, i, i, i, i, i, nzp, f32p, f32p, f32p, f32p, p]; lib.efe_rollout.restype = i
    lib.efe_trajectory.argtypes = [p, f32p, f32p, f32p, f32p, f32p, i, nzp, f32p, f32p, p]; lib.efe_trajectory.restype = i
    lib.efe_simulate.argtypes = [p, f32p, i, i, i, nzp, f32p, f32p, f32p, p]; lib.efe_simulate.restype = i
    lib.efe_action_posterior.argtypes = [p, f32p, i, i, C.c_float, f32p, f32p, p]; lib.efe_action_posterior.restype = i
    lib.efe_last_call_macs.argtypes = [p]; lib.efe_last_call_macs.restype = C.c_int64
    lib.efe_prof_enable.argtypes = [p, i]; lib.efe_prof_enable.restype = i
    lib.efe_prof_classes.argtypes = []; lib.efe_prof_classes.restype = i
    lib.efe_prof_read.argtypes = [p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]; lib.efe_prof_read.restype = i
    lib.efe_env_reset.argtypes = [p, f32p, f32p, i, nzp, p]; lib.efe_env_reset.restype = i
    lib.efe_env_step.argtypes = [p, f32p, f32p, C.c_void_p, i, i, nzp, C.c_void_p, p]; lib.efe_env_step.restype = i
    lib.efe_env_render.argtypes = [p, f32p, f32p, C.c_void_p, C.c_int64, f32p, C.c_void_p, i, p]; lib.efe_env_render.restype = i
    lib.efe_check_reward.argtypes = [p, f32p, i, f32p, p]; lib.efe_check_reward.restype = i
    lib.efe_reparameterize.argtypes = [p, f32p, f32p, i, i, nzp, f32p, f32p, p]; lib.efe_reparameterize.restype = i
    tp = C.POINTER(EfeMctsTree)
    lib.efe_mcts_select.argtypes = [p, tp, p, C.c_float, i, i, p, p, p, p, f32p, f32p, p]; lib.efe_mcts_select.restype = i
    lib.efe_mcts_expand.argtypes = [p, tp, p, p, p, f32p, f32p, p]; lib.efe_mcts_expand.restype = i
    lib.efe_mcts_backprop.argtypes = [p, tp, p, p, p, p, p, f32p, i, f32p, i, f32p, p, p]; lib.efe_mcts_backprop.restype = i
    lib.efe_mcts_stop.argtypes = [p, tp, p, p, i, C.c_float, p, p]; lib.efe_mcts_stop.restype = i
    _lib = lib
    return lib
