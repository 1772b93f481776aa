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
           'efe_check_reward', 'efe_reparameterize', 'efe_mcts_select', 'efe_mcts_expand', 'efe_mcts_backprop', 'efe_mcts_stop',
           'efe_build_id', 'efe_reserve', 'efe_rollout_scratch_bytes', 'efe_arena_stats', 'efe_env_new_image', 'efe_create_cfg', 'efe_get_config', 'efe_get_device', 'efe_ctx_alive',
           'efe_calculate_g_rows', 'efe_simulate_rows', 'efe_mcts_step']
ABI_VERSION = 6


class EfeMctsTree(C.Structure):
    _fields_ = [('W', C.c_void_p), ('N', C.c_void_p), ('Qpi', C.c_void_p), ('child', C.c_void_p), ('S', C.c_void_p),
                ('E', C.c_int32), ('cap', C.c_int32), ('A', C.c_int32), ('s_dim', C.c_int32)]


class EfeNoise(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('stage', C.c_uint32), ('pass_', C.c_uint32), ('sample', C.c_uint32),
                ('row_offset', C.c_uint32)]


_lib = None
_ops = None
OPS_PATH = os.path.join(HERE, 'libefe_torch_ops.so')


def load_ops():
    """torch.ops.efe -- the custom-op registration library (csrc/torch_ops.cpp) over the C ABI.  Loud if missing or stale."""
    global _ops
    if _ops is not None:
        return _ops
    load()                                    # the engine library first (same checks), so the ops library resolves against it
    import torch
    from . import build as _build
    if not os.path.exists(OPS_PATH):
        raise ImportError(f'{OPS_PATH} not built: run `python -c "import __graft_entry__ as g; g.build()"`')
    want, have = _build.ops_digest(), _build._stamp(OPS_PATH, 'EFE_OPS_BUILD_ID')
    if want != have:
        raise ImportError(f'{OPS_PATH} is stale: built from sources {have}, the sources here are {want}; re-run build()')
    torch.ops.load_library(OPS_PATH)
    try:        # one engine copy in the process: the ops library must have resolved to the library ctypes loaded (A/B runs with EFE_LIB_PATH)
        mapped = {line.split()[-1] for line in open('/proc/self/maps') if 'libefe_mi355x' in line or os.path.basename(LIB_PATH) in line}
        engines = {m for m in mapped if os.path.realpath(m) != os.path.realpath(OPS_PATH)}
    except OSError:
        engines = set()
    if len({os.path.realpath(m) for m in engines}) > 1:
        raise ImportError(f'two engine libraries are mapped ({sorted(engines)}): libefe_torch_ops.so did not resolve to {LIB_PATH}; '
                          'rebuild the override with build.py (it sets the SONAME libefe_mi355x.so)')
    _ops = torch.ops.efe
    return _ops


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f'{LIB_PATH} not built: run `python -c "import __graft_entry__ as g; g.build()"` '
                          '(hipcc --offload-arch=gfx950); there is no CPU fallback')
    # RTLD_GLOBAL + the library's SONAME (build.py): when libefe_torch_ops.so is loaded later, its DT_NEEDED libefe_mi355x.so resolves
    # to THIS copy even if EFE_LIB_PATH points somewhere else -- the context and every hot-path op then run the same build
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    lib.efe_abi_version.argtypes = []; lib.efe_abi_version.restype = C.c_int
    if lib.efe_abi_version() != ABI_VERSION:
        raise ImportError(f'{LIB_PATH}: ABI version {lib.efe_abi_version()} != {ABI_VERSION} (stale build: re-run build())')
    lib.efe_build_id.argtypes = []; lib.efe_build_id.restype = C.c_char_p
    if not os.environ.get('EFE_LIB_PATH'):
        # a shipped binary must come from the sources next to it: compare the digest compiled into it
        from . import build as _build
        want, have = _build.source_digest(), lib.efe_build_id().decode()
        if want != have:
            raise ImportError(f'{LIB_PATH} is stale: built from sources {have}, the sources here are {want}; '
                              're-run `python -c "import __graft_entry__ as g; g.build()"`')
    p, i, f32p = C.c_void_p, C.c_int, C.c_void_p
    nzp = C.POINTER(EfeNoise)
    lib.efe_create.argtypes = [C.POINTER(p), i]; lib.efe_create.restype = i
    lib.efe_get_config.argtypes = [p, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i)]; lib.efe_get_config.restype = i
    lib.efe_get_device.argtypes = [p, C.POINTER(i), C.c_char_p, i]; lib.efe_get_device.restype = i
    lib.efe_create_cfg.argtypes = [C.POINTER(p), i, i, i, i, i]; lib.efe_create_cfg.restype = i
    lib.efe_destroy.argtypes = [p]; lib.efe_destroy.restype = None
    lib.efe_last_error.argtypes = [p]; lib.efe_last_error.restype = C.c_char_p
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
    lib.efe_simulate.argtypes = [p, f32p, i, i, i, nzp, f32p, f32p, f32p, f32p, f32p, p]; lib.efe_simulate.restype = i
    lib.efe_reserve.argtypes = [p, C.c_int64]; lib.efe_reserve.restype = i
    lib.efe_ctx_alive.argtypes = [p]; lib.efe_ctx_alive.restype = i
    lib.efe_rollout_scratch_bytes.argtypes = [p, i, i, i]; lib.efe_rollout_scratch_bytes.restype = C.c_int64
    lib.efe_arena_stats.argtypes = [p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]; lib.efe_arena_stats.restype = i
    lib.efe_env_new_image.argtypes = [p, f32p, i, nzp, p]; lib.efe_env_new_image.restype = i
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
    lib.efe_mcts_step.argtypes = [p, tp, p, p, f32p, i, f32p, f32p, p, p, p, i, C.c_float, p, C.c_float, i, i, p, p, p, p, f32p, f32p, p, f32p, f32p, p]; lib.efe_mcts_step.restype = i
    _lib = lib
    return lib
