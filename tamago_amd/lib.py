"""ctypes binding of libtamago_hip.so (the C ABI declared in include/tamago_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, the product
path raises.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64,
                    c_size_t, c_uint64, c_void_p)

HERE = os.path.dirname(os.path.abspath(__file__))
# (TAMAGO_HIP_LIB: another build of the library - kernel experiments under tools/experiments/_bin/; default: the in-tree build)
LIB_PATH = os.environ.get("TAMAGO_HIP_LIB") or os.path.join(HERE, "libtamago_hip.so")


class TamagoHipError(RuntimeError):
    pass


class SearchConfig(Structure):
    _fields_ = [("board_size", c_int32), ("num_trees", c_int32), ("tree_size", c_int32),
                ("batch_size", c_int32), ("cgos_mode", c_int32), ("check_superko", c_int32),
                ("device", c_int32), ("reserved", c_int32)]


class RootPosition(Structure):
    _fields_ = [("cells", c_void_p), ("hash_history", c_void_p), ("hash", c_uint64),
                ("moves", c_int32), ("ko_pos", c_int32), ("ko_move", c_int32),
                ("prev_move", c_int32), ("prev_prev_move", c_int32), ("to_move", c_int32)]


class SelfplayEvent(Structure):
    """tg_selfplay_event (include/tamago_hip.h): what tg_selfplay_play_move shows its observer."""
    _fields_ = [("kind", c_int32), ("phase", c_int32), ("trees", c_int32), ("positions", c_int32),
                ("num_considered", POINTER(c_int32)), ("max_count", POINTER(c_int32)),
                ("planes_dev", c_void_p), ("policy_dev", c_void_p), ("value_dev", c_void_p),
                ("stream", c_void_p),
                ("num_children", POINTER(c_int32)), ("action", POINTER(c_int32)),
                ("children_visits", POINTER(c_int32)), ("children_value_sum", POINTER(c_double)),
                ("moves", POINTER(c_int32)), ("finished", POINTER(c_int32))]


SELFPLAY_OBSERVER = ctypes.CFUNCTYPE(None, c_void_p, POINTER(SelfplayEvent))

_SIGNATURES = {
    "tg_abi_version": (c_int, []),
    "tg_last_error": (c_char_p, []),
    "tg_device_count": (c_int, [POINTER(c_int)]),
    "tg_net_param_count": (c_size_t, [c_int]),
    "tg_net_create": (c_int, [c_int, c_int, c_void_p, c_size_t, POINTER(c_void_p)]),
    "tg_net_destroy": (c_int, [c_void_p]),
    "tg_net_board_size": (c_int, [c_void_p]),
    "tg_net_forward_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "tg_net_forward_host": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "tg_net_profile_phases": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int]),
    "tg_net_kernel_name": (c_char_p, [c_void_p, c_int]),
    "tg_net_flops_per_position": (c_double, [c_int]),
    "tg_net_executed_flops_per_position": (c_double, [c_void_p, c_int, POINTER(c_double), POINTER(c_char_p)]),
    "tg_net_range_fallbacks": (c_int, [c_void_p, POINTER(ctypes.c_ulonglong)]),
    "tg_net_range_fallback_positions": (c_int, [c_void_p, POINTER(ctypes.c_ulonglong)]),
    "tg_net_band_timeouts": (c_int, [c_void_p, POINTER(ctypes.c_ulonglong)]),
    "tg_net_set_shared_device": (c_int, [c_void_p, c_int]),
    "tg_featurize_dev": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                 c_void_p]),
    "tg_featurize_sym_dev": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_void_p, c_void_p]),
    "tg_search_create": (c_int, [POINTER(SearchConfig), POINTER(c_void_p)]),
    "tg_search_destroy": (c_int, [c_void_p]),
    "tg_search_set_zobrist": (c_int, [c_void_p, c_void_p, c_size_t]),
    "tg_search_set_root": (c_int, [c_void_p, c_int, POINTER(RootPosition)]),
    "tg_search_set_rng": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t]),
    "tg_search_rng_consumed": (c_int, [c_void_p, c_void_p]),
    "tg_search_select_puct": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "tg_search_puct_chain": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tg_search_read_root_stats": (c_int, [c_void_p] + [c_void_p] * 8),
    "tg_search_profile": (c_int, [c_void_p, c_int, c_void_p]),
    "tg_search_play": (c_int, [c_void_p, c_void_p, c_void_p]),
    "tg_search_read_positions": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "tg_search_set_noise": (c_int, [c_void_p, c_void_p]),
    "tg_search_select_gumbel": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "tg_search_root_planes": (c_int, [c_void_p, c_void_p, c_void_p]),
    "tg_search_backup": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "tg_search_read_node": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 12),
    "tg_search_num_nodes": (c_int, [c_void_p, c_void_p]),
    "tg_search_read_queue": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "tg_search_grow": (c_int, [c_void_p, c_int]),
    "tg_search_read_roots": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "tg_search_read_path": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "tg_search_seed_stream": (c_int, [c_void_p, c_int, c_void_p, c_int]),
    "tg_search_stream_state": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "tg_search_feed_streams": (c_int, [c_void_p, c_size_t, c_int]),
    "tg_search_advance_streams": (c_int, [c_void_p, c_void_p]),
    "tg_search_draw_noise": (c_int, [c_void_p, c_void_p]),
    "tg_legacy_exponentials": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "tg_glibc_log": (c_int, [c_void_p, c_size_t, c_void_p]),
    "tg_search_node_record_num_nodes": (c_int, [c_void_p, POINTER(c_int32)]),
    "tg_search_own_stream": (c_int, [c_void_p, POINTER(c_void_p)]),
    "tg_search_debug_read_window": (c_int, [c_void_p, c_int, c_size_t, c_size_t, c_void_p]),
    "tg_search_debug_stream_walk": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64]),
    "tg_trainer_create": (c_int, [c_int, c_int, c_int, c_void_p, c_size_t, POINTER(c_void_p)]),
    "tg_trainer_destroy": (c_int, [c_void_p]),
    "tg_trainer_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p]),
    "tg_trainer_read_losses": (c_int, [c_void_p, c_void_p, c_int]),
    "tg_trainer_get_params": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "tg_trainer_debug_read": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "tg_trainer_set_momentum": (c_int, [c_void_p, c_void_p, c_size_t]),
    "tg_selfplay_create": (c_int, [c_void_p, c_char_p, c_int, c_double, c_char_p, POINTER(c_void_p)]),
    "tg_selfplay_destroy": (c_int, [c_void_p]),
    "tg_selfplay_start_game": (c_int, [c_void_p, c_int, c_int, c_int]),
    "tg_selfplay_schedule": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "tg_selfplay_finish_move": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "tg_selfplay_set_observer": (c_int, [c_void_p, SELFPLAY_OBSERVER, c_void_p]),
    "tg_selfplay_play_move": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tg_selfplay_move_begin": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tg_selfplay_move_end": (c_int, [c_void_p, c_void_p, c_void_p]),
}

_lib = None


def exported_symbols():
    """Names every build of the library must export (mirrors include/tamago_hip.h)."""
    return sorted(_SIGNATURES)


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime; it must be the one already in the process when our
    # library resolves libamdhip64, so that pointers, streams and events are shared
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise TamagoHipError(
            f"{LIB_PATH} is missing - build it with `python -m tamago_amd.build` "
            "(there is no CPU fallback for the product path)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().tg_last_error()
        raise TamagoHipError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
