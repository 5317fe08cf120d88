"""ctypes binding of the C-ABI in include/tsmarl.h (lib/libtsmarl_hip.so).

This is the only place the package touches native code.  There is NO fallback: if the HIP
library is missing or a call fails, the op raises (ImportError / RuntimeError / ValueError /
MalformedBufferError) -- nothing here ever routes through oracle/ or a CPU path.
PyTorch is used by callers only to own device memory and streams; this module sees raw pointers.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtsmarl_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "tsmarl.h")

MAX_GATHER_FIELDS = 12  # include/tsmarl.h TSM_MAX_GATHER_FIELDS
ABI_VERSION = 4  # include/tsmarl.h TSM_ABI_VERSION: bumped whenever a signature or a struct layout changes
TSM_OK, TSM_ERR_INVALID, TSM_ERR_HIP, TSM_ERR_MALFORMED_BUFFER, TSM_ERR_UNSUPPORTED = range(5)


class MalformedBufferError(RuntimeError):
    """Mirror of tianshou.data.buffer.buffer_base.MalformedBufferError (buffer_base.py:17-18)."""


class tsm_field(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_bytes", C.c_int64)]


class tsm_ppo_cfg(C.Structure):
    _fields_ = [("eps_clip", C.c_double), ("dual_clip", C.c_double), ("vf_coef", C.c_double),
                ("ent_coef", C.c_double), ("value_clip", C.c_int32), ("adv_norm", C.c_int32),
                ("loss_kind", C.c_int32), ("value_group", C.c_int32)]


class tsm_slab_seg(C.Structure):
    _fields_ = [("slabs", C.c_void_p), ("offset", C.c_int64), ("n", C.c_int64), ("stride", C.c_int64),
                ("n_slab", C.c_int32), ("frag_k1", C.c_int32), ("scale_dev", C.c_void_p), ("frag_image", C.c_void_p),
                ("frag_kj", C.c_int32), ("_pad", C.c_int32)]


class tsm_slab_reduce(C.Structure):
    _fields_ = [("slabs", C.c_void_p), ("n", C.c_int64), ("stride", C.c_int64), ("n_slab", C.c_int32), ("_pad", C.c_int32),
                ("out", C.c_void_p)]


class tsm_gather_field(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("n_rows", C.c_int64), ("T", C.c_int64), ("E", C.c_int64),
                ("src_row_stride", C.c_int64), ("src_offset", C.c_int64), ("width", C.c_int32), ("src_kind", C.c_int32),
                ("dst_kind", C.c_int32), ("_pad", C.c_int32)]


class tsm_mpe_cfg(C.Structure):
    _fields_ = [("n_env", C.c_int32), ("n_agent", C.c_int32), ("max_cycles", C.c_int32), ("_pad", C.c_int32),
                ("dt", C.c_double), ("damping", C.c_double), ("contact_force", C.c_double),
                ("contact_margin", C.c_double), ("agent_size", C.c_double), ("landmark_size", C.c_double),
                ("accel", C.c_double), ("max_speed", C.c_double), ("local_ratio", C.c_double)]


class tsm_rollout_desc(C.Structure):
    _fields_ = [("params", C.c_void_p), ("param_image", C.c_void_p),
                ("obs_dim", C.c_int32), ("hidden", C.c_int32), ("n_act", C.c_int32), ("mode", C.c_int32),
                ("policy_seed", C.c_uint64), ("offset", C.c_uint64), ("offset_dev", C.c_void_p),
                ("env", tsm_mpe_cfg), ("env_seed", C.c_uint64), ("episode_ctr", C.c_void_p),
                ("agent_pos", C.c_void_p), ("agent_vel", C.c_void_p), ("landmark_pos", C.c_void_p),
                ("steps", C.c_void_p), ("auto_reset", C.c_int32), ("n_steps", C.c_int32),
                ("obs_cur_out", C.c_void_p), ("vrb_state", C.c_void_p), ("sub_size", C.c_int64),
                ("done_store", C.c_void_p), ("obs_store", C.c_void_p), ("obs_next_store", C.c_void_p),
                ("rew_store", C.c_void_p), ("logp_store", C.c_void_p), ("vs_store", C.c_void_p),
                ("vnext_store", C.c_void_p), ("act_store", C.c_void_p), ("term_store", C.c_void_p),
                ("trunc_store", C.c_void_p), ("ptr_out", C.c_void_p), ("ep_rew_out", C.c_void_p),
                ("ep_len_out", C.c_void_p), ("ep_idx_out", C.c_void_p),
                ("ep_rec", C.c_void_p), ("max_ep", C.c_int32), ("_pad2", C.c_int32),
                ("offset_inc", C.c_uint64), ("done_ctr", C.c_void_p)]


class tsm_mpe_tag_cfg(C.Structure):
    _fields_ = [("n_env", C.c_int32), ("n_adv", C.c_int32), ("n_good", C.c_int32), ("n_obst", C.c_int32),
                ("max_cycles", C.c_int32), ("_pad", C.c_int32),
                ("dt", C.c_double), ("damping", C.c_double), ("contact_force", C.c_double), ("contact_margin", C.c_double),
                ("adv_size", C.c_double), ("good_size", C.c_double), ("obst_size", C.c_double),
                ("adv_accel", C.c_double), ("good_accel", C.c_double), ("adv_speed", C.c_double), ("good_speed", C.c_double)]


class tsm_rollout_tag_desc(C.Structure):
    _fields_ = [("params", C.c_void_p * 2), ("policy_seed", C.c_uint64 * 2), ("offset", C.c_uint64 * 2),
                ("mode", C.c_int32 * 2), ("obs_dim", C.c_int32), ("hidden", C.c_int32), ("n_act", C.c_int32),
                ("env_major_counter", C.c_int32), ("offset_dev", C.c_void_p), ("env", tsm_mpe_tag_cfg), ("env_seed", C.c_uint64),
                ("episode_ctr", C.c_void_p), ("agent_pos", C.c_void_p), ("agent_vel", C.c_void_p),
                ("landmark_pos", C.c_void_p), ("steps", C.c_void_p), ("auto_reset", C.c_int32), ("n_steps", C.c_int32),
                ("obs_cur_out", C.c_void_p), ("vrb_state", C.c_void_p), ("sub_size", C.c_int64),
                ("done_store", C.c_void_p), ("obs_store", C.c_void_p), ("obs_next_store", C.c_void_p),
                ("rew_store", C.c_void_p), ("logp_store", C.c_void_p), ("vs_store", C.c_void_p),
                ("act_store", C.c_void_p), ("term_store", C.c_void_p), ("trunc_store", C.c_void_p),
                ("ptr_out", C.c_void_p), ("ep_rew_out", C.c_void_p), ("ep_len_out", C.c_void_p),
                ("ep_idx_out", C.c_void_p), ("ep_rec", C.c_void_p), ("max_ep", C.c_int32), ("_pad2", C.c_int32),
                ("offset_inc", C.c_uint64), ("done_ctr", C.c_void_p), ("vnext_store", C.c_void_p)]


class tsm_mlp_desc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("act", C.c_int32), ("dims", C.c_int32 * 9)]


_p, _i64, _i32, _f64, _int, _u64 = C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_int, C.c_uint64

# name -> (restype, argtypes); must list every function declared in include/tsmarl.h
SIGNATURES = {
    "tsm_abi_version": (_int, []),
    "tsm_last_error": (C.c_char_p, []),
    "tsm_device_info": (_int, [C.POINTER(_int), C.POINTER(_int), C.POINTER(_i64), C.c_char_p]),
    "tsm_kernel_option_get": (_int, [C.c_char_p, C.POINTER(_i32)]),
    "tsm_kernel_option_set": (_int, [C.c_char_p, _i32]),
    "tsm_mem_alloc": (_int, [C.POINTER(_p), _i64]),
    "tsm_mem_free": (_int, [_p]),
    "tsm_mem_h2d": (_int, [_p, _p, _i64, _p]),
    "tsm_mem_d2h": (_int, [_p, _p, _i64, _p]),
    "tsm_mem_set": (_int, [_p, _int, _i64, _p]),
    "tsm_stream_sync": (_int, [_p]),
    "tsm_stream_abort_capture": (_int, [_p]),
    "tsm_gae_lanes": (_int, [_p, _p, _p, _p, _p, _int, _i64, _i64, _i64, _p, _p, _f64, _f64, _f64, _p, _p, _p]),
    "tsm_gae_scan_workspace_bytes": (_i64, []),
    "tsm_gae_set_scan_workspace": (_int, [_p, _i64]),
    "tsm_gae_set_scan_error_word": (_int, [_p]),
    "tsm_gae_lanes_rms": (_int, [_p, _p, _p, _p, _p, _int, _i64, _i64, _i64, _p, _p, _f64, _f64, _p, _f64, _p, _p, _p]),
    "tsm_rms_update_work_elems": (_i64, [_i64]),
    "tsm_rms_update": (_int, [_p, _p, _i64, _p, _f64, _p, _p]),
    "tsm_mc_return_to_go_lanes": (_int, [_p, _i64, _i64, _f64, _p, _p]),
    "tsm_vrb_state_bytes": (_i64, [_i64, _i64]),
    "tsm_vrb_init": (_int, [_p, _i64, _i64, _i64, _p]),
    "tsm_vrb_reset": (_int, [_p, _i64, _i64, _i64, _int, _p]),
    "tsm_vrb_add": (_int, [_p, _i64, _i64, _i64, _p, _i64, _p, _p, _p, C.POINTER(tsm_field), _int,
                           _p, _p, _p, _p, _p]),
    "tsm_vrb_check": (_int, [_p, _i64, _i64, _p]),
    "tsm_vrb_sample_indices_all": (_int, [_p, _i64, _i64, _p, _p, _p, _p]),
    "tsm_vrb_unfinished_index": (_int, [_p, _i64, _i64, _p, _p, _p, _p]),
    "tsm_vrb_prev": (_int, [_p, _i64, _i64, _p, _p, _i64, _p, _p]),
    "tsm_vrb_next": (_int, [_p, _i64, _i64, _p, _p, _i64, _p, _p]),
    "tsm_vrb_gather": (_int, [_p, _i64, _i64, _i64, _p, _i64, _p, _p]),
    "tsm_agent_index": (_int, [_p, _i64, _i32, _p, _p, _p, _p]),
    "tsm_scatter_rows": (_int, [_p, _p, _i64, _i64, _p, _p]),
    "tsm_gather_rows": (_int, [_p, _p, _i64, _i64, _p, _p]),
    "tsm_categorical_sample": (_int, [_p, _i64, _i32, _u64, _u64, _p, _int, _p, _p, _p]),
    "tsm_categorical_logp_entropy": (_int, [_p, _p, _i64, _i32, _p, _p, _p]),
    "tsm_ppo_adv_stats": (_int, [_p, _p, _p, _i32, _p, _p]),
    "tsm_ppo_adv_stats_work_elems": (_i64, [_i32, _i64]),
    "tsm_ppo_adv_stats_pack": (_int, [_p, _p, _i32, _p, _p]),
    "tsm_ppo_adv_stats_unpack": (_int, [_p, _i32, _p, _p]),
    "tsm_ppo_adv_stats_wide": (_int, [_p, _p, _p, _i32, _i64, _p, _p, _p]),
    "tsm_ppo_loss_partial_elems": (_i64, [_i64]),
    "tsm_ppo_loss_fwd_bwd": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _p,
                                    C.POINTER(tsm_ppo_cfg), _p, _p, _p, _p]),
    "tsm_ppo_loss_finalize": (_int, [_p, _i64, C.POINTER(tsm_ppo_cfg), _p, _p]),
    "tsm_adam_work_elems": (_i64, [_i64]),
    "tsm_adam_step": (_int, [_p, _p, _i32, _i64, _p, _p, _i64, _p, _f64, _p, _f64, _f64, _f64, _f64, _f64, _p, _p, _p, _p]),
    "tsm_scatter_image": (_int, [_p, _i64, _p, _p, _p]),
    "tsm_policy_image_elems": (_i64, [_i32, _i32, _i32]),
    "tsm_policy_image_map": (_int, [_i32, _i32, _i32, _p]),
    "tsm_reduce_slabs": (_int, [_p, _i32, _i64, _f64, _p, _p]),
    "tsm_global_state": (_int, [C.POINTER(_p), _i32, _i64, _i32, _int, _p, _p]),
    "tsm_random_permutations": (_int, [_i64, _i32, _u64, _u64, _p, _i64, _i32, _i64, _p, _p]),
    "tsm_random_permutations_advance": (_int, [_i64, _i32, _u64, _p, _u64, _p, _i64, _i32, _i64, _p, _p]),
    "tsm_ctde_head_partial_elems": (_i64, [_i64]),
    "tsm_ctde_td_head": (_int, [_p, _p, _i32, _p, _p, C.c_float, _p, _p, _i32, _i64, _p, _p, _p, _p, _p]),
    "tsm_mlp_param_count": (_i64, [C.POINTER(tsm_mlp_desc)]),
    "tsm_mlp_act_elems": (_i64, [C.POINTER(tsm_mlp_desc), _i64]),
    "tsm_mlp_forward": (_int, [C.POINTER(tsm_mlp_desc), _p, _p, _i64, _p, _p]),
    "tsm_mlp_forward_cond": (_int, [C.POINTER(tsm_mlp_desc), _p, _p, _i64, _p, _p, _p]),
    "tsm_any_nonzero_u8": (_int, [_p, _i64, _p, _p]),
    "tsm_gather_fields": (_int, [C.POINTER(tsm_gather_field), C.c_int32, _p]),
    "tsm_value_next_select": (_int, [_p, _p, _p, _p, _i64, _i64, _p, _p]),
    "tsm_value_next_index": (_int, [_p, _p, _i64, _i64, _i64, _p, _p]),
    "tsm_value_next_select_env_major": (_int, [_p, _p, _p, _p, _i64, _i64, _i64, _p, _p]),
    "tsm_mlp_backward": (_int, [C.POINTER(tsm_mlp_desc), _p, _p, _i64, _p, _p, _p, _i32, _p, _i64, _p]),
    "tsm_policy_param_count": (_i64, [_i32, _i32, _i32]),
    "tsm_policy_forward": (_int, [_p, _p, _i32, _i32, _i32, _p, _i64, _int, _u64, _u64, _p, _p, _p, _p, _p, _p]),
    "tsm_mpe_spread_reset": (_int, [C.POINTER(tsm_mpe_cfg), _u64, _p, _p, _i64, _p, _p, _p, _p, _p, _p]),
    "tsm_mpe_spread_step": (_int, [C.POINTER(tsm_mpe_cfg), _u64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                   _int, _p, _u64, _p]),
    "tsm_rollout_spread": (_int, [C.POINTER(tsm_rollout_desc), _p]),
    "tsm_rollout_spread_actor": (_int, [C.POINTER(tsm_rollout_desc), _p]),
    "tsm_rollout_tag": (_int, [C.POINTER(tsm_rollout_tag_desc), _p]),
    "tsm_mpe_tag_obs_dim": (_int, [C.POINTER(tsm_mpe_tag_cfg)]),
    "tsm_mpe_tag_reset": (_int, [C.POINTER(tsm_mpe_tag_cfg), _u64, _p, _p, _i64, _p, _p, _p, _p, _p, _p]),
    "tsm_mpe_tag_step": (_int, [C.POINTER(tsm_mpe_tag_cfg), _u64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                _int, _p, _u64, _p]),
    "tsm_u64_add": (_int, [_p, _u64, _p]),
    "tsm_ppo_critic_rows_supported": (_int, [_i32, _i32, _i32]),
    "tsm_ppo_critic_rows_param_count": (_i64, [_i32, _i32]),
    "tsm_ppo_critic_rows_grid": (_int, [_i64]),
    "tsm_ppo_critic_rows_update": (_int, [_p, _i32, _i32, _i32, _p, _p, _p, _p, _i64, _i64, C.POINTER(tsm_ppo_cfg), _i32, _p, _p, _p]),
    "tsm_ppo_actor_rows_supported": (_int, [_i32, _i32, _i32]),
    "tsm_ppo_rows_init": (_int, []),
    "tsm_ppo_actor_rows_param_count": (_i64, [_i32, _i32, _i32]),
    "tsm_ppo_actor_rows_grid": (_int, [_i64]),
    "tsm_ppo_actor_rows_update": (_int, [_p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _i64, _i64, _p, C.POINTER(tsm_ppo_cfg),
                                         _i32, _p, _p, _p, _p]),
    "tsm_critic_rows_forward_supported": (_int, [_i32, _i32]),
    "tsm_critic_rows_init": (_int, [_i32, _i32]),
    "tsm_critic_rows_forward": (_int, [_p, _i32, _i32, _i32, _p, _p, _i64, _i64, _p, _p, _p]),
    "tsm_critic_rows_param_count": (_i64, [_i32, _i32, _i32]),
    "tsm_critic_rows_grad_grid": (_int, [_i64, _i32]),
    "tsm_critic_rows_w1_image_kj": (_int, [_i32]),
    "tsm_critic_rows_w1_image_elems": (_i64, [_i32]),
    "tsm_critic_rows_w1_image": (_int, [_p, _i32, _p, _p]),
    "tsm_critic_rows_grad_ppo": (_int, [_p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _i64, _i64, C.POINTER(tsm_ppo_cfg), _i32, _p, _p, _p,
                                        _p, _p, _p]),
    "tsm_critic_rows_grad_td": (_int, [_p, _p, _i32, _i32, _i32, _p, _i64, _i64, _p, _p, _i64, _i64, _p, _p, _p, _f64, _i32, _p, _p,
                                       _p, _p]),
    "tsm_critic_rows_dw1_chunks": (_int, [_i64, _i32]),
    "tsm_ctde_finalize": (_int, [_p, _i32, _p, _i32, _i64, _p, _p, _p]),
    "tsm_critic_rows_dw1": (_int, [_p, _p, _i32, _p, _i64, _i64, _i64, _i64, _i32, _p, _p, _p, _p, C.POINTER(tsm_slab_reduce), _i32, _p]),
    "tsm_p2p_ipc_handle_bytes": (_i64, []),
    "tsm_p2p_create": (_int, [_i32, _i32, _i64, C.POINTER(_p)]),
    "tsm_p2p_export": (_int, [_p, _p]),
    "tsm_p2p_import": (_int, [_p, _i32, _p]),
    "tsm_p2p_all_reduce": (_int, [_p, _p, _i64, _p]),
    "tsm_p2p_adam_step": (_int, [_p, _p, _p, _i32, _i64, _p, _p, _i64, _p, _f64, _p, _f64, _f64, _f64, _f64, _p, _p, _p]),
    "tsm_p2p_set_timeout": (_int, [_p, _f64]),
    "tsm_p2p_handshake": (_int, [_p, C.POINTER(_i32), _p]),
    "tsm_p2p_failed": (_int, [_p]),
    "tsm_p2p_error_async": (_int, [_p, _p, _p]),
    "tsm_p2p_destroy": (_int, [_p]),
    "tsm_reduce_slabs_segs": (_int, [_p, _i32, _i64, _f64, _p, _p]),
    "tsm_adam_step_segs": (_int, [_p, _p, _i32, _i64, _p, _p, _i64, _p, _f64, _p, _f64, _f64, _f64, _f64, _f64, _p, _p]),
    "tsm_ppo_update_grid": (_int, [_i64, _i32]),
    "tsm_ppo_finalize_many": (_int, [_p, _i64, _p, _p, _i32, C.POINTER(tsm_ppo_cfg), _p, _p]),
    "tsm_ppo_update_fused": (_int, [_p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _p,
                                    C.POINTER(tsm_ppo_cfg), _i32, _p, _p, _p, _p, _p]),
}

_NO_STATUS = {"tsm_critic_rows_w1_image_kj", "tsm_critic_rows_w1_image_elems", "tsm_p2p_ipc_handle_bytes", "tsm_p2p_failed", "tsm_critic_rows_forward_supported", "tsm_critic_rows_param_count", "tsm_critic_rows_grad_grid", "tsm_critic_rows_dw1_chunks", "tsm_ppo_critic_rows_supported", "tsm_ppo_critic_rows_param_count", "tsm_ppo_critic_rows_grid",
              "tsm_ppo_actor_rows_supported", "tsm_ppo_actor_rows_param_count", "tsm_ppo_actor_rows_grid",
              "tsm_rms_update_work_elems", "tsm_ppo_adv_stats_work_elems", "tsm_abi_version", "tsm_last_error", "tsm_stream_abort_capture", "tsm_vrb_state_bytes", "tsm_ppo_loss_partial_elems",
              "tsm_policy_param_count", "tsm_ppo_update_grid", "tsm_adam_work_elems", "tsm_policy_image_elems",
              "tsm_mlp_param_count", "tsm_mlp_act_elems", "tsm_ctde_head_partial_elems", "tsm_mpe_tag_obs_dim",
              "tsm_gae_scan_workspace_bytes"}

_lib = None


def header_symbols() -> list[str]:
    """Every function name declared in include/tsmarl.h."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tsm_[a-z0-9_]+)\s*\(", txt)))


def load() -> C.CDLL:
    """Load the HIP library (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m tianshou_marl_amd._build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.tsm_abi_version() != ABI_VERSION:
        raise ImportError(f"ABI version mismatch: library reports {lib.tsm_abi_version()}")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc == TSM_OK:
        return
    msg = load().tsm_last_error().decode("utf-8", "replace")
    if rc == TSM_ERR_INVALID:
        raise ValueError(msg)
    if rc == TSM_ERR_MALFORMED_BUFFER:
        raise MalformedBufferError(msg)
    raise RuntimeError(f"tsmarl error {rc}: {msg}")


def call(name: str, *args):
    """Call an ABI function; status-returning functions raise on failure."""
    fn = getattr(load(), name)
    out = fn(*args)
    if name in _NO_STATUS:
        return out
    check(out)
    return None


def ptr(t) -> int | None:
    """Device pointer of a torch tensor (None -> NULL).  Requires a contiguous CUDA/HIP tensor."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("tianshou_marl_amd ops need device (HIP) tensors; there is no CPU path")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return t.data_ptr()


_raw_stream = None


def stream_ptr() -> int:
    """hipStream_t of torch's current stream on the current device (the stream every op launches on; inside a capture, the
    capturing stream).  Asked of torch's C layer directly: `torch.cuda.current_stream().cuda_stream` builds a Stream object and
    resolves the device through four Python layers -- 4 us a call, six calls per step of the tag job (profiles/r05_host_tag.txt)."""
    global _raw_stream
    import torch

    if _raw_stream is None:
        get, dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        if get is None or dev is None:
            _raw_stream = False
        else:
            torch.cuda.init()
            _raw_stream = (get, dev)
    if _raw_stream is False:
        return torch.cuda.current_stream().cuda_stream
    get, dev = _raw_stream
    return get(dev())
