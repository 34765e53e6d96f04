"""ctypes binding of libdrt_hip.so (C ABI in include/drt_hip.h).

There is no CPU or PyTorch fallback: if the HIP library has not been built, or no
GPU is visible, the calls raise.  Build with ``python __graft_entry__.py`` (or
``make -C drt_amd/csrc``).
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DRT_HIP_LIB") or os.path.join(_HERE, "libdrt_hip.so")   # override: measurement variants only

_c = ctypes
_P = _c.c_void_p
_I64 = _c.c_int64
_D = _c.c_double

# name -> (restype, argtypes); must list every function declared in include/drt_hip.h
SIGNATURES = {
    "drt_last_error": (_c.c_char_p, []),
    "drt_version": (_c.c_int, []),
    "drt_deterministic": (_c.c_int, [_c.c_int]),
    "drt_fx_finalize": (_c.c_int, [_P, _I64, _P, _c.c_int, _P]),
    "drt_fx_add": (_c.c_int, [_P, _P, _I64, _P]),
    "drt_fx_to_limbs": (_c.c_int, [_P, _I64, _P, _P]),
    "drt_fx_from_limbs": (_c.c_int, [_P, _I64, _P, _P]),
    "drt_create": (_c.c_int, [_c.c_int, _c.POINTER(_P)]),
    "drt_destroy": (None, [_P]),
    "drt_update_mesh": (_c.c_int, [_P, _P, _I64, _P, _I64, _P]),
    "drt_update_vert": (_c.c_int, [_P, _P, _I64, _P]),
    "drt_update_vert_f64": (_c.c_int, [_P, _P, _I64, _P]),
    "drt_intersect": (_c.c_int, [_P, _P, _I64, _P, _P, _P]),
    "drt_intersect_any": (_c.c_int, [_P, _P, _I64, _P, _P]),
    "drt_intersect_bruteforce": (_c.c_int, [_P, _P, _I64, _P, _P, _P]),
    "drt_bvh_check": (_c.c_int, [_P, _P, _c.POINTER(_I64), _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int32)]),
    "drt_bvh_sorted_faces": (_c.c_int, [_P, _P, _P]),
    "drt_build_params": (_c.c_int, [_P, _c.POINTER(_c.c_float), _P]),
    "drt_tree_mode": (_c.c_int, [_P, _c.c_int, _c.c_int]),
    "drt_render_forward": (_c.c_int, [_P, _P, _P, _P, _I64, _D, _D, _P, _P, _P, _P, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _P]),
    "drt_render_backward": (_c.c_int, [_P, _P, _P, _P, _I64, _D, _D, _P, _P, _P, _P, _P, _P, _P, _P]),
    "drt_ray_loss": (_c.c_int, [_P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P]),
    "drt_ray_loss_listed": (_c.c_int, [_P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P]),
    "drt_prefill_zero": (_c.c_int, [_P, _P, _I64, _P]),
    "drt_prefill_wait": (_c.c_int, [_P, _P]),
    "drt_outputs_clean": (_c.c_int, [_P, _P, _P, _P, _I64, _P, _P, _P]),
    "drt_outputs_cancel": (_c.c_int, [_P]),
    "drt_render_seed": (_c.c_int, [_P, _P, _I64]),
    "drt_ray_loss_listed_grad": (_c.c_int, [_P, _P, _P, _P, _I64, _D, _D, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "drt_ray_loss_listed_grad_split": (_c.c_int, [_P, _P, _P, _P, _I64, _D, _D, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "drt_scale_rows3": (_c.c_int, [_P, _P, _P, _P, _P]),
    "drt_render_backward_ray_loss": (_c.c_int, [_P, _P, _P, _P, _I64, _D, _D, _P, _P, _P, _P, _P, _P, _P, _P]),
    "drt_render_ray_loss_fused": (_c.c_int, [_P, _P, _P, _P, _P, _P, _I64, _D, _D, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _P]),
    "drt_dihedral_forward": (_c.c_int, [_P, _P, _I64, _P, _P]),
    "drt_dihedral_backward": (_c.c_int, [_P, _P, _I64, _P, _P, _P]),
    "drt_sm_loss_fused": (_c.c_int, [_P, _P, _I64, _P, _P, _P]),
    "drt_silhouette_flags": (_c.c_int, [_P, _P, _I64, _P, _P, _P]),
    "drt_edge_sample_forward": (_c.c_int, [_P, _P, _P, _I64, _P, _P, _P, _P, _P, _c.c_int, _c.c_int, _P, _P]),
    "drt_edge_sample_backward": (_c.c_int, [_P, _P, _I64, _P, _P, _P, _c.c_int, _P, _P]),
    "drt_edge_sample_backward_rows": (_c.c_int, [_P, _P, _I64, _P, _P, _P, _I64, _P, _c.c_int, _P, _P]),
    "drt_vh_term": (_c.c_int, [_P, _P, _I64, _P, _c.c_int, _c.c_int, _P, _P, _P]),
    "drt_edge_sample_backward_term": (_c.c_int, [_P, _P, _I64, _P, _P, _P, _P, _c.c_int, _P, _P]),
    "drt_edge_tables_workspace": (_I64, [_I64]),
    "drt_edge_tables": (_c.c_int, [_P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "drt_subdivide_midpoint": (_c.c_int, [_P, _I64, _P, _I64, _P, _I64, _P, _c.c_int, _P, _P, _P]),
    "drt_limit_sgd_step": (_c.c_int, [_P, _P, _P, _I64, _D, _D, _c.c_int, _c.c_int, _D, _P]),
    "drt_limit_sgd_step3": (_c.c_int, [_P, _P, _P, _I64, _D, _D, _c.c_int, _c.c_int, _D, _P, _P, _P, _P, _P]),
    "drt_internal_stream": (_c.c_int, [_P, _c.c_int, _c.POINTER(_P)]),
    "drt_closest_point": (_c.c_int, [_P, _P, _I64, _P, _P, _P, _P]),
    "drt_vh_loss_fused": (_c.c_int, [_P, _P, _P, _P, _I64, _c.c_int, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P]),
    "drt_remesh_isotropic": (_c.c_int, [_P, _I64, _P, _I64, _D, _c.c_int, _D, _c.c_uint, _c.POINTER(_P)]),
    "drt_rm_split_mark": (_c.c_int, [_P, _I64, _P, _D, _P, _P]),
    "drt_rm_split_plan": (_c.c_int, [_P, _I64, _P, _P, _P, _P, _I64, _P, _P, _P]),
    "drt_rm_split_faces": (_c.c_int, [_P, _I64, _P, _P, _P, _P, _P]),
    "drt_rm_vertex_normals": (_c.c_int, [_P, _P, _P, _P, _I64, _P, _P]),
    "drt_rm_vertex_faces": (_c.c_int, [_P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "drt_rm_collapse_eval_all": (_c.c_int, [_P, _I64, _P, _P, _P, _P, _D, _D, _c.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P]),
    "drt_rm_surface_filter_list": (_c.c_int, [_P, _P, _P, _P, _P, _I64, _D, _P, _P]),
    "drt_rm_surface_filter": (_c.c_int, [_P, _P, _P, _P, _I64, _c.c_int, _D, _P, _P]),
    "drt_rm_closest_near": (_c.c_int, [_P, _P, _I64, _D, _P, _P]),
    "drt_rm_kill_faces": (_c.c_int, [_P, _P, _I64, _P, _P]),
    "drt_rm_collapse_apply": (_c.c_int, [_P, _I64, _P, _P, _P, _P, _P, _P, _I64, _D, _c.c_uint32, _c.c_int, _P, _P, _P, _P, _P, _c.c_int, _P, _P, _P]),
    "drt_rm_round_end": (_c.c_int, [_P, _c.c_int, _P]),
    "drt_rm_flip_eval": (_c.c_int, [_P, _I64, _P, _P, _P, _P, _D, _P, _P, _P, _P, _P]),
    "drt_rm_flip_apply": (_c.c_int, [_I64, _P, _P, _P, _I64, _c.c_int, _P, _P, _c.c_int, _P, _P, _P]),
    "drt_rm_smooth_target": (_c.c_int, [_P, _P, _P, _P, _I64, _P, _P]),
    "drt_rm_face_agreement": (_c.c_int, [_P, _P, _P, _I64, _P, _P]),
    "drt_rm_move_check": (_c.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P]),
    "drt_mesh_buf_size": (_c.c_int, [_P, _c.POINTER(_I64), _c.POINTER(_I64), _P]),
    "drt_mesh_buf_copy": (_c.c_int, [_P, _P, _P]),
    "drt_mesh_buf_free": (None, [_P]),
    "drt_profile_enable": (_c.c_int, [_P, _c.c_int]),
    "drt_profile_select": (_c.c_int, [_P, _c.c_uint32]),
    "drt_profile_read": (_c.c_int, [_P, _P, _P, _P]),
    "drt_profile_trace_stats": (_c.c_int, [_P, _P]),
    "drt_check_violations": (_c.c_int, [_P]),
}

_lib = None


class DrtError(RuntimeError):
    pass


def lib():
    """The loaded library; raises DrtError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DrtError(f"{LIB_PATH} not found: the HIP extension is not built "
                           "(run `python __graft_entry__.py`); there is no CPU fallback")
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("DRT_HIP_LIB") and not hasattr(h, name):
                continue                        # an OLDER build selected for an A/B run (tools/ab.sh): entry points added since are simply absent
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc):
    if rc != 0:
        raise DrtError(f"libdrt_hip error {rc}: {lib().drt_last_error().decode(errors='replace')}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
