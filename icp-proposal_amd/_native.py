"""ctypes binding of libicp_proposal_amd.so — the C ABI declared in include/icp_proposal.h.

There is deliberately no fallback: if the HIP library is missing or no GPU is usable, loading / context
creation raises.  Nothing under oracle/ is ever imported from here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ICP_LIBRARY_PATH: a test process may ask for the build with the test hooks compiled in (libicp_proposal_amd_testhooks.so)
LIB_PATH = os.environ.get("ICP_LIBRARY_PATH") or os.path.join(_HERE, "libicp_proposal_amd.so")

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int32)
c_ubyte_p = C.POINTER(C.c_uint8)


class ModelDesc(C.Structure):
    _fields_ = [("n_points", C.c_int32), ("n_triangles", C.c_int32), ("rank", C.c_int32),
                ("ref_points", c_double_p), ("mean_deformation", c_double_p), ("basis", c_double_p),
                ("variance", c_double_p), ("triangles", c_int_p)]


class MeshDesc(C.Structure):
    _fields_ = [("n_points", C.c_int32), ("n_triangles", C.c_int32), ("points", c_double_p), ("triangles", c_int_p)]


class ProposalParams(C.Structure):
    _fields_ = [("step_length", C.c_double), ("tangential_noise", C.c_double), ("noise_along_normal", C.c_double),
                ("direction", C.c_int32), ("boundary_aware", C.c_int32), ("n_model_ids", C.c_int32),
                ("n_target_points", C.c_int32), ("target_points", c_double_p)]


class EvaluatorParams(C.Structure):
    _fields_ = [("kind", C.c_int32), ("mode", C.c_int32), ("n_model_ids", C.c_int32), ("n_target_points", C.c_int32),
                ("target_points", c_double_p), ("gauss_mean", C.c_double), ("gauss_sigma", C.c_double),
                ("exp_rate", C.c_double)]


class PosteriorView(C.Structure):
    _fields_ = [("n_candidates", C.c_int32), ("corr_id", c_int_p), ("corr_aux", c_int_p), ("corr_point", c_double_p),
                ("keep", c_ubyte_p), ("alpha", c_double_p), ("M", c_double_p), ("V", c_double_p), ("S", c_double_p)]


class FitParams(C.Structure):
    _fields_ = [("direction", C.c_int32), ("n_model_ids", C.c_int32), ("model_ids", c_int_p), ("n_target_points", C.c_int32),
                ("target_points", c_double_p), ("step_length", C.c_double)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 40), ("calls", C.c_int64), ("total_ms", C.c_double), ("min_ms", C.c_double),
                ("max_ms", C.c_double)]


class MhMixture(C.Structure):
    _fields_ = [("struct_size", C.c_uint64), ("icp_weight", C.c_double * 2), ("w_icp", C.c_double), ("w_rw", C.c_double), ("rw_sigma", C.c_double),
                ("w_pose", C.c_double), ("pose_rot_sigma", C.c_double * 3), ("pose_trans_sigma", C.c_double * 3)]


class RuntimeStats(C.Structure):
    _fields_ = [("wait_timeouts", C.c_int64), ("speculation_giveups", C.c_int64), ("pipeline_fallbacks", C.c_int64),
                ("step_redos", C.c_int64), ("gate_timeouts", C.c_int64), ("reserved", C.c_int64 * 3)]


# every symbol include/icp_proposal.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "icp_ctx_create": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(MeshDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "icp_ctx_create_keyed": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(MeshDesc), C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]),
    "icp_ctx_destroy": (None, [C.c_void_p]),
    "icp_ctx_expect": (C.c_int, [C.c_int, C.c_int32]),
    "icp_ctx_set_target": (C.c_int, [C.c_void_p, C.POINTER(MeshDesc)]),
    "icp_status_string": (C.c_char_p, [C.c_int]),
    "icp_last_error": (C.c_char_p, []),
    "icp_ctx_rank": (C.c_int, [C.c_void_p]),
    "icp_ctx_device": (C.c_int, [C.c_void_p]),
    "icp_transformed_mesh": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "icp_vertex_normals": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "icp_closest_point_on_target": (C.c_int, [C.c_void_p, C.c_int32, c_double_p, c_double_p, c_int_p, c_double_p]),
    "icp_closest_target_vertex": (C.c_int, [C.c_void_p, C.c_int32, c_double_p, c_int_p, c_double_p]),
    "icp_closest_model_vertex": (C.c_int, [C.c_void_p, c_double_p, C.c_int32, c_double_p, c_int_p, c_double_p]),
    "icp_closest_point_on_model": (C.c_int, [C.c_void_p, c_double_p, C.c_int32, c_double_p, c_double_p, c_int_p, c_double_p]),
    "icp_proposal_create": (C.c_int, [C.c_void_p, C.POINTER(ProposalParams), C.POINTER(C.c_void_p)]),
    "icp_proposal_destroy": (None, [C.c_void_p]),
    "icp_proposal_num_candidates": (C.c_int, [C.c_void_p]),
    "icp_proposal_set_sampler": (C.c_int, [C.c_void_p, C.c_int32]),
    "icp_proposal_propose": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_int_p]),
    "icp_proposal_log_transition": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p]),
    "icp_proposal_posterior": (C.c_int, [C.c_void_p, c_double_p, C.POINTER(PosteriorView)]),
    "icp_evaluator_create": (C.c_int, [C.c_void_p, C.POINTER(EvaluatorParams), C.POINTER(C.c_void_p)]),
    "icp_evaluator_destroy": (None, [C.c_void_p]),
    "icp_evaluator_log_value": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p]),
    "icp_prior_log_value": (C.c_int, [C.c_int32, c_double_p, c_double_p]),
    "icp_ctx_profile_start": (C.c_int, [C.c_void_p, C.c_int32]),
    "icp_ctx_profile_search_counters": (C.c_int, [C.c_void_p, C.c_int32]),
    "icp_ctx_set_idle_hook": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "icp_chain_step_prelaunch": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "icp_ctx_profile_stop": (C.c_int, [C.c_void_p, C.POINTER(KernelStat), C.c_int32, C.POINTER(C.c_int32)]),
    "icp_chain_eval_step": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), c_double_p, c_double_p, c_double_p,
                                      c_double_p, c_double_p]),
    "icp_fit_deterministic": (C.c_int, [C.c_void_p, C.POINTER(FitParams), c_double_p, C.c_int32, C.c_int32, c_double_p, c_double_p]),
    "icp_posterior_variability": (C.c_int, [C.c_void_p, C.c_int32, c_double_p, C.c_int32, c_double_p, c_double_p]),
    "icp_mesh_metrics": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "icp_chain_step": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_int32, c_double_p, c_double_p, c_double_p,
                                 c_double_p, c_double_p, c_double_p]),
    "icp_chain_step_batched_issue": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                               C.POINTER(c_double_p), C.POINTER(c_double_p), C.POINTER(c_double_p), c_double_p,
                                               c_double_p, c_double_p, C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_void_p)]),
    "icp_chain_step_batched_collect": (C.c_int, [C.c_void_p]),
    "icp_chain_step_batched_abandon": (C.c_int, [C.c_void_p]),
    "icp_ctx_set_rotation": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "icp_ctx_rotation_convention": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "icp_ctx_runtime_stats": (C.c_int, [C.c_void_p, C.POINTER(RuntimeStats)]),
    "icp_ctx_step_paths": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "icp_chain_step_path": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "icp_proposal_basis_state": (C.c_int, [C.c_void_p, c_double_p]),
    "icp_chain_bind": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "icp_chain_bind_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "icp_release_cached_models": (None, []),
    "icp_chains_run_on_device": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.POINTER(MhMixture),
                                           C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(c_double_p), c_double_p, C.c_int32,
                                           C.POINTER(c_double_p), C.POINTER(C.c_int64)]),
    "icp_chain_step_batched": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                         C.POINTER(c_double_p), C.POINTER(c_double_p), C.POINTER(c_double_p), c_double_p, c_double_p,
                                         c_double_p, C.POINTER(C.c_int32)]),
}

_LIB = None


class IcpNativeError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__(f"{where}: status {status} ({detail})")


def lib():
    """Load libicp_proposal_amd.so (built by __graft_entry__.build() / csrc/Makefile).  Raises if absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise IcpNativeError(-2, "load", f"{LIB_PATH} not built — run `python -c 'import __graft_entry__ as g; g.build()'`; "
                                             "there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(status, where):
    if status != 0:
        L = lib()
        detail = (L.icp_last_error() or b"").decode() or (L.icp_status_string(status) or b"").decode()
        raise IcpNativeError(status, where, detail)


def step_paths(ctx_handle=None) -> dict:
    """icp_ctx_step_paths: how many chain steps took which path (of one context, or of the process)."""
    out = (C.c_int64 * 4)()
    check(lib().icp_ctx_step_paths(ctx_handle, out), "icp_ctx_step_paths")
    return {"merged": int(out[0]), "wide": int(out[1]), "per_stage": int(out[2]), "device_loop": int(out[3])}


def runtime_stats(ctx_handle=None) -> dict:
    """icp_ctx_runtime_stats: fall-back counters of one context (its handle) or, with None, of the whole process.  A normal run
    shows zero everywhere."""
    st = RuntimeStats()
    check(lib().icp_ctx_runtime_stats(ctx_handle, C.byref(st)), "icp_ctx_runtime_stats")
    return {k: int(getattr(st, k)) for k, _ in RuntimeStats._fields_ if k != "reserved"}
