"""ctypes binding of include/acmpc.h (the C-ABI drop-in boundary).

The shared library is hand-written HIP for gfx950; there is no CPU implementation behind these calls.  A
missing library or a missing GPU raises (`EngineError`), it never degrades to another path.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import numpy as np

from . import _build

OK, EINVAL, EHIP, ENODEVICE, ECAPACITY, ESTATE = 0, -1, -2, -3, -4, -5
MODE_SPATIAL, MODE_TEMPORAL = 0, 1
LAYOUT_CANDIDATE_MAJOR, LAYOUT_STEP_MAJOR = 0, 1
COEF_STRIDE = {MODE_SPATIAL: 12, MODE_TEMPORAL: 8}
REC_COST, REC_VIOLATION, REC_NFEASIBLE, REC_OWNER, REC_HEADER = 0, 1, 2, 3, 4


class EngineError(RuntimeError):
    """A call into libacmpc_hip failed; `.code` is the ACMPC_E* value."""

    def __init__(self, code: int, message: str):
        super().__init__("acmpc error %d: %s" % (code, message))
        self.code = code


class Params(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("mode", C.c_int32),
        ("device", C.c_int32),
        ("max_problems", C.c_int32),
        ("max_candidates", C.c_int32),
        ("max_steps", C.c_int32),
        ("nn_back", C.c_int32),
        ("nn_ahead", C.c_int32),
        ("centre_update", C.c_int32),
        ("lq_candidate", C.c_int32),
        ("step_cost", C.c_double * 3),
        ("r_term", C.c_double * 2),
        ("final_cost", C.c_double * 3),
        ("u_min", C.c_double * 2),
        ("u_max", C.c_double * 2),
        ("margin", C.c_double),
        ("wheelbase", C.c_double),
        ("t_min", C.c_double),
        ("dt", C.c_double),
        ("w_bound", C.c_double),
        ("softmin_lambda", C.c_double),
    ]


class PfParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("max_particles", C.c_int32),
        ("max_observation_points", C.c_int32),
        ("score_mean", C.c_double),
        ("score_sigma", C.c_double),
        ("threshold_rotation", C.c_double),
        ("threshold_offset", C.c_double),
        ("threshold_error", C.c_double),
        ("wheelbase", C.c_double),
    ]


class Tick(C.Structure):
    """acmpc_tick: the per-tick inputs of acmpc_control_tick."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("horizon", C.c_int32),
        ("localised", C.c_int32),
        ("has_end_velocity", C.c_int32),
        ("n_candidates", C.c_int32),
        ("rounds", C.c_int32),
        ("centre_is_reference", C.c_int32),
        ("qp_max_iter", C.c_int32),
        ("qp_check_every", C.c_int32),
        ("qp_method", C.c_int32),
        ("offset", C.c_double),
        ("v_min", C.c_double),
        ("v_max", C.c_double),
        ("a_min", C.c_double),
        ("a_max", C.c_double),
        ("ay_max", C.c_double),
        ("ki_min", C.c_double),
        ("end_velocity", C.c_double),
        ("sigma", C.c_double * 2),
        ("shrink", C.c_double),
        ("qp_eps_abs", C.c_double),
        ("qp_eps_rel", C.c_double),
        ("seed", C.c_uint64),
        ("map_index", C.c_int32),
        ("centreline_points", C.c_int32),
        ("pose_x", C.c_double),
        ("pose_y", C.c_double),
        ("lateral_offset", C.c_double),
    ]


class PfResample(C.Structure):
    """acmpc_pf_resample: the resampling parameters of one device-resident filter update."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_desired", C.c_int32),
        ("minimum_particles", C.c_int32),
        ("counter", C.c_uint32),
        ("seed", C.c_uint64),
        ("sigma_x", C.c_double),
        ("sigma_y", C.c_double),
        ("sigma_yaw", C.c_double),
    ]


_F32P = C.POINTER(C.c_float)
_F64P = C.POINTER(C.c_double)
_I32P = C.POINTER(C.c_int32)
_I64P = C.POINTER(C.c_int64)
_CTX = C.c_void_p

# name -> (restype, argtypes): every symbol include/acmpc.h declares
SIGNATURES = {
    "acmpc_version": (C.c_char_p, []),
    "acmpc_create": (C.c_int, [C.POINTER(Params), C.POINTER(_CTX)]),
    "acmpc_destroy": (None, [_CTX]),
    "acmpc_set_option": (C.c_int, [_CTX, C.c_char_p, C.c_char_p]),
    "acmpc_lq_plan": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
    "acmpc_lq_box_plan": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "acmpc_lq_box_stats": (C.c_int, [_CTX, C.c_void_p]),
    "acmpc_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "acmpc_rccl_comm_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "acmpc_rccl_comm_destroy": (C.c_int, [C.c_void_p]),
    "acmpc_rollout_start_clocks": (C.c_int, [_CTX, C.c_void_p, C.c_int32, _I32P]),
    "acmpc_last_error": (C.c_char_p, [_CTX]),
    "acmpc_set_paths": (C.c_int, [_CTX, C.c_void_p, C.c_int32, C.c_int32]),
    "acmpc_get_coefficients": (C.c_int, [_CTX, C.c_int32, _F32P, C.c_int32]),
    "acmpc_set_coefficients": (C.c_int, [_CTX, C.c_void_p, C.c_int32, C.c_int32]),
    "acmpc_record_floats": (C.c_int32, [C.c_int32]),
    "acmpc_solve": (C.c_int, [_CTX, _F32P, _F32P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _F32P, _I32P, _F32P]),
    "acmpc_solve_device": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "acmpc_rollout_device": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "acmpc_finalize_device": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "acmpc_softmin_device": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "acmpc_sync_tables": (C.c_int, [_CTX, C.c_void_p]),
    "acmpc_sample_device": (C.c_int, [_CTX, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int64, C.c_double, C.c_double, C.c_uint64, C.c_uint32,
                                      C.c_void_p, C.c_void_p]),
    "acmpc_finalize_sampled_device": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                                C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_uint64,
                                                C.c_uint32, C.c_void_p, C.c_void_p]),
    "acmpc_solve_sampled_device": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                             C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_uint64,
                                             C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "acmpc_solve_stream_device": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                            C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_uint64,
                                            C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "acmpc_solve_stream_flush": (C.c_int, [_CTX, C.c_void_p]),
    "acmpc_reduce_across_ranks": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    # the closed-loop entry points take their buffers as plain addresses (arr.ctypes.data): half the marshalling cost
    # of typed pointers, which at these sizes is most of the call
    "acmpc_optimize": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_void_p, C.c_double, C.c_uint64, C.c_void_p]),
    "acmpc_control_tick": (C.c_int, [_CTX, C.POINTER(Tick)] + [C.c_void_p] * 13),
    "acmpc_bind_map": (C.c_int, [_CTX, C.c_void_p, C.c_int32, C.c_double]),
    "acmpc_map_reference_path": (C.c_int, [_CTX, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32,
                                           C.c_void_p, _I32P]),
    "acmpc_tick_read_device_tables": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_void_p]),
    "acmpc_tick_read_device_frames": (C.c_int, [_CTX, C.c_void_p, C.c_int64]),
    "acmpc_speed_profile_qp_device": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_double,
                                                C.c_double, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p,
                                                C.c_void_p, C.c_int32, _I32P]),
    "acmpc_waypoint_table": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_void_p]),
    "acmpc_velocity_ceiling": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32,
                                         C.c_int32, C.c_double, C.c_void_p]),
    "acmpc_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint64]),
    "acmpc_host_free": (C.c_int, [C.c_void_p]),
    "acmpc_unpack_decision": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_double] + [C.c_void_p] * 6),
    "acmpc_unpack_decision_temporal": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double] + [C.c_void_p] * 6),
    "acmpc_search_window": (C.c_int32, [_I32P]),
    "acmpc_search_frame_floats": (C.c_int32, [C.c_int32]),
    "acmpc_search_frames": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]),
    "acmpc_philox4x32": (None, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "acmpc_speed_profile_qp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32,
                                         C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int32, _I32P]),
    "acmpc_speed_profile_exact": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double,
                                            C.c_void_p, C.c_void_p]),
    "acmpc_pf_create": (C.c_int, [C.POINTER(PfParams), _F64P, C.c_int32, _F64P, C.c_int32, _F64P, C.c_int32,
                                  C.POINTER(_CTX)]),
    "acmpc_pf_destroy": (None, [_CTX]),
    "acmpc_pf_last_error": (C.c_char_p, [_CTX]),
    "acmpc_pf_score_scale": (C.c_double, [_CTX]),
    "acmpc_pf_score": (C.c_int, [_CTX, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32] +
                       [C.c_void_p] * 6),
    "acmpc_pf_advance": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double]),
    "acmpc_pf_estimate": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, _F64P, _F64P]),
    "acmpc_pf_filter_reset": (C.c_int, [_CTX, C.c_int32]),
    "acmpc_pf_filter_set": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_int32]),
    "acmpc_pf_filter_get": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_int32, _I32P]),
    "acmpc_pf_filter_step": (C.c_int, [_CTX, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint64,
                                       C.c_uint32]),
    "acmpc_pf_filter_update": (C.c_int, [_CTX, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(PfResample),
                                         C.c_void_p]),
    "acmpc_profile_enable": (C.c_int, [_CTX, C.c_int32]),
    "acmpc_profile_collect": (C.c_int, [_CTX, _F32P, C.c_int32, _I32P]),
    "acmpc_pack_key": (C.c_int64, [C.c_float, C.c_uint32]),
    "acmpc_key_cost": (C.c_float, [C.c_int64]),
    "acmpc_key_index": (C.c_uint32, [C.c_int64]),
}

_lib: Optional[C.CDLL] = None


def library_path() -> str:
    return os.environ.get("ACMPC_HIP_LIBRARY", _build.LIB_PATH)


def _share_torchs_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 and look them up by the unversioned file
    name; this library needs them by SONAME.  Whichever of the two is loaded first decides: torch first - one
    runtime, shared (the loader matches our SONAME against torch's copy); ours first - the system's runtime is
    mapped, torch later maps its own beside it, and the second HSA runtime in the process finds no GPU
    ("No HIP GPUs are available").  So when torch is installed but not imported yet, map torch's copies first; the
    later `import torch` then resolves to the very same files.  ACMPC_SYSTEM_HIP=1 skips this."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("ACMPC_SYSTEM_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return   # an unusable bundle: fall back to the system runtime


def load_library() -> C.CDLL:
    """dlopen libacmpc_hip.so and bind every declared symbol.  Raises if the library is absent: build it with
    `python -m acmpc_amd._build` (or `__graft_entry__.build()`)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise EngineError(ENODEVICE, "HIP extension %s is missing - build it with acmpc_amd._build.build_library(); "
                          "there is no CPU fallback" % path)
    _share_torchs_hip_runtime()
    lib = C.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _f32(a: np.ndarray):
    return a.ctypes.data_as(_F32P)


def record_floats(n: int) -> int:
    return REC_HEADER + 2 * n + 3 * (n + 1)


def split_record(rec: np.ndarray, n: int):
    """[cost, violation, n_feasible, owner, u (n x 2), x ((n+1) x 3)] -> dict of views."""
    u = rec[..., REC_HEADER:REC_HEADER + 2 * n].reshape(rec.shape[:-1] + (n, 2))
    x = rec[..., REC_HEADER + 2 * n:].reshape(rec.shape[:-1] + (n + 1, 3))
    return dict(cost=rec[..., REC_COST], violation=rec[..., REC_VIOLATION], n_feasible=rec[..., REC_NFEASIBLE],
                owner=rec[..., REC_OWNER], u=u, x=x)


def lq_plan(table: np.ndarray, x0, step_cost, r_term, final_cost, u_min, u_max):
    """The LQ plan of one path on the host (acmpc_lq_plan): table [7, n] float64, x0 = (e_y, e_psi, t) -> [n, 2] float32
    (v, kappa), or None when the problem has no finite plan."""
    lib = load_library()
    table = np.ascontiguousarray(table, dtype=np.float64)
    n = table.shape[1]
    args = [np.ascontiguousarray(a, dtype=np.float64) for a in (x0, step_cost, r_term, final_cost)]
    box = [np.ascontiguousarray(a, dtype=np.float32) for a in (u_min, u_max)]
    plan = np.empty((n, 2), dtype=np.float32)
    rc = lib.acmpc_lq_plan(table.ctypes.data, n, *(a.ctypes.data for a in args), *(b.ctypes.data for b in box),
                           plan.ctypes.data)
    return plan if rc == OK else None


RCCL_UNIQUE_ID_BYTES = 128


def rccl_unique_id() -> bytes:
    """ncclGetUniqueId of the RCCL `Engine.reduce_across_ranks` resolves (acmpc_rccl_unique_id): call on ONE rank, hand the
    128 bytes to the others."""
    buffer = C.create_string_buffer(RCCL_UNIQUE_ID_BYTES)
    rc = load_library().acmpc_rccl_unique_id(buffer)
    if rc != OK:
        raise EngineError(rc, "acmpc_rccl_unique_id: no RCCL in the process and librccl.so.1 not loadable"
                          if rc == ESTATE else "acmpc_rccl_unique_id: ncclGetUniqueId failed")
    return buffer.raw


def rccl_comm_create(unique_id: bytes, n_ranks: int, rank: int, device: int = -1) -> int:
    """ncclCommInitRank (acmpc_rccl_comm_create): collective over the `n_ranks` ranks that hold `unique_id`; returns the
    communicator as an integer handle for `Engine.reduce_across_ranks` / `rccl_comm_destroy`."""
    if len(unique_id) != RCCL_UNIQUE_ID_BYTES:
        raise EngineError(EINVAL, "an ncclUniqueId is %d bytes" % RCCL_UNIQUE_ID_BYTES)
    comm = C.c_void_p()
    rc = load_library().acmpc_rccl_comm_create(C.create_string_buffer(unique_id, RCCL_UNIQUE_ID_BYTES), int(n_ranks), int(rank),
                                               int(device), C.byref(comm))
    if rc != OK:
        raise EngineError(rc, "acmpc_rccl_comm_create(n_ranks=%d, rank=%d) failed" % (n_ranks, rank))
    return comm.value


def rccl_comm_destroy(comm: int) -> None:
    if comm:
        load_library().acmpc_rccl_comm_destroy(C.c_void_p(comm))


def lq_box_plan(table: np.ndarray, x0, step_cost, r_term, final_cost, u_min, u_max, margin: float, w_bound: float,
                iterations: int = 40, state: np.ndarray = None):
    """The plan of `lq_candidate=2` for one path on the host (acmpc_lq_box_plan, csrc/acmpc_lq_box.h): the LQ plan refined
    against the QP's box rows where it is not already the optimum.  `state`: the [1 + 8 n] iterate a previous call
    returned (None: cold).  Returns dict(plan [n, 2] float32 or None, iterations, chosen, triggered, J, V, state)."""
    lib = load_library()
    table = np.ascontiguousarray(table, dtype=np.float64)
    n = table.shape[1]
    args = [np.ascontiguousarray(a, dtype=np.float64) for a in (x0, step_cost, r_term, final_cost)]
    box = [np.ascontiguousarray(a, dtype=np.float32) for a in (u_min, u_max)]
    plan = np.empty((n, 2), dtype=np.float32)
    iterate = np.zeros(1 + 8 * n) if state is None else np.array(state, dtype=np.float64)
    info = np.zeros(5)
    rc = lib.acmpc_lq_box_plan(table.ctypes.data, n, *(a.ctypes.data for a in args), *(b.ctypes.data for b in box),
                               float(margin), float(w_bound), int(iterations), iterate.ctypes.data, plan.ctypes.data,
                               info.ctypes.data)
    if rc != OK:
        return dict(plan=None, iterations=0, chosen=0, triggered=False, J=float("nan"), V=float("nan"), state=None)
    return dict(plan=plan, iterations=int(info[0]), chosen=int(info[1]), triggered=bool(info[2]), J=float(info[3]),
                V=float(info[4]), state=iterate if iterate[0] == n else None)


class _TickBuffers:
    """Everything `Engine.control_tick` hands to the library for one horizon, allocated once: the inputs are copied
    into fixed arrays (taking an array's address through `.ctypes` costs more than copying 150 doubles), the outputs
    land in ONE float64 block (+ the float32 record) whose snapshot is one copy per tick, and the argument tuple of the
    call is built once.  A tick's Python side is then a handful of microseconds instead of ten."""
    FIELDS = (("table", lambda n: (7, n)), ("decision", lambda n: (5 * n + 3,)), ("projected_control", lambda n: (2, n)),
              ("prediction", lambda n: (n, 2)), ("cum_time", lambda n: (n,)), ("times", lambda n: (n - 1,)),
              ("accelerations", lambda n: (n - 1,)), ("steer_rates", lambda n: (n - 1,)), ("info", lambda n: (8,)),
              ("coords", lambda n: (n + 1, 3)))

    def __init__(self, function, ctx, tick, n):
        self._function = function
        self._coords_in = np.empty((n + 1, 3))
        self._centre_in = np.empty((n, 2), dtype=np.float32)
        self._record = np.empty(record_floats(n), dtype=np.float32)
        self._layout = []   # (name, slice of the block, shape or None for the one-dimensional ones)
        offset, address_of = 0, {}
        for name, shape_of in self.FIELDS:
            shape = shape_of(n)
            count = int(np.prod(shape))
            self._layout.append((name, slice(offset, offset + count), shape if len(shape) > 1 else None))
            address_of[name] = offset * 8
            offset += count
        self._block = np.empty(offset)
        base = self._block.ctypes.data
        address = {name: base + byte_offset for name, byte_offset in address_of.items()}
        self._ctx = ctx
        self._outputs = (address["table"], self._record.ctypes.data, address["decision"], address["projected_control"],
                         address["prediction"], address["cum_time"], address["times"], address["accelerations"],
                         address["steer_rates"], address["info"], address["coords"])
        self._tick = None
        self.bind(tick)

    def bind(self, tick):
        """The struct object the call passes: a solver keeps using its one object, so this runs once."""
        self._tick = tick   # (kept alive here)
        tick_ref = C.byref(tick)
        coords_address, centre_address = self._coords_in.ctypes.data, self._centre_in.ctypes.data
        self._arguments = {(c, k): (self._ctx, tick_ref, coords_address if c else None, centre_address if k else None)
                                   + self._outputs for c in (False, True) for k in (False, True)}

    def call(self, tick, coords, centre):
        if tick is not self._tick:
            self.bind(tick)
        if coords is not None:
            self._coords_in[...] = coords
        if centre is not None:
            self._centre_in[...] = centre
        return self._function(*self._arguments[coords is not None, centre is not None])

    def snapshot(self):
        block = self._block.copy()   # one copy; the arrays handed out are views of it (fresh every tick)
        out = {}
        for name, where, shape in self._layout:
            out[name] = block[where] if shape is None else block[where].reshape(shape)
        out["record"] = self._record.copy()
        return out


class Engine:
    """Owns one acmpc_ctx.  Construction does no device work (fork-safe, controller.py:293-297)."""

    def __init__(self, *, mode: int, max_problems: int, max_candidates: int, max_steps: int, step_cost, r_term,
                 final_cost, u_min, u_max, margin: float, wheelbase: float, t_min: float = 0.01, dt: float = 0.05,
                 w_bound: float = 1.0e6, softmin_lambda: float = 1.0, device: int = -1, nn_window=None,
                 centre_update: str = "argmin", lq_candidate=False):
        """`nn_window=(back, ahead)` restricts mode T's nearest-waypoint search to that many waypoints round the
        previous step's nearest index; None = the nearest of ALL waypoints at every step (localiser.py:282-289's
        semantics: the kernels search an 8-waypoint window whose winner a certificate accepts as the global one, and
        scan every waypoint where it does not - the same index either way, see csrc/acmpc_frames.h).
        `lq_candidate`: the last sampling round of `optimize` / `control_tick` also holds the LQ plan (the optimum of the
        reference's control QP without its box rows, rolled forward and clipped: csrc/acmpc_lq.h) as candidate 2;
        `lq_candidate=2`: that plan refined against the QP WITH its box rows wherever a control sits on the input box or
        a state row is violated (csrc/acmpc_lq_box.h)."""
        self._lib = load_library()
        p = Params()
        p.struct_size = C.sizeof(Params)
        p.mode, p.device = mode, device
        p.max_problems, p.max_candidates, p.max_steps = max_problems, max_candidates, max_steps
        p.nn_back, p.nn_ahead = (-1, -1) if nn_window is None else (int(nn_window[0]), int(nn_window[1]))
        p.centre_update = {"argmin": 0, "softmin": 1}[centre_update]
        p.lq_candidate = int(lq_candidate)   # False / True / 2
        p.step_cost[:] = [float(v) for v in step_cost]
        p.r_term[:] = [float(v) for v in r_term]
        p.final_cost[:] = [float(v) for v in final_cost]
        p.u_min[:] = [float(v) for v in u_min]
        p.u_max[:] = [float(v) for v in u_max]
        p.margin, p.wheelbase, p.t_min, p.dt = float(margin), float(wheelbase), float(t_min), float(dt)
        p.w_bound, p.softmin_lambda = float(w_bound), float(softmin_lambda)
        self.params = p
        self.mode = mode
        self._ctx = _CTX()
        rc = self._lib.acmpc_create(C.byref(p), C.byref(self._ctx))
        if rc != OK:
            raise EngineError(rc, (self._lib.acmpc_last_error(None) or b"").decode())
        self.P = 0
        self.n = 0

    # -- plumbing -----------------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != OK:
            raise EngineError(rc, (self._lib.acmpc_last_error(self._ctx) or b"").decode())

    def lq_box_stats(self):
        """What the last `lq_candidate=2` plan of this handle did: dict(iterations, chosen, triggered, J, V)."""
        info = np.zeros(5)
        self._check(self._lib.acmpc_lq_box_stats(self._ctx, info.ctypes.data))
        return dict(iterations=int(info[0]), chosen=int(info[1]), triggered=bool(info[2]), J=float(info[3]), V=float(info[4]))

    def set_option(self, name: str, value=None):
        """One of the handle's A/B switches (tools/README.md; `name` as the environment spells it, e.g. "ACMPC_NO_SOLO").
        They are read from the environment once, when the handle is created; this sets one afterwards.  None = default."""
        self._check(self._lib.acmpc_set_option(self._ctx, name.encode(), None if value is None else str(value).encode()))

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.acmpc_destroy(self._ctx)
            self._ctx = _CTX()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host-only ----------------------------------------------------------------------------------------
    def set_paths(self, tables: np.ndarray):
        """tables: [P, 7, n] (or [7, n]) float64 ReferencePath arrays, rows [x, y, psi, kappa, ds, width, v]."""
        t = np.ascontiguousarray(tables, dtype=np.float64)
        if t.ndim == 2:
            t = t[None]
        if t.ndim != 3 or t.shape[1] != 7:
            raise ValueError("tables must be [P, 7, n]")
        self._check(self._lib.acmpc_set_paths(self._ctx, t.ctypes.data, t.shape[0], t.shape[2]))
        self.P, self.n = t.shape[0], t.shape[2]

    def set_coefficients(self, coef: np.ndarray):
        """The packed float32 tables themselves, [P, n, 12] (mode S) / [P, n, 8] (mode T) or one [n, stride] table
        (acmpc_set_coefficients): e.g. what `tick_device_tables` read back."""
        coef = np.ascontiguousarray(coef, dtype=np.float32)
        if coef.ndim == 2:
            coef = coef[None]
        P, n, stride = coef.shape
        if stride != COEF_STRIDE[self.mode]:
            raise ValueError("coefficient rows must have %d floats" % COEF_STRIDE[self.mode])
        self._check(self._lib.acmpc_set_coefficients(self._ctx, coef.ctypes.data, P, n))
        self.P, self.n = P, n

    def coefficients(self, problem: int = 0) -> np.ndarray:
        out = np.empty((self.n, COEF_STRIDE[self.mode]), dtype=np.float32)
        self._check(self._lib.acmpc_get_coefficients(self._ctx, problem, _f32(out), out.size))
        return out

    # -- host-pointer solve -------------------------------------------------------------------------------
    def solve(self, x0: np.ndarray, U: np.ndarray, layout: int = LAYOUT_CANDIDATE_MAJOR, want_costs: bool = True):
        """x0 [P,3], U [P,N,n,2] (layout 0) or [P,n,2,N] (layout 1) -> dict(best_idx, costs, record fields)."""
        x0 = np.ascontiguousarray(x0, dtype=np.float32).reshape(-1, 3)
        U = np.ascontiguousarray(U, dtype=np.float32)
        if U.ndim == 3:
            U = U[None]
        P = x0.shape[0]
        if layout == LAYOUT_CANDIDATE_MAJOR:
            _, N, n, two = U.shape
        else:
            _, n, two, N = U.shape
        if two != 2 or U.shape[0] != P:
            raise ValueError("U has the wrong shape for layout %d" % layout)
        costs = np.empty((P, N), dtype=np.float32) if want_costs else None
        best = np.empty(P, dtype=np.int32)
        rec = np.empty((P, record_floats(n)), dtype=np.float32)
        self._check(self._lib.acmpc_solve(self._ctx, _f32(x0), _f32(U), P, N, n, layout,
                                          _f32(costs) if want_costs else None, best.ctypes.data_as(_I32P), _f32(rec)))
        out = split_record(rec, n)
        out.update(best_idx=best, costs=costs, records=rec)
        return out

    # -- device-pointer entry points (pointers are plain integers, e.g. torch.Tensor.data_ptr()) ----------
    def sync_tables(self, stream: int = 0):
        self._check(self._lib.acmpc_sync_tables(self._ctx, stream))

    def sample_device(self, d_centre: int, centre_stride: int, d_u_ref: int, P: int, N: int, n: int, layout: int,
                      index_offset: int, sigma, seed: int, round_: int, d_U: int, stream: int = 0):
        self._check(self._lib.acmpc_sample_device(self._ctx, d_centre, centre_stride, d_u_ref or None, P, N, n, layout,
                                                  index_offset, float(sigma[0]), float(sigma[1]), seed, round_, d_U,
                                                  stream or None))

    def finalize_sampled_device(self, d_keys: int, d_x0: int, d_centre: int, centre_stride: int, d_u_ref: int, P: int,
                                N: int, n: int, sigma, seed: int, round_: int, d_records: int, stream: int = 0):
        self._check(self._lib.acmpc_finalize_sampled_device(self._ctx, d_keys or None, d_x0, d_centre, centre_stride,
                                                            d_u_ref or None, P, N, n, float(sigma[0]), float(sigma[1]),
                                                            seed, round_, d_records, stream or None))

    def solve_sampled_device(self, d_x0: int, d_U: int, d_centre: int, centre_stride: int, d_u_ref: int, P: int, N: int,
                             n: int, layout: int, sigma, seed: int, round_: int, d_costs: int, d_keys: int, d_records: int,
                             stream: int = 0):
        """rollout_device + finalize_sampled_device as one call (one launch where the shape allows it) for a rank that
        holds all the candidates: acmpc_solve_sampled_device."""
        self._check(self._lib.acmpc_solve_sampled_device(self._ctx, d_x0, d_U, d_centre, centre_stride, d_u_ref or None, P,
                                                         N, n, layout, float(sigma[0]), float(sigma[1]), seed, round_,
                                                         d_costs or None, d_keys or None, d_records, stream or None))

    def solve_stream_device(self, d_x0: int, d_U: int, d_centre: int, centre_stride: int, d_u_ref: int, P: int, N: int,
                            n: int, layout: int, sigma, seed: int, round_: int, d_costs: int, d_keys: int, d_records: int,
                            stream: int = 0):
        """One batch of a stream of batches (acmpc_solve_stream_device): its rollout now, its argmin and records inside
        the next call's launch or behind `solve_stream_flush`.  `d_centre` 0: the winners are read from `d_U`."""
        self._check(self._lib.acmpc_solve_stream_device(self._ctx, d_x0, d_U, d_centre or None, centre_stride, d_u_ref or None,
                                                        P, N, n, layout, float(sigma[0]), float(sigma[1]), seed, round_,
                                                        d_costs or None, d_keys or None, d_records, stream or None))

    def solve_stream_flush(self, stream: int = 0):
        self._check(self._lib.acmpc_solve_stream_flush(self._ctx, stream or None))

    def optimize(self, x0: np.ndarray, centre: np.ndarray, u_ref, n_candidates: int, rounds: int, sigma,
                 shrink: float = 0.5, seed: int = 0):
        """Sampling optimisation entirely on the device; x0 [P,3], centre/u_ref [P,n,2] -> record fields."""
        x0 = np.ascontiguousarray(x0, dtype=np.float32).reshape(-1, 3)
        centre = np.ascontiguousarray(centre, dtype=np.float32)
        if centre.ndim == 2:
            centre = centre[None]
        P, n = centre.shape[0], centre.shape[1]
        ref = None
        if u_ref is not None:
            ref = np.ascontiguousarray(u_ref, dtype=np.float32).reshape(P, n, 2)
        sig = np.array([sigma[0], sigma[1]], dtype=np.float64)
        rec = np.empty((P, record_floats(n)), dtype=np.float32)
        self._check(self._lib.acmpc_optimize(self._ctx, x0.ctypes.data, centre.ctypes.data,
                                             ref.ctypes.data if ref is not None else None, P, n_candidates, n, rounds,
                                             sig.ctypes.data, float(shrink), seed, rec.ctypes.data))
        out = split_record(rec, n)
        out["records"] = rec
        return out

    def control_tick(self, tick: Tick, coords: np.ndarray, centre):
        """One whole control tick on the device (acmpc_control_tick).  coords [H,3] float64 - or None: the path is
        cut out of the bound map at tick.map_index / the pose - centre [n,2] float32 or None (with
        tick.centre_is_reference).  Returns a dict of fresh arrays: table [7,n], record, decision [5n+3],
        projected_control [2,n], prediction [n,2], cum_time [n], times / accelerations / steer_rates [n-1], info [8],
        coords [H,3] (the path used)."""
        n = tick.horizon - 1
        buffers = self.__dict__.get("_tick_buffers")
        if buffers is None:
            buffers = self._tick_buffers = {}
        buf = buffers.get(n)
        if buf is None:
            buf = buffers[n] = _TickBuffers(self._lib.acmpc_control_tick, self._ctx, tick, n)
        rc = buf.call(tick, coords, centre)
        if rc != OK:
            self._check(rc)
        return buf.snapshot()

    def bind_map(self, centre: np.ndarray, spacing: float):
        """Centre polyline [M,2] float64 of the map the tick may cut its reference path from (acmpc_bind_map)."""
        centre = np.ascontiguousarray(centre, dtype=np.float64)
        self._check(self._lib.acmpc_bind_map(self._ctx, centre.ctypes.data, centre.shape[0], float(spacing)))

    def map_reference_path(self, horizon: int, map_index: int = -1, pose=(0.0, 0.0), lateral_offset: float = 0.0,
                           centreline_points: int = 500):
        """(coords [H,3], first map index): the window kernel alone (acmpc_map_reference_path)."""
        coords = np.empty((horizon, 3))
        first = C.c_int32(0)
        self._check(self._lib.acmpc_map_reference_path(self._ctx, int(map_index), float(pose[0]), float(pose[1]),
                                                       float(lateral_offset), horizon, centreline_points,
                                                       coords.ctypes.data, C.byref(first)))
        return coords, first.value

    def tick_device_tables(self, n: int):
        """(x0 [3], u_ref [n,2], coef [n,12] - mode T: [n,8]) the last tick's prologue left on the device (test hook)."""
        x0 = np.empty(3, dtype=np.float32)
        u_ref = np.empty((n, 2), dtype=np.float32)
        coef = np.empty((n, COEF_STRIDE[self.mode]), dtype=np.float32)
        self._check(self._lib.acmpc_tick_read_device_tables(self._ctx, x0.ctypes.data, u_ref.ctypes.data, coef.ctypes.data))
        return x0, u_ref, coef

    def tick_device_frames(self, n: int):
        """The verified search's frames the last tick's prologue tabulated (mode T, exhaustive search; test hook)."""
        out = np.empty(self._lib.acmpc_search_frame_floats(n), dtype=np.float32)
        self._check(self._lib.acmpc_tick_read_device_frames(self._ctx, out.ctypes.data, out.size))
        return out

    def speed_profile_qp_device(self, v_hi, ds, a_min, a_max, v_min, max_iter=4000, eps_abs=1e-3, eps_rel=1e-3,
                                warm=None, check_every=10):
        """The prologue's ADMM alone on the GPU (test hook); same return as `speed_profile_qp`."""
        v_hi = np.ascontiguousarray(v_hi, dtype=np.float64)
        ds = np.ascontiguousarray(ds, dtype=np.float64)
        n = v_hi.shape[0]
        v, y = np.zeros(n), np.zeros(2 * n - 1)
        if warm is not None:
            v[:], y[:] = warm
        iters = C.c_int32(0)
        rc = self._lib.acmpc_speed_profile_qp_device(self._ctx, v_hi.ctypes.data, ds.ctypes.data, n, float(a_min),
                                                     float(a_max), float(v_min), int(max_iter), int(check_every),
                                                     float(eps_abs), float(eps_rel), v.ctypes.data, y.ctypes.data,
                                                     1 if warm is not None else 0, C.byref(iters))
        if rc < 0:
            self._check(rc)
        return v, y, ("solved" if rc == 0 else "maximum iterations reached"), iters.value

    def rollout_start_clocks(self, capacity: int = 1 << 20) -> np.ndarray:
        """Start times [us, from the earliest] of the workgroups of the last rollout launch - needs
        `set_option("ACMPC_START_CLOCKS", "1")` before it (acmpc_rollout_start_clocks).  Empty without a stamped launch."""
        raw = np.zeros(capacity, dtype=np.uint64)
        count = C.c_int32(0)
        self._check(self._lib.acmpc_rollout_start_clocks(self._ctx, raw.ctypes.data, capacity, C.byref(count)))
        ticks = raw[:count.value].astype(np.int64)
        return (ticks - ticks.min()) / 100.0 if count.value else np.zeros(0)

    def profile_enable(self, capacity: int):
        """Attach event pairs to the next `capacity` rollout launches (no extra packets on the stream)."""
        self._check(self._lib.acmpc_profile_enable(self._ctx, capacity))
        self._profile_capacity = capacity

    def profile_collect(self) -> np.ndarray:
        """Durations [ms] of the rollout launches timed since the last enable/collect; re-arms the pairs."""
        cap = getattr(self, "_profile_capacity", 0)
        out = np.empty(max(cap, 1), dtype=np.float32)
        count = C.c_int32(0)
        self._check(self._lib.acmpc_profile_collect(self._ctx, _f32(out), cap, C.byref(count)))
        return out[:count.value].copy()

    def solve_device(self, d_x0: int, d_U: int, P: int, N: int, n: int, layout: int, d_costs: int, d_keys: int,
                     d_records: int, stream: int = 0):
        self._check(self._lib.acmpc_solve_device(self._ctx, d_x0, d_U, P, N, n, layout, d_costs or None,
                                                 d_keys or None, d_records or None, stream or None))

    def rollout_device(self, d_x0: int, d_U: int, P: int, N: int, n: int, layout: int, index_offset: int,
                       d_costs: int, d_keys: int, stream: int = 0):
        self._check(self._lib.acmpc_rollout_device(self._ctx, d_x0, d_U, P, N, n, layout, index_offset,
                                                   d_costs or None, d_keys or None, stream or None))

    def finalize_device(self, d_keys: int, d_x0: int, d_U: int, P: int, N: int, n: int, layout: int,
                        index_offset: int, d_records: int, stream: int = 0):
        self._check(self._lib.acmpc_finalize_device(self._ctx, d_keys or None, d_x0, d_U, P, N, n, layout, index_offset,
                                                    d_records, stream or None))

    def reduce_across_ranks(self, rccl_comm: int, d_keys: int, P: int, stream: int = 0):
        """In-place all-reduce(MIN) of P packed keys over an `ncclComm_t` the caller owns (RCCL resolved at run time)."""
        self._check(self._lib.acmpc_reduce_across_ranks(self._ctx, rccl_comm, d_keys, P, stream or None))

    def softmin_device(self, d_costs: int, d_keys: int, d_U: int, P: int, N: int, n: int, layout: int, d_mean: int,
                       d_weight_sum: int = 0, stream: int = 0):
        self._check(self._lib.acmpc_softmin_device(self._ctx, d_costs, d_keys, d_U, P, N, n, layout, d_mean,
                                                   d_weight_sum or None, stream or None))


def waypoint_table(coords: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    """H x 3 (x, y, width) -> 7 x n ReferencePath table (native; spatial_mpc.py:125-154)."""
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    if coords.ndim != 2 or coords.shape[1] != 3:
        raise ValueError("waypoint coordinates must be H x 3")
    table = np.empty((7, coords.shape[0] - 1))
    if load_library().acmpc_waypoint_table(coords.ctypes.data, coords.shape[0], eps, table.ctypes.data) != OK:
        raise EngineError(EINVAL, "acmpc_waypoint_table: need at least 3 points")
    return table


def velocity_ceiling(kappa: np.ndarray, ay_max: float, ki_min: float, v_min: float, v_max: float, localised: bool,
                     end_velocity) -> np.ndarray:
    kappa = np.ascontiguousarray(kappa, dtype=np.float64)
    out = np.empty(kappa.shape[0])
    rc = load_library().acmpc_velocity_ceiling(kappa.ctypes.data, kappa.shape[0], float(ay_max), float(ki_min),
                                               float(v_min), float(v_max), 1 if localised else 0,
                                               0 if end_velocity is None else 1,
                                               0.0 if end_velocity is None else float(end_velocity), out.ctypes.data)
    if rc != OK:
        raise EngineError(rc, "acmpc_velocity_ceiling: bad arguments")
    return out


_unpack_local = threading.local()   # per thread: n -> (arrays, their addresses), persistent outputs copied out per
                                     # call (ctypes releases the GIL during the call, so threads must not share them)


def unpack_decision(z: np.ndarray, n: int, table: np.ndarray, wheelbase: float):
    """dec.x -> (projected_control [2,n], prediction [n,2], cum_time [n], times, accelerations, steer_rates [n-1])."""
    z = np.ascontiguousarray(z, dtype=np.float64)
    table = np.ascontiguousarray(table, dtype=np.float64)
    if z.shape[0] != 5 * n + 3 or table.shape != (7, n):
        raise ValueError("decision vector / table do not match n = %d" % n)
    buffers = getattr(_unpack_local, "buffers", None)
    if buffers is None:
        buffers = _unpack_local.buffers = {}
    cached = buffers.get(n)
    if cached is None:
        arrays = (np.empty((2, n)), np.empty((n, 2)), np.empty(n), np.empty(n - 1), np.empty(n - 1), np.empty(n - 1))
        cached = buffers[n] = (arrays, tuple(a.ctypes.data for a in arrays))
    arrays, addresses = cached
    rc = load_library().acmpc_unpack_decision(z.ctypes.data, n, table.ctypes.data, float(wheelbase), *addresses)
    if rc != OK:
        raise EngineError(rc, "acmpc_unpack_decision: bad arguments")
    return tuple(a.copy() for a in arrays)


def pinned_empty(shape, dtype=np.float32) -> np.ndarray:
    """A NumPy array in page-locked host memory (`acmpc_host_alloc`): a control matrix built in it is read in place by
    `Engine.solve`'s rollout (no copy in front of the kernel: 58 us instead of 74 per 4 096-candidate solve).  Initialises the HIP runtime - call in the process that solves.  The
    memory is returned to the runtime when the array (and every view of it) has been collected."""
    import weakref
    lib = load_library()
    count = int(np.prod(shape))
    nbytes = max(count * np.dtype(dtype).itemsize, 1)
    pointer = C.c_void_p()
    rc = lib.acmpc_host_alloc(C.byref(pointer), nbytes)
    if rc != OK:
        raise EngineError(rc, (lib.acmpc_last_error(None) or b"").decode())
    buffer = (C.c_char * nbytes).from_address(pointer.value)
    weakref.finalize(buffer, lib.acmpc_host_free, pointer.value)
    return np.frombuffer(buffer, dtype=dtype, count=count).reshape(shape)


def unpack_decision_temporal(z: np.ndarray, n: int, dt: float, wheelbase: float):
    """A mode T plan's dec.x -> the same six arrays (`acmpc_unpack_decision_temporal`: prediction = the rolled poses,
    cum_time = i dt, derivatives of the plan's own controls)."""
    z = np.ascontiguousarray(z, dtype=np.float64)
    if z.shape[0] != 5 * n + 3:
        raise ValueError("decision vector does not match n = %d" % n)
    arrays = (np.empty((2, n)), np.empty((n, 2)), np.empty(n), np.empty(n - 1), np.empty(n - 1), np.empty(n - 1))
    rc = load_library().acmpc_unpack_decision_temporal(z.ctypes.data, n, float(dt), float(wheelbase),
                                                       *(a.ctypes.data for a in arrays))
    if rc != OK:
        raise EngineError(rc, "acmpc_unpack_decision_temporal: bad arguments")
    return arrays


def search_window():
    """(waypoints per window, how many of them behind the previous nearest one) of mode T's verified search."""
    back = C.c_int32(0)
    width = load_library().acmpc_search_window(C.byref(back))
    return int(width), int(back.value)


def search_frames(coef: np.ndarray) -> np.ndarray:
    """coef [P][n][8] packed waypoint rows -> [P][acmpc_search_frame_floats(n)] frames of the verified search, as
    acmpc_set_paths hands them to the kernels (host computation)."""
    coef = np.ascontiguousarray(coef, dtype=np.float32)
    P, n = coef.shape[0], coef.shape[1]
    lib = load_library()
    floats = lib.acmpc_search_frame_floats(n)
    out = np.empty((P, floats), dtype=np.float32)
    rc = lib.acmpc_search_frames(coef.ctypes.data, P, n, out.ctypes.data, out.size)
    if rc != OK:
        raise EngineError(rc, "acmpc_search_frames: bad arguments")
    return out


def speed_profile_qp(v_hi: np.ndarray, ds: np.ndarray, a_min: float, a_max: float, v_min: float, max_iter: int = 4000,
                     eps_abs: float = 1e-3, eps_rel: float = 1e-3, warm=None, check_every: int = 10):
    """Native tridiagonal ADMM for the speed-profile QP; returns (v, y, status, iterations)."""
    v_hi = np.ascontiguousarray(v_hi, dtype=np.float64)
    ds = np.ascontiguousarray(ds, dtype=np.float64)
    n = v_hi.shape[0]
    v = np.zeros(n)
    y = np.zeros(2 * n - 1)
    if warm is not None:
        v[:], y[:] = warm
    iters = C.c_int32(0)
    rc = load_library().acmpc_speed_profile_qp(v_hi.ctypes.data, ds.ctypes.data, n, float(a_min), float(a_max),
                                               float(v_min), int(max_iter), int(check_every), float(eps_abs),
                                               float(eps_rel),
                                               v.ctypes.data, y.ctypes.data, 1 if warm is not None else 0,
                                               C.byref(iters))
    if rc < 0:
        raise EngineError(rc, "acmpc_speed_profile_qp: bad arguments")
    return v, y, ("solved" if rc == 0 else "maximum iterations reached"), iters.value


def speed_profile_exact(v_hi: np.ndarray, ds: np.ndarray, a_min: float, a_max: float, v_min: float):
    """The speed-profile QP's exact optimum in two passes (acmpc_speed_profile_exact): (v, y = 0), or None where the
    problem is infeasible or not of the shape the passes solve - `speed_profile_qp` is for those."""
    v_hi = np.ascontiguousarray(v_hi, dtype=np.float64)
    ds = np.ascontiguousarray(ds, dtype=np.float64)
    n = v_hi.shape[0]
    v, y = np.zeros(n), np.zeros(2 * n - 1)
    rc = load_library().acmpc_speed_profile_exact(v_hi.ctypes.data, ds.ctypes.data, n, float(a_min), float(a_max),
                                                  float(v_min), v.ctypes.data, y.ctypes.data)
    if rc < 0:
        raise EngineError(rc, "acmpc_speed_profile_exact: bad arguments")
    return (v, y) if rc == 0 else None


def philox4x32(counter, key) -> np.ndarray:
    ctr = (C.c_uint32 * 4)(*[int(v) for v in counter])
    k = (C.c_uint32 * 2)(*[int(v) for v in key])
    out = (C.c_uint32 * 4)()
    load_library().acmpc_philox4x32(ctr, k, out)
    return np.array(list(out), dtype=np.uint32)


def pack_key(cost: float, index: int) -> int:
    return int(load_library().acmpc_pack_key(float(cost), int(index)))


def key_cost(key: int) -> float:
    return float(load_library().acmpc_key_cost(int(key)))


def key_index(key: int) -> int:
    return int(load_library().acmpc_key_index(int(key)))
