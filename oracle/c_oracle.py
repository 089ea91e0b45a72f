"""ctypes loader of oracle/libacmpc_oracle.so - the scalar C restatement (TEST INFRASTRUCTURE ONLY).

Used by tests (bit-for-bit cross-check of the NumPy oracle) and by bench.py's `cpu_baseline` leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libacmpc_oracle.so")


class Weights(C.Structure):
    _fields_ = [("q", C.c_float * 3), ("r", C.c_float * 2), ("qn", C.c_float * 3), ("ulo", C.c_float * 2),
                ("uhi", C.c_float * 2), ("tmin", C.c_float), ("wbound", C.c_float), ("dt", C.c_float),
                ("nn_back", C.c_int), ("nn_ahead", C.c_int)]


_lib = None


def load(build_if_missing: bool = True) -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) and build_if_missing:
            subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
        _lib = C.CDLL(LIB_PATH)
        fp = C.POINTER(C.c_float)
        for name in ("acmpc_oracle_rollout_spatial", "acmpc_oracle_rollout_temporal"):
            fn = getattr(_lib, name)
            fn.restype = None
            fn.argtypes = [fp, fp, fp, C.c_int, C.c_int64, C.c_int, C.POINTER(Weights), fp, fp, fp]
        _lib.acmpc_oracle_rollout_spatial_blocked.restype = None
        _lib.acmpc_oracle_rollout_spatial_blocked.argtypes = [fp, fp, fp, C.c_int64, C.c_int, C.POINTER(Weights), fp, fp]
        _lib.acmpc_oracle_rollout_spatial_batch.restype = None
        _lib.acmpc_oracle_rollout_spatial_batch.argtypes = [fp, fp, fp, C.c_int, C.c_int64, C.c_int, C.POINTER(Weights),
                                                            fp, fp]
        _lib.acmpc_oracle_argmin.restype = C.c_int64
        _lib.acmpc_oracle_argmin.argtypes = [fp, C.c_int64]
    return _lib


def make_weights(Q, R, QN, u_lo, u_hi, w_bound, dt=0.05, t_min=0.01, nn_window=None) -> Weights:
    w = Weights()
    w.q[:] = [float(np.float32(v)) for v in Q]
    w.r[:] = [float(np.float32(v)) for v in R]
    w.qn[:] = [float(np.float32(v)) for v in QN]
    w.ulo[:] = [float(np.float32(v)) for v in u_lo]
    w.uhi[:] = [float(np.float32(v)) for v in u_hi]
    w.tmin, w.wbound, w.dt = t_min, w_bound, dt
    w.nn_back, w.nn_ahead = (-1, -1) if nn_window is None else nn_window
    return w


def rollout(mode: int, x0, coef, U, layout: int, weights: Weights, return_states: bool = False):
    """mode 0: spatial, 1: temporal.  U is [N,n,2] (layout 0) or [n,2,N] (layout 1), float32."""
    lib = load()
    fp = C.POINTER(C.c_float)
    x0 = np.ascontiguousarray(x0, dtype=np.float32)
    coef = np.ascontiguousarray(coef, dtype=np.float32)
    U = np.ascontiguousarray(U, dtype=np.float32)
    N, n = (U.shape[0], U.shape[1]) if layout == 0 else (U.shape[2], U.shape[0])
    costs = np.empty(N, dtype=np.float32)
    viol = np.empty(N, dtype=np.float32)
    states = np.empty((N, n + 1, 3), dtype=np.float32) if return_states else None
    fn = lib.acmpc_oracle_rollout_spatial if mode == 0 else lib.acmpc_oracle_rollout_temporal
    fn(x0.ctypes.data_as(fp), coef.ctypes.data_as(fp), U.ctypes.data_as(fp), layout, N, n, C.byref(weights),
       costs.ctypes.data_as(fp), viol.ctypes.data_as(fp), states.ctypes.data_as(fp) if return_states else None)
    return (costs, viol, states) if return_states else (costs, viol)


def rollout_spatial_blocked(x0, coef, U_step_major, weights: Weights):
    """Vectorisable form of the mode-S rollout for U[n][2][N] (bit-identical to `rollout(0, ..., layout=1)`)."""
    lib = load()
    fp = C.POINTER(C.c_float)
    x0 = np.ascontiguousarray(x0, dtype=np.float32)
    coef = np.ascontiguousarray(coef, dtype=np.float32)
    U = np.ascontiguousarray(U_step_major, dtype=np.float32)
    n, N = U.shape[0], U.shape[2]
    costs = np.empty(N, dtype=np.float32)
    viol = np.empty(N, dtype=np.float32)
    lib.acmpc_oracle_rollout_spatial_blocked(x0.ctypes.data_as(fp), coef.ctypes.data_as(fp), U.ctypes.data_as(fp), N, n,
                                             C.byref(weights), costs.ctypes.data_as(fp), viol.ctypes.data_as(fp))
    return costs, viol


def rollout_spatial_batch(x0, coef, U_step_major, weights: Weights):
    """P problems in one call: x0 [P,3], coef [P,n,12], U [P,n,2,N] -> costs [P,N], viol [P,N]."""
    lib = load()
    fp = C.POINTER(C.c_float)
    x0 = np.ascontiguousarray(x0, dtype=np.float32)
    coef = np.ascontiguousarray(coef, dtype=np.float32)
    U = np.ascontiguousarray(U_step_major, dtype=np.float32)
    P, n, _, N = U.shape
    costs = np.empty((P, N), dtype=np.float32)
    viol = np.empty((P, N), dtype=np.float32)
    lib.acmpc_oracle_rollout_spatial_batch(x0.ctypes.data_as(fp), coef.ctypes.data_as(fp), U.ctypes.data_as(fp), P, N, n,
                                           C.byref(weights), costs.ctypes.data_as(fp), viol.ctypes.data_as(fp))
    return costs, viol


def argmin(costs: np.ndarray) -> int:
    costs = np.ascontiguousarray(costs, dtype=np.float32)
    return int(load().acmpc_oracle_argmin(costs.ctypes.data_as(C.POINTER(C.c_float)), costs.size))
