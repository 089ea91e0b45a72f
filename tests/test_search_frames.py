"""Mode T's verified nearest-waypoint search, checked on the CPU: the frames acmpc_set_paths tabulates (host code of the
C-ABI library, `acmpc_search_frames`) and the kernel's acceptance test (csrc/acmpc_device.h: nearest_verified_window),
restated here in float32 with exact fused multiply-adds.  Whatever the path and wherever the pose: when the test
ACCEPTS a window's first minimum, it is the first minimum over ALL waypoints (localiser.py:282-289 semantics) of the
float32 keys - the property that makes the windowed kernel bit-identical to the exhaustive one.  And it accepts often
enough to be worth having: nearly always on a racing line's neighbourhood, however far to the side the pose is."""
import numpy as np
import pytest

import acmpc_oracle as orc
from acmpc_amd import _capi
from test_support import make_problem

ACROSS_MAX = np.float32(32.0)


def _coef(xy):
    H = xy.shape[0]
    table = orc.construct_waypoints(np.column_stack([xy, np.full(H, 9.5)]))
    table[orc.ROW_V] = 20.0
    return orc.coefficients_temporal(table, 0.97)


def _keys(coef):
    # (the path's own frame: positions relative to waypoint 0, float32 differences - csrc/acmpc_device.h: start_temporal)
    x, y = coef[:, orc.CT_X] - coef[0, orc.CT_X], coef[:, orc.CT_Y] - coef[0, orc.CT_Y]
    return np.float32(-2.0) * x, np.float32(-2.0) * y, orc.fma32(y, y, x * x)


def _own_frame(coef, X, Y):
    return (np.asarray(X, dtype=np.float32) - coef[0, orc.CT_X]).astype(np.float32), \
           (np.asarray(Y, dtype=np.float32) - coef[0, orc.CT_Y]).astype(np.float32)


def _med3(a, b, c):
    return np.sort(np.stack([a, b, c]), axis=0)[1]


def _accepted(coef, frames, X, Y, lo):
    """The kernel's window search + acceptance test for poses (X, Y) [M] at window positions lo [M]."""
    width, _ = _capi.search_window()
    ka, kb, kc = _keys(coef)
    X, Y = _own_frame(coef, X, Y)
    idx = lo[:, None] + np.arange(width)[None, :]
    d = orc.fma32(Y[:, None], kb[idx], orc.fma32(X[:, None], ka[idx], kc[idx]))
    j = lo + np.argmin(d, axis=1)                      # first minimum of the window
    best = d[np.arange(len(lo)), j - lo]
    rows = frames.reshape(-1, 8)[lo]
    alpha = orc.fma32(rows[:, 0], X, orc.fma32(rows[:, 1], Y, rows[:, 2]))
    beta = orc.fma32(rows[:, 0], Y, orc.fma32(-rows[:, 1], X, rows[:, 3]))
    with np.errstate(invalid="ignore"):
        along = _med3(alpha, rows[:, 4] - alpha, np.zeros_like(alpha))
        across = _med3(np.abs(beta) - rows[:, 5], np.zeros_like(beta), np.full_like(beta, ACROSS_MAX))
    bound = np.fmin(orc.fma32(across, across, orc.fma32(along, along, rows[:, 7])), rows[:, 6])
    recovered = orc.fma32(Y, Y, orc.fma32(X, X, best))
    return j, np.abs(recovered) < bound


def _exhaustive(coef, X, Y):
    ka, kb, kc = _keys(coef)
    X, Y = _own_frame(coef, X, Y)
    d = orc.fma32(Y[:, None], kb[None, :], orc.fma32(X[:, None], ka[None, :], kc[None, :]))
    return np.argmin(d, axis=1)


def _paths():
    rng = np.random.default_rng(11)
    out = {}
    for seed in range(4):
        out["racing_%d" % seed] = make_problem(orc, "monza", 50, 4, seed=seed)["coords"][:, :2]
    up = [(0.0, 3.0 * i) for i in range(24)]
    for gap in (2.0, 6.0, 9.0, 20.0):   # out and back `gap` metres apart: waypoints far apart in index, close in space
        turn = [(gap / 2 + gap / 2 * np.cos(a), 69.0 + gap / 2 * np.sin(a)) for a in np.radians([135.0, 90.0, 45.0])]
        down = [(gap, 69.0 - 3.0 * i) for i in range(23)]
        out["fold_%g" % gap] = np.array(up + turn + down)
    t = np.linspace(0.0, 1.9 * np.pi, 50)
    out["loop"] = np.column_stack([25.0 * np.cos(t), 25.0 * np.sin(t)])   # almost closes on itself
    out["spiral"] = np.column_stack([(6.0 + 3.0 * t) * np.cos(2.5 * t), (6.0 + 3.0 * t) * np.sin(2.5 * t)])
    out["figure_eight"] = np.column_stack([40.0 * np.sin(t), 25.0 * np.sin(2.0 * t)])   # crosses itself
    walk = np.cumsum(rng.normal(0.0, 1.0, (50, 2)) + np.array([0.5, 1.0]), axis=0)
    out["random_walk"] = walk
    stall = np.column_stack([np.zeros(50), 3.0 * np.arange(50.0)])
    stall[20:29] = stall[20]   # nine coincident waypoints: a window whose chord has no length
    out["stall"] = stall
    out["far_from_origin"] = out["racing_1"] + np.array([4000.0, -2500.0])
    out["horizon_101"] = make_problem(orc, "monza", 101, 4, seed=5)["coords"][:, :2]
    out["shortest"] = np.column_stack([np.zeros(9), 3.0 * np.arange(9.0)])   # n = 8: the window is the whole path
    return out


def _poses(coef, rng, count):
    """Poses that matter: beside every stretch of the path at all lateral offsets, round the waypoints, and far away."""
    n = coef.shape[0]
    j = rng.integers(0, n, count)
    x, y, psi = coef[j, orc.CT_X], coef[j, orc.CT_Y], coef[j, orc.CT_PSI]
    lateral = rng.choice([0.3, 2.0, 6.0, 15.0, 40.0, 300.0], count) * rng.standard_normal(count)
    forward = rng.uniform(-4.0, 4.0, count)
    X = x + forward * np.cos(psi) - lateral * np.sin(psi)
    Y = y + forward * np.sin(psi) + lateral * np.cos(psi)
    return X.astype(np.float32), Y.astype(np.float32), j


@pytest.mark.parametrize("name", sorted(_paths()))
def test_an_accepted_index_is_the_exhaustive_one(name):
    xy = _paths()[name]
    coef = _coef(xy)
    n = coef.shape[0]
    width, back = _capi.search_window()
    frames = _capi.search_frames(coef[None])[0]
    assert frames.shape == (8 * (n - width + 1),) and (frames[7::8] < 0.0).all()
    rng = np.random.default_rng(len(name))
    X, Y, near = _poses(coef, rng, 20000)
    truth = _exhaustive(coef, X, Y)
    accepted_any = 0
    # every window a rollout could be in when the pose is near waypoint `near`: previous index anywhere within reach
    for shift in range(-width, width + 1):
        lo = np.clip(near + shift - back, 0, n - width)
        j, ok = _accepted(coef, frames, X, Y, lo)
        np.testing.assert_array_equal(j[ok], truth[ok], err_msg="%s: accepted a non-global minimum (shift %d)" % (name, shift))
        accepted_any += int(np.count_nonzero(ok))
    # and all windows at all: a pose nowhere near its window must not be accepted either
    for lo_fixed in range(n - width + 1):
        j, ok = _accepted(coef, frames, X[:4000], Y[:4000], np.full(4000, lo_fixed))
        np.testing.assert_array_equal(j[ok], truth[:4000][ok], err_msg="%s: window %d" % (name, lo_fixed))
    if name != "stall":
        assert accepted_any > 0


@pytest.mark.parametrize("seed", range(4))
def test_random_paths(seed):
    """Paths nobody designed: random walks with drift, arcs of random curvature and step, points thrown into a box,
    circles with jitter, gentle curves kilometres from the origin - 15 of each kind per seed."""
    rng = np.random.default_rng(9000 + seed)
    width, back = _capi.search_window()
    accepted = total = 0
    for trial in range(15):
        n1 = int(rng.integers(9, 120))
        kind = trial % 5
        if kind == 0:
            xy = np.cumsum(rng.normal(0, 1, (n1, 2)) * rng.uniform(0.1, 5) + rng.normal(0, 2, 2), axis=0)
        elif kind == 1:
            heading = np.cumsum(rng.normal(0, rng.uniform(0.01, 0.6), n1))
            xy = np.cumsum(np.column_stack([np.cos(heading), np.sin(heading)]) * rng.uniform(0.5, 4), axis=0)
        elif kind == 2:
            xy = rng.uniform(-30, 30, (n1, 2))
        elif kind == 3:
            turn = np.linspace(0, rng.uniform(1, 12), n1)
            xy = rng.uniform(3, 60) * np.column_stack([np.cos(turn), np.sin(turn)]) + rng.normal(0, 0.01, (n1, 2))
        else:
            heading = np.cumsum(rng.normal(0, 0.05, n1))
            xy = np.cumsum(np.column_stack([np.cos(heading), np.sin(heading)]) * 3, axis=0) + rng.uniform(-1e4, 1e4, 2)
        coef = _coef(xy.astype(np.float32).astype(np.float64))
        n = coef.shape[0]
        frames = _capi.search_frames(coef[None])[0]
        X, Y, near = _poses(coef, rng, 3000)
        truth = _exhaustive(coef, X, Y)
        for shift in (-5, -2, 0, 1, 4):
            lo = np.clip(near + shift - back, 0, n - width)
            j, ok = _accepted(coef, frames, X, Y, lo)
            np.testing.assert_array_equal(j[ok], truth[ok], err_msg="seed %d trial %d shift %d" % (seed, trial, shift))
            accepted += int(np.count_nonzero(ok))
            total += len(ok)
    assert accepted > 0.05 * total


def test_non_finite_poses_and_paths_are_never_accepted():
    coef = _coef(_paths()["racing_0"])
    n = coef.shape[0]
    width, _ = _capi.search_window()
    frames = _capi.search_frames(coef[None])[0]
    bad = np.array([np.nan, np.inf, -np.inf, 3.0e38, -3.0e38, 1.0e20], dtype=np.float32)
    X, Y = np.meshgrid(np.concatenate([bad, [0.0]]), np.concatenate([bad, [1.0]]))
    X, Y = X.ravel().astype(np.float32), Y.ravel().astype(np.float32)
    keep = ~(np.isfinite(X) & np.isfinite(Y) & (np.abs(X) < 1e6) & (np.abs(Y) < 1e6))
    for lo in (0, n // 2, n - width):
        with np.errstate(all="ignore"):
            _, ok = _accepted(coef, frames, X[keep], Y[keep], np.full(np.count_nonzero(keep), lo))
        assert not ok.any()
    broken = coef.copy()
    broken[7, orc.CT_X] = np.nan
    frames = _capi.search_frames(broken[None])[0]
    Xs, Ys, near = _poses(coef, np.random.default_rng(3), 2000)
    with np.errstate(all="ignore"):
        _, ok = _accepted(broken, frames, Xs, Ys, np.clip(near - 3, 0, n - width))
    assert not ok.any()


def test_the_yield_where_sampled_candidates_are():
    """On racing-line reference paths the test accepts nearly every pose within the corridor's neighbourhood AND far to
    the side of it (the case a ball round the winning waypoint loses): the fallback scan stays the exception."""
    width, back = _capi.search_window()
    rng = np.random.default_rng(2)
    for seed in range(6):
        coef = _coef(make_problem(orc, "monza", 50, 4, seed=seed)["coords"][:, :2])
        n = coef.shape[0]
        frames = _capi.search_frames(coef[None])[0]
        for reach, least in ((12.0, 0.995), (25.0, 0.8)):   # [m] to either side of the path
            j = rng.integers(1, n - 1, 20000)
            lateral = rng.uniform(-reach, reach, 20000)
            forward = rng.uniform(-1.5, 1.5, 20000)
            psi = coef[j, orc.CT_PSI]
            X = (coef[j, orc.CT_X] + forward * np.cos(psi) - lateral * np.sin(psi)).astype(np.float32)
            Y = (coef[j, orc.CT_Y] + forward * np.sin(psi) + lateral * np.cos(psi)).astype(np.float32)
            lo = np.clip(j - 1 - back, 0, n - width)   # the previous step's nearest waypoint was the one before
            _, ok = _accepted(coef, frames, X, Y, lo)
            assert ok.mean() > least, (seed, reach, ok.mean())


def test_frames_per_problem_are_independent():
    a, b = _coef(_paths()["racing_0"]), _coef(_paths()["fold_6"])
    both = _capi.search_frames(np.stack([a, b]))
    np.testing.assert_array_equal(both[0], _capi.search_frames(a[None])[0])
    np.testing.assert_array_equal(both[1], _capi.search_frames(b[None])[0])
    with pytest.raises(_capi.EngineError):
        _capi.search_frames(a[None, :5])


@pytest.mark.parametrize("name", ["racing_1", "far_from_origin"])
def test_the_key_picks_the_nearest_waypoint_wherever_the_path_is(name):
    """Not bit equality with the same arithmetic but NEAREST-NESS, against float64 distances: the key e = c + a X + b Y of
    a waypoint is |p - w|^2 - |p|^2, and in float32 it cancels catastrophically when |p| is large - given 4 km from the
    origin the same path once made it pick a waypoint up to 1.2 m farther than the nearest for 5 % of the poses within 8 m
    of it.  Mode T now works in the path's own frame (positions relative to waypoint 0: csrc/acmpc_device.h start_temporal),
    so the key resolves what it resolves in the vehicle frame wherever the caller put the path."""
    coef = _coef(_paths()[name])
    rng = np.random.default_rng(3)
    n = coef.shape[0]
    j = rng.integers(0, n, 50000)
    psi = coef[j, orc.CT_PSI].astype(np.float64)
    lateral, forward = rng.uniform(-8.0, 8.0, j.size), rng.uniform(-4.0, 4.0, j.size)
    X = (coef[j, orc.CT_X] + forward * np.cos(psi) - lateral * np.sin(psi)).astype(np.float32)
    Y = (coef[j, orc.CT_Y] + forward * np.sin(psi) + lateral * np.cos(psi)).astype(np.float32)
    picked = _exhaustive(coef, X, Y)
    wx, wy = coef[:, orc.CT_X].astype(np.float64), coef[:, orc.CT_Y].astype(np.float64)
    d = np.hypot(X.astype(np.float64)[:, None] - wx[None, :], Y.astype(np.float64)[:, None] - wy[None, :])
    excess = d[np.arange(j.size), picked] - d.min(axis=1)
    # (a pose on the bisector of two waypoints may go either way: within the key's resolution, ~1e-2 m^2 / (2 x 3 m))
    assert excess.max() < 5e-3, "%s: picked a waypoint %.3f m farther than the nearest" % (name, excess.max())
    assert np.count_nonzero(picked != d.argmin(axis=1)) < 0.002 * j.size
