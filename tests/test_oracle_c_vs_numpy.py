"""The two restatements of the sampling composition (NumPy and C) must agree bit-for-bit (not GPU)."""
import numpy as np
import pytest

import acmpc_oracle as orc
import c_oracle
from test_support import make_problem


@pytest.mark.parametrize("track,H,N", [("monza", 20, 128), ("monza", 50, 777), ("nordschleife", 80, 300)])
@pytest.mark.parametrize("mode", [0, 1])
def test_c_matches_numpy(track, H, N, mode):
    prob = make_problem(orc, track, H, N, seed=H + N)
    cfg = prob["cfg"]
    w = c_oracle.make_weights(cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6)
    args = (prob["U"], cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6)
    if mode == 0:
        coef = orc.coefficients_spatial(prob["table"], prob["limits"].margin)
        want = orc.rollout_spatial(prob["x0"], coef, *args, dtype=np.float32, return_states=True)
        x0 = prob["x0"]
    else:
        coef = orc.coefficients_temporal(prob["table"], prob["limits"].margin)
        want = orc.rollout_temporal(prob["pose0"], coef, *args, 0.05, dtype=np.float32, return_states=True)
        x0 = prob["pose0"]
    for layout in (0, 1):
        U = prob["U"] if layout == 0 else np.ascontiguousarray(prob["U"].transpose(1, 2, 0))
        cost, viol, states = c_oracle.rollout(mode, x0, coef, U, layout, w, return_states=True)
        np.testing.assert_array_equal(cost, want[0])
        np.testing.assert_array_equal(viol, want[1])
        np.testing.assert_array_equal(states, want[2])
    assert c_oracle.argmin(want[0]) == orc.pick_best(want[0])[0]
    if mode == 0:  # the vectorised form bench.py times as the CPU baseline
        U_sm = np.ascontiguousarray(prob["U"].transpose(1, 2, 0))
        cost, viol = c_oracle.rollout_spatial_blocked(x0, coef, U_sm, w)
        np.testing.assert_array_equal(cost, want[0])
        np.testing.assert_array_equal(viol, want[1])
        costs, viols = c_oracle.rollout_spatial_batch(np.stack([x0, x0]), np.stack([coef, coef]), np.stack([U_sm, U_sm]), w)
        np.testing.assert_array_equal(costs[1], want[0])
        np.testing.assert_array_equal(viols[0], want[1])


def test_argmin_first_minimum_and_nonfinite():
    c = np.array([3.0, np.nan, 1.0, 1.0, np.inf, -np.inf], dtype=np.float32)
    assert orc.pick_best(c)[0] == 2 and c_oracle.argmin(c) == 2
    assert orc.pick_best(np.array([np.nan, np.inf], dtype=np.float32))[0] == 0


@pytest.mark.parametrize("window", [(2, 5), (0, 1), (3, 12)])
def test_windowed_nearest_search_c_matches_numpy_and_exhaustive(window):
    """Mode T with the search window: C == NumPy bit for bit; and on these inputs (progress < 1 waypoint per step)
    a (2, 5) window finds the same waypoint as the exhaustive scan at every step of every candidate."""
    prob = make_problem(orc, "monza", 50, 400, seed=11)
    if window != (2, 5):   # overflowing / NaN states: both restatements must go wrong the same way
        prob["U"][5, 3, 0] = np.nan
        prob["U"][9, 0, 1] = np.inf
        prob["U"][11, 10, 0] = 3.0e38
        prob["U"][13, 20, 0] = -1.0e30
    cfg = prob["cfg"]
    coef = orc.coefficients_temporal(prob["table"], prob["limits"].margin)
    args = (prob["pose0"], coef, prob["U"], cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"],
            prob["u_hi"], 1.0e6, 0.05)
    want = orc.rollout_temporal(*args, dtype=np.float32, return_states=True, nn_window=window)
    w = c_oracle.make_weights(cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6,
                              nn_window=window)
    cost, viol, states = c_oracle.rollout(1, prob["pose0"], coef, prob["U"], 0, w, return_states=True)
    np.testing.assert_array_equal(cost, want[0])
    np.testing.assert_array_equal(viol, want[1])
    np.testing.assert_array_equal(states, want[2])
    if window == (2, 5):
        full = orc.rollout_temporal(*args, dtype=np.float32, return_states=True)
        np.testing.assert_array_equal(want[3], full[3])  # nearest indices
        np.testing.assert_array_equal(want[0], full[0])
