"""Reference path cut out of the map on the device: nearest map point, 150 m window, vehicle frame,
resample to 500 float32 points, every 10th kept with widths linspace(10, 6, H) - against the NumPy statement
(`workloads.local_centreline`) followed by the reference's own downsample vectors' restatement (G11,
`orc.downsample_centreline`, pinned in tests/test_oracle_golden.py)."""
import copy

import numpy as np
import pytest

import acmpc_oracle as orc
from test_support import PlaceholderVehicle

pytestmark = pytest.mark.gpu


def _engine(track, n=49):
    from acmpc_amd import MODE_SPATIAL, Engine
    eng = Engine(mode=MODE_SPATIAL, max_problems=1, max_candidates=256, max_steps=n, step_cost=[1, 1, 0], r_term=[1, 1],
                 final_cost=[1, 0, 0], u_min=[0, -1], u_max=[50, 1], margin=1.0, wheelbase=2.65)
    eng.bind_map(track["centre"], track["spacing"])
    return eng


@pytest.mark.parametrize("name,H", [("monza", 50), ("silverstone", 50), ("nordschleife", 80), ("spa", 100)])
def test_window_kernel_matches_the_numpy_statement(name, H):
    from acmpc_amd import workloads
    track = workloads.synthetic_track(name)
    eng = _engine(track, n=H - 1)
    M = len(track["centre"])
    points = H * (500 // H)
    rng = np.random.default_rng(3)
    for index in [0, 1, M - 1, M - 150, M // 3] + list(rng.integers(0, M, 6)):
        for lateral in (0.0, 0.35):
            want = orc.downsample_centreline(workloads.local_centreline(track, int(index), lateral, points=points), H)
            got, first = eng.map_reference_path(H, map_index=int(index), lateral_offset=lateral, centreline_points=points)
            assert first == index
            # float32 positions: NumPy's matmul may fuse a multiply-add the kernel does not -> at most the last bit
            ulp = np.spacing(np.abs(want[:, :2]).astype(np.float32)).astype(np.float64)
            assert np.all(np.abs(got[:, :2] - want[:, :2]) <= ulp), (name, index)
            np.testing.assert_array_equal(got[:, 2], want[:, 2])          # widths linspace(10, 6, H)
            assert np.mean(got[:, :2] == want[:, :2]) > 0.98
    eng.close()


def test_nearest_map_point_is_the_kd_tree_answer():
    """Poses off the centre line: the window starts at the map point nearest to the pose - first minimum of the
    squared distance, the reference's KD-tree query semantics (G8, `orc.nearest_waypoint`)."""
    from acmpc_amd import workloads
    track = workloads.synthetic_track("silverstone")
    eng = _engine(track)
    rng = np.random.default_rng(9)
    centre = track["centre"]
    seeds = rng.integers(0, len(centre), 40)
    poses = centre[seeds] + rng.normal(0, 2.0, (40, 2))
    _, want = orc.nearest_waypoint(poses, centre)
    for pose, index in zip(poses, want):
        got, first = eng.map_reference_path(50, map_index=-1, pose=pose)
        assert first == index
        np.testing.assert_array_equal(got, eng.map_reference_path(50, map_index=int(index))[0])
    eng.close()


def test_get_control_at_equals_get_control_on_the_same_path():
    """The drop-in controller fed from the map on the device against the same controller fed the NumPy-built path."""
    from acmpc_amd import workloads
    from acmpc_amd.mpc import build_mpc
    track = workloads.synthetic_track("silverstone")
    cfg = copy.deepcopy(workloads.RACING_CONTROL["silverstone"])
    cfg["speed_profile_constraints"]["v_max"] = float(cfg["unlocalised_max_speed"])
    # (without the LQ candidate: with the path cut out of the map on the device the host plans for the previous tick's
    # problem, with the path handed in for this tick's - tests/test_gpu_restated_solve.py pins both; here the paths are compared)
    cfg.update(n_candidates=4096, lq_candidate=False)
    a, b = build_mpc(copy.deepcopy(cfg), PlaceholderVehicle()), build_mpc(copy.deepcopy(cfg), PlaceholderVehicle())
    a.bind_map(track)
    for i in range(30):
        index = (i * 6) % len(track["centre"])
        a.get_control_at(map_index=index, lateral_offset=0.1, offset=0.1)
        path = workloads.reference_path_from_centreline(workloads.local_centreline(track, index, 0.1), 50)
        b.get_control(path, offset=0.1)
        assert a.infeasibility_counter == 0 and b.infeasibility_counter == 0
        np.testing.assert_allclose(a.reference_coordinates, path, rtol=0, atol=2e-5)
        if np.array_equal(a.reference_coordinates, path):       # same float32 path -> same plan, bit for bit
            np.testing.assert_array_equal(a.projected_control, b.projected_control)
            np.testing.assert_array_equal(a.cum_time, b.cum_time)
        else:
            np.testing.assert_allclose(a.projected_control, b.projected_control, rtol=1e-3, atol=1e-4)


_REBIND_SCRIPT = r"""
import copy, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
from acmpc_amd import workloads
from acmpc_amd.mpc import build_mpc
from test_support import PlaceholderVehicle

def controller():
    cfg = copy.deepcopy(workloads.RACING_CONTROL["silverstone"])
    cfg["speed_profile_constraints"]["v_max"] = float(cfg["unlocalised_max_speed"])
    cfg.update(n_candidates=2048)
    return build_mpc(cfg, PlaceholderVehicle())

maps = [workloads.synthetic_track("silverstone"), workloads.synthetic_track("monza")]
assert len(maps[0]["centre"]) != len(maps[1]["centre"])
a = controller()
for track in maps + maps[:1]:           # A, then B of another length, then A again
    a.bind_map(track)
    fresh = controller()
    fresh.bind_map(track)
    for index in (5, 4000, len(track["centre"]) - 40):
        a.get_control_at(map_index=index)
        fresh.get_control_at(map_index=index)
        want = a._control_solver._engine.map_reference_path(50, map_index=index)[0]
        np.testing.assert_array_equal(a.reference_coordinates, want)
        np.testing.assert_array_equal(a.reference_coordinates, fresh.reference_coordinates)
print("rebind ok")
"""


@pytest.mark.parametrize("graph", ["0", "1"])
def test_rebinding_a_map_does_not_replay_a_stale_tick_graph(graph):
    """ACMPC_TICK_GRAPH=1 replays a captured graph whose kernel arguments hold the map's address, length and window size:
    binding another map must drop it (the switch is read once per process, hence the child process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("ACMPC_TICK_GRAPH", None)
    if graph == "1":
        env["ACMPC_TICK_GRAPH"] = "1"
    proc = subprocess.run([sys.executable, "-c", _REBIND_SCRIPT, os.path.join(root, "ac-mpc_amd"), os.path.join(root, "tests")],
                          capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0 and "rebind ok" in proc.stdout, (proc.stdout + proc.stderr)[-3000:]
