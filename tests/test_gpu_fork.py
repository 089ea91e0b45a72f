"""The reference builds its MPC objects in the parent process and then forks the control process
(src/acmpc/control/controller.py:94-100,293-297; multiprocessing's default start method on Linux).  Everything the
parent does - build_mpc, the particle scorer's construction, even loading the library - must leave the GPU untouched,
and the forked child must be able to solve."""
import multiprocessing as mp
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import copy, multiprocessing as mp, os, sys
sys.path.insert(0, os.path.join(%(root)r, "ac-mpc_amd"))
import numpy as np
from acmpc_amd import workloads
from acmpc_amd.mpc import build_mpc
from acmpc_amd.particle_filter import ParticleScorer

cfg = copy.deepcopy(workloads.RACING_CONTROL["monza"])
cfg["speed_profile_constraints"]["v_max"] = 28.0
mpc = build_mpc(cfg, workloads.PlaceholderVehicle())                     # parent, before the fork
track = workloads.synthetic_track("monza")
scorer = ParticleScorer(dict(n_particles=100, score_distribution=dict(mean=0, sigma=10),
                             thresholds=dict(offset=10, rotation=90, minimum_particles=20, track_limit=20.0)),
                        dict(centre=track["centre"], left=track["left"], right=track["right"]))
path = workloads.reference_path_from_centreline(workloads.local_centreline(track, 0), 50)


def child(queue):
    try:
        mpc.get_control(path, offset=0.1)
        states = np.zeros((100, 3), dtype=np.float32)
        states[:, :2] = track["centre"][:100]
        est, _, _ = scorer.estimate_location(np.ones(100, dtype=np.float32), states)
        queue.put(("ok", int(mpc.infeasibility_counter), mpc.projected_control.shape, float(est[0])))
    except Exception as e:                                                   # noqa: BLE001
        queue.put(("error", repr(e)))


ctx = mp.get_context("fork")
results = []
for _ in range(2):                                                           # two children, one after the other
    q = ctx.Queue()
    p = ctx.Process(target=child, args=(q,))
    p.start()
    results.append(q.get(timeout=240))
    p.join(timeout=60)
    assert p.exitcode == 0, p.exitcode
print(results)
assert all(r[0] == "ok" and r[1] == 0 and r[2] == (2, 49) for r in results), results
'''


def test_objects_built_in_the_parent_solve_in_forked_children():
    proc = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout + proc.stderr
