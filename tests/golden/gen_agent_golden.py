#!/usr/bin/env python3
"""Golden vector for SURVEY.md 8(a) row a20: the steering target ElTuarMPC._process_yaw hands to its PID
(/root/reference/src/acmpc/agent.py:106-115), recorded by running the reference's own method.

The reference module imports under the inert stand-ins of gen_golden.py plus four more for what agent.py pulls in
(aci.interface, the dashboard / perception / map-maker processes, simple_pid); the PID itself is out of scope
(SURVEY section 2), so the fake `self` carries a "PID" that returns target - current: the method then returns exactly
its steering target.

    python tests/golden/gen_agent_golden.py        # rewrites tests/golden/agent_steer_target.npz
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/src")
import gen_golden  # noqa: E402


def main():
    gen_golden.install_stubs()

    def module(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    class Inert:
        def __init__(self, *a, **k):
            pass

    module("aci.interface", AssettoCorsaInterface=Inert)
    module("acmpc.dashboard")
    module("acmpc.dashboard.dashboard", DashBoardProcess=Inert)
    module("acmpc.perception.perception", Perceiver=Inert)
    module("acmpc.mapping")
    module("acmpc.mapping.map_maker", MapMaker=Inert)
    module("simple_pid", PID=Inert)
    from acmpc.agent import ElTuarMPC

    delta_max = gen_golden.DELTA_MAX
    yaw = np.concatenate([np.linspace(-0.6, 0.6, 41), [delta_max, -delta_max, 0.0, 1e-9, np.nextafter(delta_max, 1.0)]])
    current = np.random.default_rng(20).uniform(-1, 1, yaw.shape[0])
    out = np.empty_like(yaw)
    for i, (y, c) in enumerate(zip(yaw, current)):
        fake = SimpleNamespace(controller=SimpleNamespace(delta_max=delta_max), pose={"steering_angle": float(c)},
                               _steering_pid=lambda cur, target: target - cur)
        out[i] = ElTuarMPC._process_yaw(fake, float(y)) - 0.0
        assert fake.steering_command == out[i]
    # the method adds the PID's output to the current angle: with the stand-in PID that is c + (target - c)
    path = os.path.join(HERE, "agent_steer_target.npz")
    np.savez(path, yaw=yaw, current=current, delta_max=np.float64(delta_max), steering=out)
    print("wrote %s: %d cases" % (path, yaw.shape[0]))


if __name__ == "__main__":
    main()
