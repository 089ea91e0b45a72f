"""Shared test inputs: the reference's per-track `racing.control` blocks (configs/<track>.yaml:67-81) and the
builder-chosen placeholder vehicle (the reference's vehicle file is a network asset)."""
from types import SimpleNamespace

RACING = {
    "monza": dict(horizon=50, unlocalised_max_speed=28,
                  speed_profile_constraints=dict(v_min=8.0, v_max=84.0, a_min=-1.3, a_max=1.0, ay_max=5.5,
                                                 ki_min=0.005, end_velocity=14.0),
                  step_cost=[4.0e-3, 5.0e-2, 0.0], r_term=[1.0e-2, 10.0], final_cost=[1.0, 0.0, 0.1]),
    "spa": dict(horizon=50, unlocalised_max_speed=8.0,
                speed_profile_constraints=dict(v_min=5.0, v_max=84.0, a_min=-1.0, a_max=1.0, ay_max=4.0,
                                               ki_min=0.003, end_velocity=20.0),
                step_cost=[1.0e-3, 0.0, 0.0], r_term=[1.0e-2, 10.0], final_cost=[1.0, 0.0, 0.1]),
    "nordschleife": dict(horizon=50, unlocalised_max_speed=20,
                         speed_profile_constraints=dict(v_min=12.0, v_max=84.0, a_min=-1.0, a_max=1.0,
                                                        ay_max=3.0, ki_min=0.0, end_velocity=14.0),
                         step_cost=[2.0e-4, 0.0, 0.0], r_term=[1.0e-2, 10.0], final_cost=[1.0, 0.0, 0.1]),
    "silverstone": dict(horizon=50, unlocalised_max_speed=32.0,
                        speed_profile_constraints=dict(v_min=8.0, v_max=84.0, a_min=-1.0, a_max=1.0,
                                                       ay_max=5.0, ki_min=0.003, end_velocity=20.0),
                        step_cost=[2.0e-3, 5.0e-2, 0.0], r_term=[1.0e-2, 10.0], final_cost=[1.0, 0.0, 0.1]),
}

WHEELBASE, WIDTH, DELTA_MAX = 2.65, 1.94, 0.30


class PlaceholderVehicle:
    """Duck-type of ace.steering.SteeringGeometry as used at dynamics.py:11-13."""

    def __init__(self):
        self.vehicle_data = SimpleNamespace(wheelbase=WHEELBASE, width=WIDTH)

    def max_steering_angle(self):
        return DELTA_MAX
