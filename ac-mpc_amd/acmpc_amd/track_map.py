"""Map ingestion: the reference's map file format and the two small reductions the
agent performs on it, so that real AARK maps can feed the engine when they are available.

* `load_track_map` / `remove_near_duplicate_points`: /root/reference/src/acmpc/utils/load.py:9-35,61-65 - a map is a
  pickled dict in a `.npy` (or a JSON file) with `outside_track` / `inside_track` / `centre_track` polylines; the
  loader renames them left / right / centre and drops points closer than 0.1 mm to their predecessor.
* `lap_reference_path`: agent.py:286-296 - the centre line with the constant 9.5 m road width, ready for
  `SpatialMPC.construct_waypoints` / `compute_map_speed_profile`.
* `reference_speed_window`: agent.py:24-28,137-143 - the localised v_max = mean of the lap speed profile over
  [index - 25, index + 75) with wrap-around.
"""
from __future__ import annotations

import json
from typing import Dict

import numpy as np

REFERENCE_SPEED_WINDOW_BEHIND = 25   # agent.py:24-28
REFERENCE_SPEED_WINDOW_AHEAD = 75
ROAD_WIDTH = 9.5                     # agent.py:288
DUPLICATE_DISTANCE = 1.0e-4          # load.py:34


def remove_near_duplicate_points(track: np.ndarray) -> np.ndarray:
    """Keep the first point and every point farther than 0.1 mm from its predecessor."""
    step = np.diff(track, axis=0)
    keep = np.concatenate([[True], np.hypot(step[:, 0], step[:, 1]) > DUPLICATE_DISTANCE])
    return track[keep]


def _read(path: str) -> Dict[str, np.ndarray]:
    if path.endswith(".npy"):
        return np.load(path, allow_pickle=True).item()
    if path.endswith(".json"):
        with open(path) as handle:
            raw = json.load(handle)
        return {"centre_track": np.array(raw["Centre"]), "outside_track": np.array(raw["Outside"]),
                "inside_track": np.array(raw["Inside"])}
    raise ValueError("unsupported map file %r (expected .npy or .json)" % path)


def load_track_map(path: str) -> Dict[str, np.ndarray]:
    """{"left", "right", "centre"} polylines, de-duplicated - what `utils.load.track_map` returns."""
    raw = _read(path)
    return {name: remove_near_duplicate_points(np.asarray(raw[key]))
            for name, key in (("left", "outside_track"), ("right", "inside_track"), ("centre", "centre_track"))}


def lap_reference_path(centre_track: np.ndarray, road_width: float = ROAD_WIDTH) -> np.ndarray:
    """M x 3 `[x, y, width]` for the whole lap."""
    return np.column_stack([centre_track[:, 0], centre_track[:, 1], np.full(len(centre_track), road_width)])


def reference_speed_window(reference_speeds: np.ndarray, centre_index: int) -> float:
    """Mean lap speed over [index - 25, index + 75), indices wrapping round the lap."""
    idx = np.arange(centre_index - REFERENCE_SPEED_WINDOW_BEHIND, centre_index + REFERENCE_SPEED_WINDOW_AHEAD)
    return float(np.mean(reference_speeds.take(idx, mode="wrap")))
