"""Waypoint table of the controller: same surface as the reference's `ReferencePath`
(/root/reference/src/acmpc/control/paths.py:4-72) - a 7 x n float64 structure of arrays whose rows are
`[x, y, psi, kappa, ds, width, v]`, with a named view per row.  The row order is part of the C ABI
(`acmpc_set_paths`), so it is defined once here.
"""
from __future__ import annotations

import numpy as np

ROWS = ("xs", "ys", "psis", "kappas", "distances", "widths", "velocities")


def _row_view(index: int):
    def getter(self) -> np.ndarray:
        return self._reference_path[index]

    def setter(self, values) -> None:
        self._reference_path[index, :] = values

    return property(getter, setter)


class ReferencePath:
    """n waypoints; `path.xs`, `.ys`, `.psis`, `.kappas`, `.distances`, `.widths`, `.velocities` are writable
    views of the rows of `path.table` (shape 7 x n)."""

    def __init__(self, n_positions: int):
        self._n_positions = int(n_positions)
        self._reference_path = np.zeros((len(ROWS), self._n_positions))

    @classmethod
    def from_table(cls, table: np.ndarray) -> "ReferencePath":
        path = cls(table.shape[1])
        path._reference_path[:] = table
        return path

    @classmethod
    def adopt(cls, table: np.ndarray) -> "ReferencePath":
        """A path that takes `table` (7 x n float64, C-contiguous) as its storage without copying it: for a table
        nobody else writes to."""
        path = cls.__new__(cls)
        path._n_positions = table.shape[1]
        path._reference_path = table
        return path

    @property
    def table(self) -> np.ndarray:
        return self._reference_path

    def __len__(self) -> int:
        return self._n_positions

    def get_state(self, index: int) -> np.ndarray:
        """[x, y, psi] of waypoint `index` (paths.py:68-72)."""
        return self._reference_path[:3, index]


for _i, _name in enumerate(ROWS):
    setattr(ReferencePath, _name, _row_view(_i))
