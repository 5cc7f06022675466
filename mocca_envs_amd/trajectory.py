"""CassieTrajectory -- the reference motion of the Cassie mocap / phase envs, re-created.

`env_cassie.py:10` imports `CassieTrajectory` from a module `loadstep` that is not in the reference tree (SURVEY.md
section 0.5); its call sites fix the interface:

    traj.joint_angles(t)      env_cassie.py:601-602   14 angles, order of Cassie.ordered_joints
    traj.joint_speeds(t)      env_cassie.py:604-605   14 speeds
    traj.rod_joint_angles(t)  env_cassie.py:589-599   right z, right y, left z, left y
    traj.max_time()           env_cassie.py:639       period of the cycle

and the data it read is still there (data/robots/cassie/mocap/).  tools/gen_cassie_mocap.py turns those two files into
`data/cassie_mocap.npz`; this class serves it.  Time lookup follows the convention of the cycle's origin (the OSU
`CassieTrajectory.state`): frame `int((t mod T) / T * n_frames)`, no interpolation -- a decision of this re-creation, like
the rod angles (loop-closing least squares on this project's model), documented in DESIGN.md section 3.
"""
from __future__ import annotations

import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cassie_mocap.npz")
TABLE_STRIDE = 32   # floats per frame of the device table: 14 angles, 14 speeds, 4 rod angles


class CassieTrajectory:
    def __init__(self, path: str = _DATA):
        d = np.load(path)
        self.time = d["time"].astype(np.float64)
        self.angles = d["joint_angles"].astype(np.float64)
        self.speeds = d["joint_speeds"].astype(np.float64)
        self.rods = d["rod_angles"].astype(np.float64)
        self.rod_bodies = d["rod_bodies"].astype(np.int32)

    def __len__(self) -> int:
        return len(self.time)

    def max_time(self) -> float:
        return float(self.time[-1])

    def index(self, t: float) -> int:
        tmax = self.max_time()
        i = int((t % tmax) / tmax * len(self.time))
        return min(i, len(self.time) - 1)

    def joint_angles(self, t: float) -> np.ndarray:
        return self.angles[self.index(t)].copy()

    def joint_speeds(self, t: float) -> np.ndarray:
        return self.speeds[self.index(t)].copy()

    def rod_joint_angles(self, t: float) -> np.ndarray:
        return self.rods[self.index(t)].copy()

    def table(self) -> np.ndarray:
        """[n_frames][32] float32 for mocca_set_trajectory: angles 0..13, speeds 14..27, rod angles 28..31."""
        return np.concatenate([self.angles, self.speeds, self.rods], axis=1).astype(np.float32)
