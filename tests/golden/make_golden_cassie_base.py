#!/usr/bin/env python3
"""Fixture for tests/test_golden_cassie_mocap.py::test_stance_foot_*: the floating-base pose of every frame of the reference's Cassie
walking cycle (build container only -- it reads /root/reference).

  /root/reference/mocca_envs/data/robots/cassie/mocap/stepdata.bin   1682 x 98 float64: time (1), qpos (35), qvel (32), ...
                                                                     qpos[0:3] pelvis position, qpos[3:7] pelvis quaternion (w, x, y, z)

-> tests/golden/cassie_mocap_base.npz: time[1682], base_pos[1682][3], base_quat_wxyz[1682][4] (numbers only, 94 KB).
The joint angles of the same frames are product data already (mocca_envs_amd/data/cassie_mocap.npz, tools/gen_cassie_mocap.py)."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sd = np.fromfile("/root/reference/mocca_envs/data/robots/cassie/mocap/stepdata.bin", dtype="<f8").reshape(-1, 98)
np.savez_compressed(os.path.join(HERE, "cassie_mocap_base.npz"), time=sd[:, 0], base_pos=sd[:, 1:4], base_quat_wxyz=sd[:, 4:8])
print("frames", len(sd), "z range", sd[:, 3].min(), sd[:, 3].max())
