#!/usr/bin/env python3
"""Build-container tool: the terrain of the planner envs as a data file of this package.

Walker3DPlannerEnv.create_terrain (env_locomotion.py:1015-1021) loads `data/objects/misc/height_field_map_0.npy` -- 128 x 128 heights,
4 grid points per metre -- from the reference's data directory.  This writes the same numbers (float32, the precision the stepper
computes in) to mocca_envs_amd/data/height_field_map_0.npz with the grid's size and scale.  Numbers only.
Re-run:  python tools/gen_height_field.py
"""
import os

import numpy as np

REF = "/root/reference/mocca_envs/data/objects/misc/height_field_map_0.npy"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mocca_envs_amd", "data", "height_field_map_0.npz")

if __name__ == "__main__":
    d = np.load(REF)
    assert d.shape == (128 * 128,)
    np.savez_compressed(OUT, heights=d.reshape(128, 128).astype(np.float32), scale=np.array(4.0), source=np.array("height_field_map_0.npy"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
