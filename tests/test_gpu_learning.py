"""End to end: plain PPO on the drop-in surface makes `Walker3DCustomEnv-v0` last longer -- observation, reward, termination, auto-reset, the
in-kernel Monitor and the in-place rollout writes all have to be right for that (tools/ppo_demo.py; the full run, return 9.5 -> 3 150 in 7
minutes, is profiles/r06_ppo_walker3d.jsonl).  130 iterations = 17 M env-steps, about ten seconds on one MI355X.  Needs a real MI355X: -m gpu."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ppo_on_the_trainer_surface_learns_to_stay_up(tmp_path):
    out = str(tmp_path / "ppo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ppo_demo.py"), "--iters", "130", "--minutes", "5", "--fixed-std", "--log-std", "-1.2",
                        "--out", out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in open(out + ".jsonl")]
    first, last = lines[0], lines[-1]
    assert first["mean_length"] < 40 and first["mean_return"] < 60          # untrained: the robots fall within ~23 steps
    assert last["iter"] == 130 and last["env_steps"] == 130 * 4096 * 32
    assert last["mean_length"] > 3 * first["mean_length"] and last["mean_return"] > 150, (first, last)     # measured: ~110 steps, return ~210
    assert os.path.exists(out + "_policy.npz")
