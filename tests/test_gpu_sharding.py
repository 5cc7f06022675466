"""The sharded HIP path with real processes: two ranks, each its own process and its own libmocca_hip handle with
env_offset = rank x envs, reproduce bit for bit the envs they own of the unsharded batch (draws are keyed by the global
env id) -- the property that makes `bench.py --gpus N` a weak-scaling run of ONE job rather than N unrelated ones.
On a 1-GPU box both ranks share the device.  Needs a real MI355X: -m gpu."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from mocca_envs_amd.vec_env import VecEnv
rank, world, n, out = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
dev = rank % torch.cuda.device_count()
torch.cuda.set_device(dev)
env = VecEnv("Walker3DCustomEnv-v0", n, device=dev, auto_reset=True, seed=77, env_offset=rank * n)
obs = [env.reset().cpu().numpy().copy()]
acts = np.random.default_rng(5).uniform(-1, 1, (60, world * n, 21)).astype(np.float32)[:, rank * n:(rank + 1) * n]
rew, done = [], []
for k in range(60):
    o, r, d, _ = env.step(torch.from_numpy(acts[k]).to(env.device))
    obs.append(o.cpu().numpy().copy()); rew.append(r.cpu().numpy().copy()); done.append(d.cpu().numpy().copy())
np.savez(out, obs=np.array(obs), rew=np.array(rew), done=np.array(done), state=env.get_state().cpu().numpy())
"""


def test_two_process_shards_reproduce_the_unsharded_batch(tmp_path):
    n = 96
    outs = [str(tmp_path / f"r{r}.npz") for r in range(2)] + [str(tmp_path / "full.npz")]
    jobs = [(0, 2, n, outs[0]), (1, 2, n, outs[1]), (0, 1, 2 * n, outs[2])]
    procs = [subprocess.Popen([sys.executable, "-c", WORKER, ROOT, str(r), str(w), str(k), o]) for r, w, k, o in jobs]
    for p in procs:
        assert p.wait(timeout=600) == 0
    a, b, full = (np.load(o) for o in outs)
    for key in ("obs", "rew", "done"):
        np.testing.assert_array_equal(np.concatenate([a[key], b[key]], axis=1), full[key], err_msg=key)
    np.testing.assert_array_equal(np.concatenate([a["state"], b["state"]], axis=0), full["state"])
    assert int((full["done"] != 0).sum()) > 0          # the run crossed in-kernel auto-resets (global-id keyed draws)


def test_bench_two_ranks_on_this_box():
    """bench.py's own N-rank launch on real hardware (ranks share the GPU when the box has one): one JSON line, n_gpus = 2."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--envs", "512", "--oversubscribe", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["envs_per_gpu"] == 512
    # every rank names the GPU it ran on; the line counts the distinct ones (1 on a one-GPU box: marked oversubscribed, never a scaling point)
    devs = out["per_rank"]["device"]
    assert len(devs) == 2 and all(d["name"] and (d["uuid"] or d["pci_bus_id"]) for d in devs)
    import torch
    assert out["per_rank"]["distinct_devices"] == min(2, torch.cuda.device_count()) and out["oversubscribed"]
    if torch.cuda.device_count() == 1:     # without --oversubscribe the same launch refuses to call itself a 2-GPU run
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--envs", "256",
                            "--no-cpu-baseline", "--preroll", "10", "--preroll-seconds", "0"], capture_output=True, text=True, timeout=900)
        assert r.returncode != 0


@pytest.mark.gpu
def test_sub_batches_on_their_own_streams_reproduce_the_single_batch():
    """bench.py --stagger: a GPU's batch as two handles bound to two HIP streams (`VecEnv.stream`), stepping without
    waiting for each other -- every env's results equal the single handle's bit for bit (draws are keyed by the global env id), on the
    48-row and on the compact kernel instance."""
    import torch
    from mocca_envs_amd.vec_env import VecEnv
    n, steps = 1024, 60
    for kw in ({}, {"max_rows": 32}):
        one = VecEnv("Walker3DCustomEnv-v0", n, auto_reset=True, seed=5, **kw)
        one.reset()
        halves = []
        for k in range(2):
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                e = VecEnv("Walker3DCustomEnv-v0", n // 2, auto_reset=True, seed=5, env_offset=k * (n // 2), **kw)
                e.reset()
            e.stream = st
            halves.append(e)
        torch.cuda.synchronize()
        g = torch.Generator(device="cuda"); g.manual_seed(2)
        acts = torch.rand(steps, n, 21, device="cuda", generator=g) * 2 - 1
        parts = [acts[:, :n // 2].contiguous(), acts[:, n // 2:].contiguous()]
        torch.cuda.synchronize()
        for t in range(steps):
            one.step(acts[t])
            for k, e in enumerate(halves):
                e.step(parts[k][t])            # no synchronisation between the halves, nor with the single batch
        torch.cuda.synchronize()
        nd = 13 + 2 * 21
        both = torch.cat([halves[0].get_state(), halves[1].get_state()])
        assert torch.equal(one.get_state()[:, :nd], both[:, :nd])
        assert torch.equal(one.obs, torch.cat([halves[0].obs, halves[1].obs])) and torch.equal(one.done, torch.cat([halves[0].done, halves[1].done]))
        for e in halves + [one]:
            e.close()


@pytest.mark.gpu
def test_default_bench_line_carries_both_brackets_and_the_baselines():
    """`python bench.py` as the driver runs it (smaller batch here): ONE JSON line with `roofline`, `cpu_baseline` (incl. the `pybullet` leg's
    answer), the physics-law `sensitivity` bracket and the behaviour `workload_sensitivity` bracket, whose closed-loop workloads must replay
    exactly from their snapshots."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--envs", "512", "--steps", "10", "--warmup", "3", "--preroll", "50",
                        "--preroll-seconds", "0"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["pybullet"]["kind"] == "reference" and ("n/a" in cb["pybullet"]["sample"] or cb["pybullet"]["value"] > 0)
    assert set(out["sensitivity"]["variants"]) == {"as_built", "limit_rows_from_predicted_gap", "absolute_2cm_margins", "pyramid_friction", "warmstart_0.85", "all_four"}
    ws = out["workload_sensitivity"]
    assert set(ws["workloads"]) == {"uniform_0.3", "zero_actions", "pd_to_t_pose", "ppo_policy"} and ws["range"][0] <= ws["range"][1]
    assert ws["workloads"]["ppo_policy"]["reset_fraction_per_step"] < 0.01           # the trained policy walks: episodes run into the TimeLimit (which falls on the window's last step: 1 / 200)
    for name, w in ws["workloads"].items():
        assert w["replay_exact"] and w["value"] > 0 and 0 <= w["reset_fraction_per_step"] < 0.2, name
    assert ws["workloads"]["pd_to_t_pose"]["reset_fraction_per_step"] < ws["headline"]["reset_fraction_per_step"]      # the controller keeps robots up longer
    assert out["per_rank"]["device"][0]["name"] and out["per_rank"]["distinct_devices"] == 1
