"""Multi-GPU path on CPU: two gloo ranks own disjoint env shards, agree on the max time, and a shard
reproduces exactly the envs it owns of the unsharded batch (RNG keyed by global env id)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mocca_envs_amd import sharding


def test_env_range_and_throughput():
    assert sharding.env_range(0, 8, 4096) == (0, 4096)
    assert sharding.env_range(7, 8, 4096) == (7 * 4096, 8 * 4096)
    with pytest.raises(ValueError):
        sharding.env_range(8, 8, 4096)
    assert sharding.aggregate_throughput(4096, 8, 100, 2.0) == 4096 * 8 * 100 / 2.0
    assert sharding.max_over_ranks(1.5) == 1.5


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the slowest rank defines the job time
        t = sharding.max_over_ranks(1.0 + rank, dist)
        # each rank steps its own shard with the CPU oracle standing in for the GPU (same global-id keyed draws)
        from mocca_envs_amd import model as M
        from oracle.oracle import Oracle
        n = 4
        lo, hi = sharding.env_range(rank, world, n)
        # the oracle keys draws by local env index, so emulate the offset by creating the full batch and masking
        o = Oracle(M.compile_walker3d().to_bytes(), 0, n * world, "f32")
        obs = o.reset(seed=123)[lo:hi]
        gathered = [torch.zeros(n, obs.shape[1]) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(obs.copy()))
        dist.barrier()
        q.put((rank, t, [g.numpy() for g in gathered]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0][1] == out[1][1] == 2.0          # MAX over ranks
    a, b = out[0][2], out[1][2]
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)       # both ranks see the same gathered shards
    assert not np.array_equal(a[0], a[1])          # different shards, different episodes


def _bench(*argv, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` (no torchrun) spawns N ranks, rendezvous over gloo on 127.0.0.1, reports n_gpus = N and the
    whole-job rate over the SLOWEST rank's time (the dry run makes rank r take (r + 1) ms per step)."""
    out = _bench("--gpus", "2", "--steps", "40", "--warmup", "1", "--dry-run")
    assert out["n_gpus"] == 2 and out["dry_run"] and out["scaling"] == "weak"
    assert 2.0 <= out["ms_per_step"] < 6.0                                # rank 1's time (a 80 ms sleep; a loaded host oversleeps), not rank 0's
    assert abs(out["value"] - 2 * 4096 / (1e-3 * out["ms_per_step"])) < 1e-3 * out["value"]
    assert "x2" in out["config"]["parallelism"]
    # every rank's own numbers travel in the line; ms_per_step is the max of the per-rank list
    pr = out["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and len(pr["kernel_ms"]) == 2 and pr["kernel_ms"] == [1.0, 2.0]
    assert 1.0 <= pr["ms_per_step"][0] < 3.0 and out["ms_per_step"] == max(pr["ms_per_step"])
    # ... and which device each rank ran on (host, local index, name / uuid / PCI bus id where a GPU is behind it): a scaling line proves N GPUs
    assert [d["rank"] for d in pr["device"]] == [0, 1] and [d["local_index"] for d in pr["device"]] == [0, 1] and pr["distinct_devices"] == 2
    assert all(set(d) == {"rank", "host", "local_index", "name", "uuid", "pci_bus_id"} for d in pr["device"])
    out = _bench("--gpus", "1", "--steps", "2", "--dry-run", "--envs", "8192")
    assert out["n_gpus"] == 1 and out["config"]["envs_per_gpu"] == 8192
    # the contract's fields, and the untimed pre-roll is declared in the line (it is data preparation, not part of W or K)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in out, key
    assert out["steps"] == 2 and out["config"]["preroll_steps"] == 1000 and "preroll_seconds" in out["config"] and out["dtype"] == "f32" and out["vs_baseline"] is None
    assert _bench("--gpus", "1", "--steps", "2", "--dry-run", "--preroll", "7")["config"]["preroll_steps"] == 7


def test_a_slow_barrier_stays_outside_the_timed_window():
    """The start / stop rendezvous is a gloo barrier over TCP: every rank's clock stops at its own synchronize, BEFORE the stop
    barrier.  A barrier that takes 0.25 s (25 x the 10 ms window of this run) must not show up in ms_per_step or value."""
    out = _bench("--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run", "--test-barrier-delay", "0.25")
    assert 2.0 <= out["ms_per_step"] < 10.0, out["ms_per_step"]          # 5 steps x 2 ms on rank 1 (more on a loaded host); with the barrier inside: > 50 ms
    assert out["value"] > 2 * 4096 / 10e-3


def test_the_launcher_counts_gpus_without_a_hip_runtime():
    """`python bench.py --gpus N` is a parent that only spawns ranks: it must not import torch (which loads the HIP runtime)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.argv = ['bench.py', '--gpus', '2', '--steps', '1', '--dry-run']; sys.path.insert(0, %r); import bench; "
            "n = bench.visible_gpus(); assert isinstance(n, int) and n >= -1; "
            "rc = bench.self_launch(bench.parse_args()); "
            "assert 'torch' not in sys.modules, 'the launcher imported torch'; sys.exit(rc)" % root)
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_bench_as_a_torchrun_rank():
    """Under torch.distributed.run the process is one rank: WORLD_SIZE from the environment decides, a lone rank 0 of world 1."""
    out = _bench("--gpus", "1", "--steps", "2", "--dry-run", env=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"))
    assert out["n_gpus"] == 1


def test_bench_physics_variants_flip_exactly_the_documented_fields():
    """bench.py's sensitivity bracket: each variant blob differs from the compiled one in the law it names and in nothing else."""
    import bench
    from mocca_envs_amd.vec_env import compile_model_for
    base = compile_model_for(bench.ENV_ID)
    assert (base.limit_at_violation, base.friction_cone, base.warmstart) == (1, 1, 0.0) and 0 < base.slot_margin[0] < 0.01
    seen = {}
    for v in bench.PHYSICS_VARIANTS:
        m = bench.physics_variant_model(bench.ENV_ID, v)
        seen[v] = (m.limit_at_violation, round(float(m.slot_margin[0]), 4), m.friction_cone, round(float(m.warmstart), 2))
        assert (m.max_rows, m.max_contacts, m.n_iters, m.sweep_alternate) == (base.max_rows, base.max_contacts, base.n_iters, 0)
    b = (1, round(float(base.slot_margin[0]), 4), 1, 0.0)
    assert seen["limit_rows_from_predicted_gap"] == (0,) + b[1:]
    assert seen["absolute_2cm_margins"] == (1, 0.02, 1, 0.0)
    assert seen["pyramid_friction"] == (1, b[1], 0, 0.0)
    assert seen["warmstart_0.85"] == (1, b[1], 1, 0.85)
    assert seen["all_four"] == (0, 0.02, 0, 0.85)


def test_bench_pybullet_leg_reports_the_reference_or_says_why_not(monkeypatch, tmp_path):
    """bench.py cpu_baseline["pybullet"]: `import pybullet` is probed at run time.  Absent (this image): the row
    says so.  Present: the raw-PyBullet loop runs -- exercised here end to end against tests/fake_pybullet.py (PyBullet's API over the f64
    oracle), which is a stand-in and is labelled as one."""
    import sys
    import bench
    sys.modules.pop("pybullet", None)
    out = bench.pybullet_baseline()
    assert out["value"] is None and out["kind"] == "reference" and "pybullet not installed" in out["sample"]
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fake_pybullet import make_module
    fake = make_module()
    calls = {"step": 0, "disconnect": 0}
    step = fake.stepSimulation

    def counted():
        calls["step"] += 1
        step()

    fake.stepSimulation = counted
    fake.getEulerFromQuaternion = lambda q: (0.0, 0.0, 0.0)
    fake.disconnect = lambda *a: calls.__setitem__("disconnect", calls["disconnect"] + 1)
    out = bench.pybullet_baseline(n_steps=30, module=fake)
    assert calls == {"step": 30, "disconnect": 1}
    assert out["kind"] == "reference" and out["cores"] == 1 and out["unit"] == "env-steps/s" and out["value"] > 0 and "stand-in" in out["sample"]
    # installed but no assets: says which variable is missing instead of failing inside loadMJCF
    monkeypatch.setitem(sys.modules, "pybullet", fake)
    monkeypatch.setenv("MOCCA_REF_DATA", str(tmp_path))
    out = bench.pybullet_baseline()
    assert out["value"] is None and "MOCCA_REF_DATA" in out["sample"]


def test_bench_workload_bracket_names_its_workloads():
    import bench
    assert bench.WORKLOADS == ("uniform_0.3", "zero_actions", "pd_to_t_pose", "ppo_policy") and bench.PD_KP > 0 and bench.PD_KD > 0
    import numpy as np
    w = np.load(bench.PPO_POLICY)       # the trained policy the fourth workload runs: MLP 52-256-256-21 + observation statistics
    assert w["pi_0_weight"].shape == (256, 52) and w["pi_4_weight"].shape == (21, 256) and w["obs_mean"].shape == (52,) and str(w["env_id"]) == bench.ENV_ID
    a = bench.parse_args(["--no-workload-bracket", "--no-physics-bracket"])
    assert a.no_workload_bracket and a.no_physics_bracket
