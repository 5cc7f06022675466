"""`mocca_envs_amd.multi`: one env batch behind several handles -- sub-batches of one GPU that step independently
(`SubBatchedVecEnv.step_async` / `wait`), shards of one process on several devices (`ShardedVecEnv`).  Every env's results must equal
the single handle's BIT FOR BIT (each env is its own world, /root/reference/mocca_envs/env_base.py:55; draws are keyed by the global
env id), including when the actions are computed from the observations on torch's current stream while the other sub-batches step --
the double-buffered trainer loop of INTEGRATION.md.  Needs a real MI355X: -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _policy(w):
    """A stand-in policy that is a function of the observation ROW by ROW with a fixed operation order (a GEMM would pick another tiling,
    and so another rounding, for 256 rows than for 1024): act_j = tanh(sum of three scaled observation entries)."""
    import torch
    idx, sc = w
    return lambda obs: torch.tanh(obs[:, idx[0]] * sc[0] + obs[:, idx[1]] * sc[1] + obs[:, idx[2]] * sc[2]).contiguous()


def _weights(obs_dim, act_dim, device, seed):
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    idx = torch.randint(0, obs_dim, (3, act_dim), generator=g).to(device)
    sc = (torch.randn(3, act_dim, generator=g) * 1.5).to(device)
    return idx, sc


@pytest.mark.parametrize("env_id,k,kw", [("Walker3DCustomEnv-v0", 2, {}), ("Walker3DCustomEnv-v0", 4, {"max_rows": 32}),
                                         ("Walker3DStepperEnv-v0", 2, {}), ("CassieEnv-v0", 2, {})])
def test_double_buffered_sub_batches_reproduce_the_single_handle(env_id, k, kw):
    import torch
    from mocca_envs_amd.multi import SubBatchedVecEnv
    from mocca_envs_amd.vec_env import VecEnv
    n, steps = (256, 12) if "Cassie" in env_id else (1024, 150)
    one = VecEnv(env_id, n, auto_reset=True, seed=5, terminal_obs=True, **kw)
    sub = SubBatchedVecEnv(env_id, n, sub_batches=k, auto_reset=True, seed=5, terminal_obs=True, **kw)
    if "Stepper" in env_id:
        cur = (torch.arange(n) % 10).float()
        one.set_env_params({"curriculum": cur}); sub.set_env_params({"curriculum": cur})      # per-env parameters reach every sub-batch
    policy = _policy(_weights(one.obs_dim, one.act_dim, "cuda", 2))
    o1 = one.reset().clone()
    o2 = sub.reset().clone()
    assert torch.equal(o1, o2)
    # the single handle, synchronous
    ref = []
    for t in range(steps):
        o, r, d, i = one.step(policy(one.obs))
        ref.append((o.clone(), r.clone(), d.clone(), i.clone(), one.terminal_obs.clone()))
    # the sub-batches, double-buffered: wait(i) -> policy on the current stream -> step_async(i); no host synchronisation anywhere
    got = [[None] * k for _ in range(steps)]
    for t in range(steps):
        for i in range(k):
            obs, rew, done, info = sub.wait(i)
            if t > 0:
                got[t - 1][i] = (obs.clone(), rew.clone(), done.clone(), info.clone(), sub.terminal_obs[sub.slices[i]].clone())
            sub.step_async(i, policy(obs))
    for i in range(k):
        obs, rew, done, info = sub.wait(i)
        got[steps - 1][i] = (obs.clone(), rew.clone(), done.clone(), info.clone(), sub.terminal_obs[sub.slices[i]].clone())
    torch.cuda.synchronize()
    n_done = 0
    for t in range(steps):
        for j, name in enumerate(("obs", "rew", "done", "info", "terminal_obs")):
            assert torch.equal(ref[t][j], torch.cat([got[t][i][j] for i in range(k)])), (t, name)
        n_done += int((ref[t][2] != 0).sum())
    assert n_done > 0 or "Cassie" in env_id          # the run crossed in-kernel auto-resets
    assert torch.equal(one.get_state(), sub.get_state()) and torch.equal(one.get_task()[:, :23], sub.get_task()[:, :23])
    # the synchronous convenience form gives the same next step
    a = policy(one.obs)
    o, r, d, _ = one.step(a)
    o2, r2, d2, _ = sub.step(a)
    assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(d, d2)
    one.close(); sub.close()


def test_sub_batch_setters_and_seed():
    import torch
    from mocca_envs_amd.multi import SubBatchedVecEnv, make_vec_env
    from mocca_envs_amd.vec_env import VecEnv
    n = 256
    one = VecEnv("Walker3DCustomEnv-v0", n, seed=3)
    sub = make_vec_env("Walker3DCustomEnv-v0", n, sub_batches=4, seed=3)
    assert isinstance(sub, SubBatchedVecEnv) and sub.n_parts == 4
    gain = torch.linspace(0.5, 1.2, n)
    ev = (torch.arange(n) % 2).float()
    for e in (one, sub):
        e.set_robot_params({"applied_gain": gain})
        e.evaluation_mode(ev)
        e.seed(11)
        e.reset()
    assert torch.equal(one.obs, sub.obs) and torch.equal(one.get_task()[:, :23], sub.get_task()[:, :23])
    a = torch.rand(n, 21, device="cuda") * 2 - 1
    for _ in range(30):
        one.step(a); sub.step(a)
    assert torch.equal(one.obs, sub.obs) and torch.equal(one.get_state(), sub.get_state())
    with pytest.raises(ValueError):
        SubBatchedVecEnv("Walker3DCustomEnv-v0", 100, sub_batches=3)
    with pytest.raises(ValueError):
        sub.step(a[:10])
    one.close(); sub.close()


@pytest.mark.parametrize("gather", [True, False])
def test_single_process_shards_reproduce_the_unsharded_batch(gather):
    """ShardedVecEnv: one handle + stream per listed device.  On a one-GPU box the same device is listed several times (the code path is
    the same: per-shard device guards, streams, cross-'device' copies); with more GPUs visible the shards spread over them."""
    import torch
    from mocca_envs_amd.multi import ShardedVecEnv
    from mocca_envs_amd.vec_env import VecEnv
    n, steps = 768, 120
    ndev = torch.cuda.device_count()
    devices = [d % ndev for d in range(max(3, min(ndev, 8)))] if ndev < 3 else list(range(min(ndev, 8)))
    while n % len(devices):
        devices = devices[:-1]
    one = VecEnv("Walker3DCustomEnv-v0", n, device=0, auto_reset=True, seed=9)
    sh = ShardedVecEnv("Walker3DCustomEnv-v0", n, devices=devices, auto_reset=True, seed=9, gather=gather)
    o1 = one.reset()
    o2 = sh.reset()
    cat = (lambda x: x) if gather else (lambda x: torch.cat([p.to("cuda:0") for p in x]))
    assert torch.equal(o1, cat(o2))
    w = _weights(one.obs_dim, one.act_dim, "cuda:0", 4)
    policy = _policy(w)
    n_done = 0
    for t in range(steps):
        a = policy(one.obs)
        o, r, d, i = one.step(a)
        o2, r2, d2, i2 = sh.step(a)                   # one tensor on device 0: its rows travel to their shards
        assert torch.equal(o, cat(o2)) and torch.equal(r, cat(r2)) and torch.equal(d, cat(d2)) and torch.equal(i, cat(i2)), t
        n_done += int((d != 0).sum())
    assert n_done > 0
    # per-shard action tensors + wait(): the asynchronous form
    acts = [_policy((w[0].to(e.device), w[1].to(e.device)))(e.obs) for e in sh.parts]
    for i, a in enumerate(acts):
        sh.step_async(i, a)
    one.step(torch.cat([a.to("cuda:0") for a in acts]))
    got = torch.cat([sh.wait(i)[0].to("cuda:0") for i in range(sh.n_parts)])
    assert torch.equal(one.obs, got)
    assert torch.equal(one.get_state(), sh.get_state())
    distinct = len(set(devices))
    print(f"\n[multi] ShardedVecEnv ran {len(devices)} shards on {distinct} distinct device(s) {sorted(set(devices))} of {ndev} visible"
          + ("" if distinct >= 2 else " -- the cross-device copies and device switches ran between IDENTICAL devices (one-GPU box)"))
    one.close(); sh.close()


def test_gather_on_a_device_that_holds_no_shard():
    """`gather_device` other than every shard's device: observations, rewards and done flags cross from each shard's GPU to a third one
    (multi.py _collect: peer-to-peer copy_ between the devices' current streams).  Needs >= 2 GPUs to mean anything -- with one visible it
    is reported as NOT EXERCISED rather than run between a device and itself, which the test above already does."""
    import torch
    from mocca_envs_amd.multi import ShardedVecEnv
    from mocca_envs_amd.vec_env import VecEnv
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip(f"{ndev} GPU visible: cross-device gather NOT EXERCISED on this box (multi-GPU scaling is unmeasured, DESIGN.md section 7)")
    shard_devs = list(range(1, ndev)) if ndev > 2 else [1]
    n = 96 * len(shard_devs)
    one = VecEnv("Walker3DCustomEnv-v0", n, device=0, auto_reset=True, seed=13)
    sh = ShardedVecEnv("Walker3DCustomEnv-v0", n, devices=shard_devs, auto_reset=True, seed=13, gather=True, gather_device=0)
    assert all(e.device.index != 0 for e in sh.parts) and sh.obs.device.index == 0
    assert torch.equal(one.reset(), sh.reset())
    policy = _policy(_weights(one.obs_dim, one.act_dim, "cuda:0", 5))
    for t in range(80):
        a = policy(one.obs)
        o, r, d, i = one.step(a)
        o2, r2, d2, i2 = sh.step(a)
        assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(d, d2) and torch.equal(i, i2), t
    assert torch.equal(one.get_state(), sh.get_state())
    print(f"\n[multi] gather on device 0 from shards on devices {shard_devs}: exact")
    one.close(); sh.close()


def test_bench_stagger_goes_through_the_product_api_and_keeps_its_parameters():
    """`bench.py --stagger` used to drop --curriculum (and every other handle parameter) when it rebuilt its sub-batches; it now
    drives SubBatchedVecEnv, whose set_param reaches every handle: a curriculum-9 Stepper line must show curriculum 9's reset fraction
    (~1.5 % per step), not curriculum 0's (~4.5 %)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fr = {}
    for cur in (0, 9):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--env-id", "Walker3DStepperEnv-v0", "--envs", "2048", "--stagger", "2",
                            "--curriculum", str(cur), "--steps", "50", "--warmup", "100", "--preroll", "600", "--preroll-seconds", "0", "--no-cpu-baseline"],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        assert out["config"]["pipelined"] is True and f"curriculum {cur}" in out["config"]["workload"]
        fr[cur] = out["config"]["reset_fraction_per_step"]
    print(fr)
    assert fr[9] < 0.6 * fr[0], fr
