"""`mocca_envs_amd.trainer_api`: the VecPyTorch-shaped surface SymmetricRL / ALLSTEPS drive (/root/reference/README.md:33-39), over one VecEnv.
Checked against a plain VecEnv stepped beside it with the same seed and actions: observations and rewards bit for bit; Monitor's episode
return / length, TimeLimitMask's `bad_transition` and the Stepper's `steps_reached` (env_locomotion.py:562-566) -- all produced INSIDE the
step kernel since ABI 7 (include/mocca.h mocca_set_episode_stats) -- against a hand-kept ledger.  Then the two things the lazy surface
promises: records read many steps late are still the right ones, and `policy -> mocca_step` replays bit-identically from a CUDA graph.
Needs a real MI355X: -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env_id,sub_batches", [("Walker3DCustomEnv-v0", 1), ("Walker3DStepperEnv-v0", 2), ("Walker2DCustomEnv-v0", 1), ("CassieEnv-v0", 1)])
def test_trainer_surface_matches_a_hand_kept_ledger(env_id, sub_batches):
    import torch
    from mocca_envs_amd.trainer_api import make_vec_envs
    from mocca_envs_amd.vec_env import VecEnv
    n = 64
    steps = 1100 if "2D" in env_id else (60 if "Cassie" in env_id else 260)   # Walker2DCustomEnv never terminates (env_locomotion.py:302-309): only the TimeLimit ends its episodes
    envs = make_vec_envs(env_id, seed=4, num_processes=n, log_dir=None, sub_batches=sub_batches, terminal_observation=True)
    ref = VecEnv(env_id, n, auto_reset=True, seed=4, terminal_obs=True)
    assert envs.num_envs == n and envs.observation_space.shape == (ref.obs_dim,) and envs.action_space.shape == (ref.act_dim,)
    assert float(envs.action_space.high.max()) == 1.0 and float(envs.action_space.low.min()) == -1.0
    obs = envs.reset()
    assert torch.equal(obs, ref.reset()) and obs.dtype == torch.float32 and obs.is_cuda
    g = torch.Generator(device="cuda").manual_seed(9)
    ret, length = np.zeros(n), np.zeros(n, int)
    n_eps = n_bad = 0
    tot = np.zeros(4)
    for t in range(steps):
        a = torch.rand(n, ref.act_dim, device="cuda", generator=g) * 2 - 1
        obs, rew, done, infos = envs.step(a)
        o2, r2, d2, i2 = ref.step(a)
        assert torch.equal(obs, o2) and torch.equal(rew, r2.unsqueeze(1)) and rew.shape == (n, 1)
        assert torch.equal(envs.done, d2)                                   # the device-side done byte (bit0 terminated, bit1 TimeLimit)
        dn = np.asarray(done)
        assert dn.dtype == bool and dn.shape == (n,) and (dn == (d2.cpu().numpy() != 0)).all() and len(done) == n
        assert bool(done.any()) == bool(dn.any()) and [bool(x) for x in done] == dn.tolist() and done[3] == dn[3]
        assert len(infos) == n
        ret += r2.cpu().numpy(); length += 1
        d2h, i2h = d2.cpu().numpy(), i2.cpu().numpy()
        masks, bad = envs.masks.cpu().numpy()[:, 0], envs.bad_masks.cpu().numpy()[:, 0]
        assert envs.masks.shape == (n, 1) and envs.bad_masks.shape == (n, 1)
        for i, info in enumerate(infos):            # the trainers' own loop over the N dicts
            if done[i]:
                assert abs(info["episode"]["r"] - ret[i]) < 1e-3 * (1 + abs(ret[i])) and info["episode"]["l"] == length[i]
                # a2c-ppo-acktr's TimeLimitMask: bad_transition on ANY done at max_episode_steps, also when the env terminates in that very step
                timeout = bool(d2h[i] & 2)
                assert ("bad_transition" in info) == timeout and masks[i] == 0.0 and bad[i] == (0.0 if timeout else 1.0)
                assert torch.equal(info["terminal_observation"], ref.terminal_obs[i])
                if "Stepper" in env_id:
                    assert info["steps_reached"] == i2h[i]
                n_eps += 1; n_bad += timeout
                tot += (ret[i], length[i], 1, timeout)
                ret[i], length[i] = 0.0, 0
            else:
                assert info == {} and masks[i] == 1.0 and bad[i] == 1.0
        assert sorted(k for k, _ in infos.finished()) == list(np.nonzero(dn)[0])
        e = infos.episodes()                       # the same records as arrays
        assert e["env"].tolist() == np.nonzero(dn)[0].tolist() and e["l"].tolist() == [infos[i]["episode"]["l"] for i in e["env"]]
        assert e["r"].tolist() == [infos[i]["episode"]["r"] for i in e["env"]] and e["truncated"].tolist() == [bool(d2h[i] & 2) for i in e["env"]]
    assert n_eps > n // 2
    if "2D" in env_id:
        assert n_bad == n_eps > 0                   # every Walker2D episode ends by the TimeLimit
    dev_tot = envs.episode_totals.cpu().numpy()     # the on-device Monitor: sums over every episode that ended
    assert dev_tot[2] == n_eps and dev_tot[3] == n_bad and dev_tot[1] == tot[1] and abs(dev_tot[0] - tot[0]) < 1e-3 * (1 + abs(tot[0]))
    # the curriculum and mirror calls the trainers make
    if "Stepper" in env_id:
        envs.set_env_params({"curriculum": 5})
        assert envs.env_method("set_env_params", {"curriculum": 7}) == [None] * n
    if "2D" not in env_id and "Cassie" not in env_id:
        assert len(envs.get_mirror_indices()) == 6
    envs.close(); ref.close()


@pytest.mark.parametrize("events", [True, False])
def test_records_read_late_are_still_the_right_ones(events):
    """`done` / `infos` are lazy: nothing is fetched until the trainer looks.  Kept for 30 steps (the ring holds 4) and read afterwards they
    must equal what an eager twin saw step by step; `eager_done=True` hands out real numpy arrays.  With `record_events=False` a look waits for
    everything issued so far instead of for its own step's event: same answers."""
    import torch
    from mocca_envs_amd.trainer_api import make_vec_envs
    n, steps = 128, 90
    lazy = make_vec_envs("Walker3DCustomEnv-v0", seed=2, num_processes=n, record_slots=4, record_events=events)
    eager = make_vec_envs("Walker3DCustomEnv-v0", seed=2, num_processes=n, eager_done=True)
    lazy.reset(); eager.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    kept, seen = [], []
    for t in range(steps):
        a = torch.rand(n, 21, device="cuda", generator=g) * 2 - 1
        _, _, d1, i1 = lazy.step(a)
        _, _, d2, i2 = eager.step(a)
        assert isinstance(d2, np.ndarray) and d2.dtype == bool
        seen.append((d2.copy(), {k: dict(v) for k, v in i2.finished()}))
        kept.append((d1, i1))
        if len(kept) == 30:
            for (dl, il), (de, ie) in zip(kept, seen):
                assert (np.asarray(dl) == de).all() and dict(il.finished()) == ie
            kept, seen = [], []
    assert sum(int(np.asarray(d).sum()) for d, _ in kept) >= 0
    lazy.close(); eager.close()


def test_policy_and_step_replay_from_a_cuda_graph():
    """The collection loop without the host: `policy -> mocca_step` captured once in a torch.cuda.CUDAGraph (default parameters: the pace
    calibrates itself on the device since ABI 7) and replayed; observations, rewards, done bytes, masks, episode totals and the
    simulation state must equal, bit for bit, those of an eager twin that launched every kernel from Python."""
    import torch
    from mocca_envs_amd.vec_env import VecEnv
    n, warm, steps = 256, 3, 120
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(5)
    w1 = torch.randn(52, 64, device=dev, generator=g) * 0.3
    w2 = torch.randn(64, 21, device=dev, generator=g) * 0.3

    def policy(o):
        return torch.tanh(torch.tanh(o @ w1) @ w2)

    envs = []
    for _ in range(2):
        e = VecEnv("Walker3DCustomEnv-v0", n, auto_reset=True, seed=6)
        e.episode_stats(True)
        e.reset()
        envs.append(e)
    eager, graphed = envs
    for _ in range(warm + steps):
        eager.step(policy(eager.obs))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                   # torch wants the captured ops to have run once on a side stream
        for _ in range(warm):
            graphed.step(policy(graphed.obs))
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        graphed.step(policy(graphed.obs))           # obs is read and rewritten in place: the graph is the whole loop body
    for _ in range(steps):                          # (capturing records the launches, it does not run them)
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(eager.obs, graphed.obs) and torch.equal(eager.rew, graphed.rew) and torch.equal(eager.done, graphed.done)
    assert torch.equal(eager.get_state(), graphed.get_state()) and torch.equal(eager.get_task(), graphed.get_task())
    assert torch.equal(eager.ep["masks"], graphed.ep["masks"]) and torch.equal(eager.ep["bad_masks"], graphed.ep["bad_masks"])
    te, tg = eager.ep["totals"].cpu().numpy(), graphed.ep["totals"].cpu().numpy()
    assert te[2] > 0 and te[1] == tg[1] and te[2] == tg[2] and te[3] == tg[3] and abs(te[0] - tg[0]) < 1e-3 * (1 + abs(te[0]))   # (atomic float sums: order differs)
    for e in envs:
        e.close()


def test_capture_rollout_is_the_eager_rollout():
    """`TorchVecEnv.capture_rollout`: 16 x {policy, step, sink} as one graph; three replays fill the trainer's storage with exactly what 48 eager
    steps of a twin produce, and the device totals count the same episodes."""
    import torch
    from mocca_envs_amd.trainer_api import make_vec_envs
    n, T, warm = 192, 16, 2
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(8)
    w = torch.randn(52, 21, device=dev, generator=g) * 0.4
    policy = lambda o: torch.tanh(o @ w)
    a, b = (make_vec_envs("Walker3DCustomEnv-v0", seed=12, num_processes=n, record_events=False) for _ in range(2))
    store = {k: torch.zeros(T, n, d, device=dev) for k, d in (("obs", 52), ("rew", 1), ("masks", 1), ("bad", 1), ("act", 21))}

    def sink(t, obs, rew, masks, bad, act):
        store["obs"][t].copy_(obs); store["rew"][t].copy_(rew); store["masks"][t].copy_(masks); store["bad"][t].copy_(bad); store["act"][t].copy_(act)

    a.reset(); b.reset()
    graph = a.capture_rollout(policy, T, sink=sink, warmup=warm)
    obs = b.venv.obs
    for _ in range(warm):
        b.step(policy(obs))
    for r in range(3):
        graph.replay()
        torch.cuda.synchronize()
        for t in range(T):
            act = policy(obs)
            o, rw, _, _ = b.step(act)
            assert torch.equal(store["act"][t], act) and torch.equal(store["obs"][t], o) and torch.equal(store["rew"][t], rw), (r, t)
            assert torch.equal(store["masks"][t], b.masks) and torch.equal(store["bad"][t], b.bad_masks), (r, t)
    ta, tb = a.episode_totals.cpu().numpy(), b.episode_totals.cpu().numpy()
    assert tb[2] > 0 and ta[1] == tb[1] and ta[2] == tb[2] and abs(ta[0] - tb[0]) < 1e-3 * (1 + abs(tb[0]))
    a.close(); b.close()


def test_step_into_the_trainers_storage_is_the_same_step():
    """`step(action, into=...)`: the launch writes observation, reward, masks and bad_masks straight into rows of the trainer's rollout storage
    (PPO's `rollouts.insert` without copy kernels) -- the same bits a twin leaves in its own buffers; and the same from one CUDA graph
    (`capture_rollout(into=...)`), whose policy reads row t and writes its action into the storage too."""
    import torch
    from mocca_envs_amd.trainer_api import make_vec_envs
    n, T = 160, 12
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(21)
    w = torch.randn(52, 21, device=dev, generator=g) * 0.4
    a, b, c = (make_vec_envs("Walker3DCustomEnv-v0", seed=5, num_processes=n, record_events=False) for _ in range(3))
    S = {"obs": torch.zeros(T + 1, n, 52, device=dev), "reward": torch.zeros(T, n, 1, device=dev), "masks": torch.ones(T + 1, n, 1, device=dev),
         "bad_masks": torch.ones(T + 1, n, 1, device=dev), "act": torch.zeros(T, n, 21, device=dev)}
    G = {k: torch.zeros_like(v) for k, v in S.items()}
    row = lambda st: (lambda t: {"obs": st["obs"][t + 1], "reward": st["reward"][t], "masks": st["masks"][t + 1], "bad_masks": st["bad_masks"][t + 1]}
                      if t >= 0 else {"obs": st["obs"][0]})
    S["obs"][0].copy_(a.reset()); obs_b = b.reset(); c.reset()
    graph = c.capture_rollout(lambda o, t: torch.tanh(o @ w, out=G["act"][t]), T, into=row(G), warmup=0)
    for r in range(3):
        graph.replay()
        for t in range(T):
            act = torch.tanh(S["obs"][t] @ w)
            o, rw, d, infos = a.step(act, into=row(S)(t))
            assert o.data_ptr() == S["obs"][t + 1].data_ptr() and rw.data_ptr() == S["reward"][t].data_ptr()
            o2, r2, _, _ = b.step(torch.tanh(obs_b @ w))
            assert torch.equal(o, o2) and torch.equal(rw, r2) and torch.equal(S["masks"][t + 1], b.masks) and torch.equal(S["bad_masks"][t + 1], b.bad_masks), (r, t)
            assert (np.asarray(d) == (S["masks"][t + 1][:, 0].cpu().numpy() == 0.0)).all()        # the lazy records still work beside it
            obs_b = o2
        torch.cuda.synchronize()
        for k in ("obs", "reward", "masks", "bad_masks"):
            assert torch.equal(G[k][1:] if k in ("obs", "masks", "bad_masks") else G[k], S[k][1:] if k in ("obs", "masks", "bad_masks") else S[k]), (r, k)
        for st in (S, G):                                            # rollouts.after_update()
            st["obs"][0].copy_(st["obs"][T])
    for e in (a, b, c):
        e.close()
