"""The two instances of the step kernel are the same program: mocca_r32.hip (32 rows / 10 contacts per env, compact LDS layout, picked by
mocca_create from the blob's caps) against the 48-row instance forced onto the SAME blob (MOCCA_PARAM_KERNEL_VARIANT = 1) -- observations,
rewards, done flags, state, task and terrain records must agree BIT FOR BIT over free-running rollouts with in-kernel auto-resets, for
every topology the compact instance is built for; plus the traffic switch of the slots' normal impulses (MOCCA_PARAM_PERSIST_IMPULSES).
Needs a real MI355X: -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

IDS = ["Walker3DCustomEnv-v0", "Walker3DStepperEnv-v0", "Walker3DPlannerEnv-v0", "MikePlannerEnv-v0", "Child3DCustomEnv-v0", "MikeStepperEnv-v0",
       "Walker2DCustomEnv-v0", "Crab2DCustomEnv-v0", "LaikagoCustomEnv-v0", "LaikagoStepperEnv-v0"]


@pytest.mark.parametrize("env_id", IDS)
def test_compact_and_full_instances_agree_bit_for_bit(env_id):
    import torch
    from mocca_envs_amd import lib as L
    from mocca_envs_amd.vec_env import VecEnv
    n, steps = 512, 300
    a_env = VecEnv(env_id, n, auto_reset=True, seed=21, max_rows=32)
    b_env = VecEnv(env_id, n, auto_reset=True, seed=21, max_rows=32)
    b_env.set_param(L.PARAM_KERNEL_VARIANT, 1)
    ka, kb = a_env.kernel_info(), b_env.kernel_info()
    assert ka["lds_bytes"] <= 8192 and kb["lds_bytes"] > 8192, (ka, kb)          # two different kernels ...
    assert ka["max_blocks_per_cu"] >= 20 and ka["scratch_bytes"] <= 128, ka      # ... the compact one at >= 5 waves per SIMD
    if "Stepper" in env_id:
        a_env.set_param(L.PARAM_CURRICULUM, 9); b_env.set_param(L.PARAM_CURRICULUM, 9)
    dbg_a, dbg_b = a_env.set_debug(True), b_env.set_debug(True)
    oa, ob = a_env.reset(), b_env.reset()
    assert torch.equal(oa, ob)
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    n_done = 0
    for t in range(steps):
        act = (torch.rand(n, a_env.act_dim, device="cuda", generator=g) * 2 - 1) * (1.0 if t % 3 else 0.3)
        ra, rb = a_env.step(act), b_env.step(act)
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), (env_id, t)
        n_done += int((ra[2] != 0).sum())
    assert torch.equal(a_env.get_state()[:, :13 + 2 * a_env.model.n_joints], b_env.get_state()[:, :13 + 2 * a_env.model.n_joints])
    assert torch.equal(a_env.get_task(), b_env.get_task())
    assert torch.equal(dbg_a, dbg_b)                      # same active sets, same clamp masks and signatures (48-row lane numbering), same cap-pressure counters
    if "Stepper" in env_id:
        assert torch.equal(a_env.get_terrain(), b_env.get_terrain())
    if "2D" not in env_id:        # (Walker2DCustomEnv never sets done, env_locomotion.py:302-309: only the TimeLimit ends its episodes)
        assert n_done > n // 4, "the rollout must cross in-kernel resets"
    assert int(dbg_a[:, 15].max()) >= 12, "the sample must contain contact-rich substeps"
    a_env.close(); b_env.close()


@pytest.mark.parametrize("env_id", ["Walker3DCustomEnv-v0", "Walker3DStepperEnv-v0", "CassieEnv-v0", "LaikagoStepperEnv-v0", "Crab2DCustomEnv-v0", "MikePlannerEnv-v0"])
def test_wide_and_full_instances_agree_bit_for_bit(env_id):
    """mocca_r64.hip (64 rows / 20 contacts, the accuracy instance) forced onto a blob with the product's caps (MOCCA_PARAM_KERNEL_VARIANT = 2)
    against the 48-row instance on the same blob: the same program with another lane layout (friction rows on lanes 62 - 2i / 63 - 2i) and
    another LDS layout -- everything must agree bit for bit, debug records included (recorded in the 48-row lane numbering for such a blob)."""
    import torch
    from mocca_envs_amd import lib as L
    from mocca_envs_amd.vec_env import VecEnv
    n, steps = 512, (40 if "Cassie" in env_id else 300)
    a_env = VecEnv(env_id, n, auto_reset=True, seed=21)
    b_env = VecEnv(env_id, n, auto_reset=True, seed=21)
    b_env.set_param(L.PARAM_KERNEL_VARIANT, 2)
    ka, kb = a_env.kernel_info(), b_env.kernel_info()
    assert ka["lds_bytes"] <= 10240 and kb["lds_bytes"] > 16384, (ka, kb)
    if "Stepper" in env_id:
        a_env.set_param(L.PARAM_CURRICULUM, 9); b_env.set_param(L.PARAM_CURRICULUM, 9)
    dbg_a, dbg_b = a_env.set_debug(True), b_env.set_debug(True)
    assert torch.equal(a_env.reset(), b_env.reset())
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    for t in range(steps):
        act = (torch.rand(n, a_env.act_dim, device="cuda", generator=g) * 2 - 1) * (1.0 if t % 3 else 0.3)
        for x, y in zip(a_env.step(act), b_env.step(act)):
            assert torch.equal(x, y), (env_id, t)
    nd = 13 + 2 * a_env.model.n_joints
    assert torch.equal(a_env.get_state()[:, :nd], b_env.get_state()[:, :nd]) and torch.equal(a_env.get_task(), b_env.get_task())
    assert torch.equal(dbg_a, dbg_b)
    assert int(dbg_a[:, 15].max()) >= (6 if "Cassie" in env_id else 12)
    with pytest.raises(L.MoccaError):
        VecEnv(env_id, 4, max_rows=64).set_param(L.PARAM_KERNEL_VARIANT, 1)     # caps beyond the 48-row instance cannot be forced onto it
    a_env.close(); b_env.close()


def test_the_wide_instance_holds_what_the_product_caps_drop():
    """Stepper curriculum 9 (BASELINE config 3's hard end): with 64 rows / 20 contacts the cap-pressure counters of the debug record stay at
    zero where the 48 / 12 caps drop contacts or rows (profiles/archive/r04_cap_pressure.jsonl: 11 % of the envs at least once in 1000 steps), and
    row counts above 48 are really solved."""
    import torch
    from mocca_envs_amd import lib as L
    from mocca_envs_amd.vec_env import VecEnv
    n, steps = 2048, 400
    res = {}
    for name, kw in (("capped", {}), ("wide", {"max_rows": 64})):
        env = VecEnv("Walker3DStepperEnv-v0", n, auto_reset=True, seed=3, **kw)
        assert (env.model.max_rows, env.model.max_contacts) == ((64, 20) if kw else (48, 12))
        env.set_param(L.PARAM_CURRICULUM, 9)
        dbg = env.set_debug(True)
        env.reset()
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        rows_max = 0
        for t in range(steps):
            env.step(torch.rand(n, 21, device="cuda", generator=g) * 2 - 1)
            if t % 20 == 0:
                rows_max = max(rows_max, int(dbg[:, 0].max()))
        d = dbg.cpu().numpy()
        res[name] = dict(contact_drops=int(d[:, 12].sum()), row_drops=int(d[:, 13].sum()), envs_capped=int(((d[:, 12] + d[:, 13]) > 0).sum()),
                         wanted_max=int(d[:, 15].max()), rows_max=rows_max)
        env.close()
    print(res)
    assert res["capped"]["envs_capped"] > 0 and res["capped"]["wanted_max"] > 48
    assert res["wide"]["wanted_max"] > 48 and res["wide"]["envs_capped"] <= res["capped"]["envs_capped"] // 20


def test_caps_above_the_compact_instance_run_the_full_one():
    from mocca_envs_amd.vec_env import VecEnv
    for kw, compact in (({}, False), ({"max_rows": 33}, False), ({"max_rows": 32, "max_contacts": 11}, False), ({"max_rows": 32}, True), ({"max_rows": 16}, True)):
        env = VecEnv("Walker3DCustomEnv-v0", 4, **kw)
        assert (env.kernel_info()["lds_bytes"] <= 8192) == compact, kw
        env.close()
    env = VecEnv("CassieEnv-v0", 4)                       # loop closures: 48-row instance whatever the caps
    assert env.kernel_info()["lds_bytes"] > 8192
    env.close()


@pytest.mark.parametrize("max_rows", [None, 32])
def test_impulses_are_persisted_on_request_only(max_rows):
    """A blob that does not warm-start neither loads nor stores the slots' normal impulses (state words 13 + 2 NJ ..): what set_state put
    there survives a step untouched; with MOCCA_PARAM_PERSIST_IMPULSES they are the last substep's impulses (standing robots: positive on the
    feet), and the dynamic state does not depend on the switch."""
    import torch
    from mocca_envs_amd import lib as L
    from mocca_envs_amd.vec_env import VecEnv
    n = 64
    envs = [VecEnv("Walker3DCustomEnv-v0", n, auto_reset=False, seed=5, max_rows=max_rows) for _ in range(2)]
    assert envs[0].model.warmstart == 0.0
    envs[1].set_param(L.PARAM_PERSIST_IMPULSES, 1)
    nd = 13 + 2 * envs[0].model.n_joints
    for e in envs:
        e.reset()
        st = e.get_state()
        st[:, nd:] = 7.0
        e.set_state(st)
    act = torch.zeros(n, 21, device="cuda")
    for _ in range(30):
        for e in envs:
            e.step(act)
    s0, s1 = envs[0].get_state(), envs[1].get_state()
    assert torch.equal(s0[:, :nd], s1[:, :nd])
    assert (s0[:, nd:] == 7.0).all()
    assert (s1[:, nd:] != 7.0).all() and (s1[:, nd:] >= 0).all() and (s1[:, nd:].sum(dim=1) > 0).float().mean() > 0.5
    for e in envs:
        e.close()


@pytest.mark.parametrize("env_id,kw", [("Walker3DCustomEnv-v0", {}), ("Walker3DStepperEnv-v0", {"max_rows": 32}), ("CassieEnv-v0", {})])
def test_launch_order_does_not_change_results(env_id, kw):
    """MOCCA_PARAM_ORDER_EVERY: the step kernel starts the heaviest envs first (a permutation rebuilt every K steps from the row counts of
    the step before); every env's results are bit-identical to index order, and the permutation really is one."""
    import torch
    from mocca_envs_amd import lib as L
    from mocca_envs_amd.vec_env import VecEnv
    n, steps = 777, 60 if env_id.startswith("Cassie") else 200     # odd batch size: the last workgroups must not run off the permutation
    a_env = VecEnv(env_id, n, auto_reset=True, seed=8, **kw)
    b_env = VecEnv(env_id, n, auto_reset=True, seed=8, **kw)
    b_env.set_param(L.PARAM_ORDER_EVERY, 3)
    b_env.set_param(L.PARAM_PACE_TICKS, 180000)                    # ... and with pace priorities instead of row-count priorities (timing only)
    a_env.reset(); b_env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(4)
    for t in range(steps):
        act = (torch.rand(n, a_env.act_dim, device="cuda", generator=g) * 2 - 1) * (0.1 if env_id.startswith("Cassie") else 1.0)
        for x, y in zip(a_env.step(act), b_env.step(act)):
            assert torch.equal(x, y), (env_id, t)
    nd = 13 + 2 * a_env.model.n_joints
    assert torch.equal(a_env.get_state()[:, :nd], b_env.get_state()[:, :nd]) and torch.equal(a_env.get_task(), b_env.get_task())
    b_env.set_param(L.PARAM_ORDER_EVERY, 0)                        # back to index order
    act = torch.zeros(n, a_env.act_dim, device="cuda")
    for x, y in zip(a_env.step(act), b_env.step(act)):
        assert torch.equal(x, y)
    a_env.close(); b_env.close()
