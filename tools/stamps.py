"""Per-phase timeline of one wave's last physics substep (diagnostic build -DMOCCA_STAMPS: raw s_memtime marks, plain
stores, no waits).  Reports the mean over waves that executed every phase.  usage: stamps.py [env-id] [n_envs] [ppo]
`ppo` (Walker3DCustomEnv-v0 only): the batch is driven by the trained policy of profiles/ppo_policy_walker3d.npz (tools/ppo_demo.py) after a
600-step closed-loop pre-roll -- the per-phase timeline of bench.py's `ppo_policy` workload instead of the headline's random torques."""
import ctypes as C, os, subprocess, sys
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
so = "/tmp/libmocca_stamps.so"
subprocess.check_call([sys.executable, "-m", "mocca_envs_amd.build", "--out", so, "-DMOCCA_STAMPS"], cwd=R, stdout=subprocess.DEVNULL)
os.environ["MOCCA_LIB_PATH"] = so
os.environ["MOCCA_ALLOW_DIAGNOSTIC_BUILD"] = "1"   # lib.load() refuses diagnostic builds otherwise
import torch
from mocca_envs_amd.vec_env import VecEnv
env_id = sys.argv[1] if len(sys.argv) > 1 else "Walker3DCustomEnv-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = VecEnv(env_id, n, auto_reset=True, seed=1000)
env.reset()
tape = torch.rand(64, n, env.act_dim, device="cuda") * 2 - 1
act = lambda rep: tape[rep % 64]
if len(sys.argv) > 3 and sys.argv[3] == "ppo":
    w = {k: torch.from_numpy(v).cuda() for k, v in np.load(os.path.join(R, "profiles", "ppo_policy_walker3d.npz")).items() if v.dtype.kind == "f"}

    def act(rep):
        o = ((env.obs - w["obs_mean"]) / torch.sqrt(w["obs_var"] + 1e-8)).clamp(-10.0, 10.0)
        h = torch.tanh(torch.tanh(o @ w["pi_0_weight"].T + w["pi_0_bias"]) @ w["pi_2_weight"].T + w["pi_2_bias"])
        return (h @ w["pi_4_weight"].T + w["pi_4_bias"] + w["log_std"].exp() * torch.randn(n, env.act_dim, device="cuda")).contiguous()
    for rep in range(600):
        env.step(act(rep))
    env_id += " under the trained PPO policy"
lib = C.CDLL(so)
SEQ = [(30, "substep start"), (29, "stage joints"), (0, "kinematics walk"), (15, "geom points"), (13, "collide: terrain"),
       (14, "collide: self pairs"), (1, "collide: epilogue"), (10, "aba: inward levels"), (11, "aba: base 6x6"),
       (12, "aba: outward walk"), (2, "aba: epilogue"), (16, "rows: limit compaction"), (17, "rows: build row"),
       (5, "rows: ancestor masks"), (18, "sweeps: anymask reduction"), (19, "sweeps: inward"), (6, "sweeps: base + outward"),
       (7, "Delassus build"), (20, "pgs: warm start"), (8, "pgs: iterations"), (9, "apply"), (3, "solve: epilogue"), (4, "integrate")]
acc = np.zeros(len(SEQ) - 1); cnt = 0; tot_all = []; slow = np.zeros(len(SEQ) - 1); nslow = 0
nw = min(n, 8192)
for rep in range(40):
    env.step(act(rep))
    torch.cuda.synchronize()
    if rep < 20: continue
    buf = np.zeros((nw, 32), np.uint64)
    lib.mocca_debug_stamps(buf.ctypes.data_as(C.c_void_p), nw)
    t = buf[:, [k for k, _ in SEQ]].astype(np.int64)
    d = np.diff(t, axis=1)
    ok = (d >= 0).all(axis=1) & (d.sum(axis=1) < 10_000_000)   # waves whose last substep ran every phase, in order
    acc += d[ok].sum(axis=0); cnt += ok.sum()
    tt = d[ok].sum(axis=1); tot_all.append(tt)
    thr = np.percentile(tt, 99)
    slow += d[ok][tt >= thr].sum(axis=0); nslow += (tt >= thr).sum()
mean = acc / cnt
print(f"{env_id}, {n} envs: mean s_memtime ticks per phase of the last substep ({cnt} wave samples); total {mean.sum():.0f}")
for (k, nm), v in zip(SEQ[1:], mean):
    print(f"  {nm:28s} {v:9.0f}  {100 * v / mean.sum():6.2f} %")
tt = np.concatenate(tot_all)
print("per-wave substep total: p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % tuple(np.percentile(tt, [10, 50, 90, 99, 100])))
print("slowest 1 % of waves, mean ticks per phase:")
for (k, nm), v in zip(SEQ[1:], slow / nslow):
    print(f"  {nm:28s} {v:9.0f}")
# whole-kernel view of the last launch: prologue -> substeps -> obs/reward -> reset + write-back
t = buf.astype(np.int64)
t0 = t[:, 28].min()
sub, ob, rs = t[:, 27] - t[:, 28], t[:, 26] - t[:, 27], t[:, 25] - t[:, 26]
was_reset = t[:, 24] != 0
end = t[:, 25] - t0
print("whole kernel, last launch: wave end time after the first wave's prologue: p50 %.0f p90 %.0f p99 %.0f max %.0f ticks" % tuple(np.percentile(end, [50, 90, 99, 100])))
print("  substeps: mean %.0f p99 %.0f max %.0f | obs+reward: mean %.0f | tail (reset + write-back): no reset %.0f, reset %.0f (%d waves reset)"
      % (sub.mean(), np.percentile(sub, 99), sub.max(), ob.mean(), rs[~was_reset].mean(), rs[was_reset].mean() if was_reset.any() else 0, was_reset.sum()))
late = np.argsort(end)[-8:]
print("  the 8 last waves: end", end[late], "substeps", sub[late], "reset", was_reset[late].astype(int))
