"""Kernel time under modified model caps (diagnostic): which part of the constraint solver costs what.
usage: python tools/param_time.py [n_envs]"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import torch
from mocca_envs_amd.vec_env import VecEnv, compile_model_for
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env_id = "Walker3DCustomEnv-v0"
def run(tag, **kw):
    m = compile_model_for(env_id)
    for k, v in kw.items(): setattr(m, k, v)
    m.finalize_tables()
    env = VecEnv(env_id, n, auto_reset=True, seed=1000, model_blob=m.to_bytes())
    env.reset()
    tape = torch.rand(64, n, env.act_dim, device="cuda") * 2 - 1
    for i in range(30): env.step(tape[i % 64])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200): env.step(tape[i % 64])
    e1.record(); torch.cuda.synchronize()
    print(f"{tag:40s} {n} envs {1000 * e0.elapsed_time(e1) / 200:7.1f} us")
    env.close()
run("default")
run("n_iters=0", n_iters=0)
run("n_iters=1", n_iters=1)
run("max_contacts=0 (limit rows only)", max_contacts=0)
run("max_contacts=4", max_contacts=4)
run("max_rows=16", max_rows=16)
run("limit_slack=-1 (no limit rows)", limit_slack=-1.0)
run("max_contacts=0, no limit rows", max_contacts=0, limit_slack=-1.0)
