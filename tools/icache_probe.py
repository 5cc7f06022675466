"""Is the step kernel sensitive to its code size (90 KB against a 64 KB instruction cache)?  Same dynamic instruction stream
(Walker3D WITHOUT self-collision pairs: the two-path row sweep is never taken), two builds: the product, and one with the
two-path sweep compiled out (-DMOCCA_NO_TWO_PATHS, ~12 KB less code inside the substep loop).  usage: python tools/icache_probe.py"""
import os, subprocess, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
libs = {"product": None, "no_two_paths": "/tmp/libmocca_n2p.so"}
subprocess.check_call([sys.executable, "-m", "mocca_envs_amd.build", "--out", libs["no_two_paths"], "-DMOCCA_NO_TWO_PATHS"], cwd=R, stdout=subprocess.DEVNULL)
CODE = r'''
import sys, torch
sys.path.insert(0, %r)
from mocca_envs_amd import model as M
from mocca_envs_amd.vec_env import VecEnv
m = M.compile_walker3d(self_collision=False)
for n in (1024, 4096):
    env = VecEnv("Walker3DCustomEnv-v0", n, auto_reset=True, seed=1000, model_blob=m.to_bytes())
    env.reset()
    tape = torch.rand(64, n, 21, device="cuda") * 2 - 1
    for i in range(200): env.step(tape[i %% 64])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(400): env.step(tape[i %% 64])
    e1.record(); torch.cuda.synchronize()
    print(sys.argv[1], n, "envs", round(1000 * e0.elapsed_time(e1) / 400, 1), "us")
    env.close()
''' % R
for rnd in range(2):
    for tag, so in libs.items():
        env = dict(os.environ, MOCCA_ALLOW_DIAGNOSTIC_BUILD="1")
        if so: env["MOCCA_LIB_PATH"] = so
        subprocess.check_call([sys.executable, "-c", CODE, tag], env=env)
