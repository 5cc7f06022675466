"""Per-phase share of a wave's timeline (diagnostic build -DMOCCA_STAMPS; never quote its run time, only shares)."""
import ctypes as C, os, subprocess, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
so = "/tmp/libmocca_stamps.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-shared", "-fPIC", "-DMOCCA_STAMPS",
                       "-I" + R + "/include", "-I" + R + "/mocca_envs_amd/csrc", "-o", so, R + "/mocca_envs_amd/csrc/mocca_api.hip"])
os.environ["MOCCA_LIB_PATH"] = so
import torch
from mocca_envs_amd.vec_env import VecEnv
env_id = sys.argv[1] if len(sys.argv) > 1 else "Walker3DCustomEnv-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = VecEnv(env_id, n, auto_reset=True, seed=1000)
env.reset()
tape = torch.rand(64, n, env.act_dim, device="cuda") * 2 - 1
for i in range(20): env.step(tape[i % 64])
torch.cuda.synchronize()
lib = C.CDLL(so); buf = (C.c_ulonglong * 32)()
lib.mocca_debug_stamps(buf)
for i in range(50): env.step(tape[i % 64])
torch.cuda.synchronize()
lib.mocca_debug_stamps(buf)
v = list(buf)
names = ["kinematics walk", "geom points + collide", "ABA passes", "constraint solve (all)", "integrate",
         "  solve: row setup", "  solve: response sweeps", "  solve: Delassus build", "  solve: warm start + PGS", "  solve: apply", "  aba: inward levels", "  aba: base 6x6", "  aba: outward walk",
         "  collide: terrain", "  collide: self pairs"]
tot = sum(v[:5])
for k, nm in enumerate(names):
    print(f"{nm:32s} {100.0 * v[k] / tot:6.2f} %   {v[k] / (50 * n * (50 if 'Cassie' in env_id else 4)):9.0f} ticks/substep/wave")
