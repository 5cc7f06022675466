"""The C ABI used by a caller that is not Python: tests/abi_client/abi_client.c (plain C, include/mocca.h, HIP's C runtime API for the
device buffers) is built with gcc, run on the GPU, and must print exactly what the same episodes give through the Python binding --
rewards, done flags and observation rows bit for bit.  PyTorch is one user of the boundary, not part of it.  Needs a real MI355X."""
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fnv1a(b: bytes) -> int:
    h = 2166136261
    for x in b:
        h = ((h ^ x) * 16777619) & 0xFFFFFFFF
    return h


@pytest.mark.parametrize("env_id,task", [("Walker3DCustomEnv-v0", 0), ("Walker3DStepperEnv-v0", 1)])
def test_a_plain_c_caller_gets_the_same_bits(tmp_path, env_id, task):
    import torch
    from mocca_envs_amd.vec_env import VecEnv, compile_model_for
    exe = str(tmp_path / "abi_client")
    libdir, rocm = os.path.join(ROOT, "mocca_envs_amd"), os.environ.get("ROCM_PATH", "/opt/rocm")
    # a C compiler, not hipcc: the client has no device code, only the HIP runtime's C API for its buffers
    subprocess.check_call([shutil.which("gcc") or "gcc", "-std=c99", "-O2", os.path.join(ROOT, "tests", "abi_client", "abi_client.c"),
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(rocm, "include"), "-D__HIP_PLATFORM_AMD__", "-o", exe,
                           "-L" + libdir, "-lmocca_hip", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(rocm, "lib")])
    n, steps, seed = 96, 40, 77
    blob = compile_model_for(env_id).to_bytes()
    (tmp_path / "model.blob").write_bytes(blob)
    out = subprocess.run([exe, str(tmp_path / "model.blob"), str(task), str(n), str(steps), str(seed)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    rows, stats = [l.split() for l in lines[:-1]], lines[-1].split()
    assert len(rows) == n and stats[0] == "episodes" and stats[5] == "totals"
    # the same episodes through the Python binding
    env = VecEnv(env_id, n, auto_reset=True, seed=seed)
    env.reset()
    ep_count = ep_len = 0
    length = np.zeros(n, int)
    lcg = np.uint32(12345)
    for t in range(steps):
        a = np.empty(n * env.act_dim, np.float32)
        with np.errstate(over="ignore"):
            for i in range(a.size):
                lcg = np.uint32(lcg * np.uint32(1664525) + np.uint32(1013904223))
                a[i] = np.float32(int(lcg) >> 8) * np.float32(2.0 / 16777216.0) - np.float32(1.0)
        obs, rew, done, _ = env.step(torch.from_numpy(a.reshape(n, env.act_dim)).cuda())
        length += 1
        fin = done.cpu().numpy() != 0
        ep_count += int(fin.sum()); ep_len += int(length[fin].sum()); length[fin] = 0
    # Monitor's statistics as the C client read them out of the pinned record ring, and the device totals: the same episodes, counted here by hand
    assert [int(x) for x in stats[1:4]] == [ep_count, ep_len, 0] and [int(x) for x in stats[6:9]] == [ep_count, ep_len, 0] and ep_count > 0
    obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
    n_done = 0
    for e, (idx, rbits, d, h) in enumerate(rows):
        assert int(idx) == e
        assert int(rbits, 16) == int(rew[e:e + 1].view(np.uint32)[0]), f"env {e} reward"
        assert int(d) == int(done[e]), f"env {e} done"
        assert int(h, 16) == _fnv1a(obs[e].tobytes()), f"env {e} observation"
        n_done += int(d) != 0
    env.close()
