import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mocca_envs_amd.vec_env import VecEnv
import ctypes as C
for n in (4096,):
    env = VecEnv("Walker3DCustomEnv-v0", n, auto_reset=True, seed=1)
    env.reset()
    a = torch.rand(n, env.act_dim, device="cuda") * 2 - 1
    for i in range(300): env.step(a)
    torch.cuda.synchronize()
    # host enqueue cost: 12 calls into an empty queue (the queue never fills: 12 x ~100 us of GPU work)
    for rep in range(3):
        torch.cuda.synchronize()
        t = []
        t0 = time.perf_counter()
        for i in range(12):
            env.step(a); t.append(time.perf_counter())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        d = [1e6 * (t[0] - t0)] + [1e6 * (t[i] - t[i - 1]) for i in range(1, 12)]
        print(f"n={n} host us per env.step call: " + " ".join(f"{x:.1f}" for x in d) + f" | total window {1e6*(t1-t0):.0f} us for 12 steps")
    # pieces
    N = 2000
    t0 = time.perf_counter()
    for i in range(N): env._stream()
    t1 = time.perf_counter()
    for i in range(N): (C.c_void_p(a.data_ptr()), C.c_void_p(env.obs.data_ptr()), C.c_void_p(env.rew.data_ptr()), C.c_void_p(env.done.data_ptr()), C.c_void_p(env.info.data_ptr()))
    t2 = time.perf_counter()
    for i in range(N): a.device != env.device or a.dtype != torch.float32 or not a.is_contiguous(); a.shape != (env.n_envs, env.act_dim)
    t3 = time.perf_counter()
    print(f"_stream() {1e6*(t1-t0)/N:.2f} us, five pointers {1e6*(t2-t1)/N:.2f} us, argument checks {1e6*(t3-t2)/N:.2f} us")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); ev0.record(); ta = time.perf_counter()
        for i in range(20): env.step(a)
        tb = time.perf_counter(); ev1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"window of 20: wall {1e6*(t1-t0):.0f} us, events {1e3*ev0.elapsed_time(ev1):.0f} us, ev0.record {1e6*(ta-t0):.1f} us, 20 enqueues {1e6*(tb-ta):.0f} us, record+sync after last enqueue {1e6*(t1-tb):.0f} us")
