"""Follow-up of tools/short_window_probe.py: is it GPU-busy TIME that makes the 20-launch window fast, or launches queued without a
synchronize in between?  Reads the GPU's current shader / fabric / memory clock from sysfs inside every window (while the GPU is busy)."""
import glob, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mocca_envs_amd.vec_env import VecEnv
env = VecEnv("Walker3DCustomEnv-v0", 4096, auto_reset=True, seed=1000)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
tape = torch.rand(64, 4096, 21, device="cuda", generator=g) * 2 - 1
torch.cuda.synchronize()
for i in range(1000): env.step(tape[(i + 17) % 64])
CLK = {k: sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_%s" % k)) for k in ("sclk", "fclk", "mclk")}

def clocks():
    out = []
    for k, fs in CLK.items():
        for f in fs[:1]:
            try:
                cur = [l for l in open(f).read().splitlines() if l.rstrip().endswith("*")]
                out.append("%s %s" % (k, cur[0].split(":")[1].strip(" *") if cur else "?"))
            except Exception as e:
                out.append("%s n/a" % k)
    return ", ".join(out)

def windows(tag, k=2):
    for rep in range(k):
        n_done = torch.zeros((), device="cuda")
        for i in range(5):
            n_done += (env.step(tape[i % 64])[2] != 0).sum()
        n_done.item()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for i in range(20): env.step(tape[i % 64])
        e1.record()
        c = clocks()                      # ~1 ms into the 2 ms window
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        print("%-34s window %d: events/launch %.1f us | mid-window %s" % (tag, rep, e0.elapsed_time(e1) * 50, c), flush=True)

def busy(n, chunk=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        env.step(tape[i % 64])
        if chunk and i % chunk == chunk - 1: torch.cuda.synchronize()
    c = clocks()
    e1.record(); torch.cuda.synchronize()
    print("%d launches, %s: %.1f us per launch | %s" % (n, "synchronize every %d" % chunk if chunk else "back to back", e0.elapsed_time(e1) * 1e3 / n, c), flush=True)

print("idle clocks:", clocks())
windows("after the 1000-step preroll")
busy(20000, 256); windows("after 2 s in chunks of 256")
time.sleep(3.0); windows("after 3 s of sleep")
busy(20000); windows("after 2 s back to back")
time.sleep(3.0); windows("after 3 s of sleep")
busy(20000, 256); windows("after 2 s in chunks of 256")
busy(20000, 2048); windows("after 2 s in chunks of 2048")
