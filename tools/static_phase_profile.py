"""Static instruction count of the step kernel by phase: reads device assembly made with -gline-tables-only and attributes every
instruction to the outermost function of its inline chain below substep() / the kernel body (depth 2: that function > its callee).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Iinclude -Imocca_envs_amd/csrc -S --cuda-device-only -gline-tables-only \
        mocca_envs_amd/csrc/mocca_api.hip -o /tmp/api.s
  awk '/^_ZN5mocca17mocca_step_kernelI12TopoWalker3DLi0ELb0EEEvNS_8StepArgsE:/{f=1} f{print} f&&/^\.Lfunc_end/{exit}' /tmp/api.s > /tmp/k.s
  python tools/static_phase_profile.py /tmp/k.s [depth]

Static, not dynamic: the substep body appears once (the kernel loops over it), loops that are not unrolled count once, and code behind a
wave-uniform branch counts although most waves skip it (reset_env; the libm fallback inside integrate).  Run from the repo root."""
import re, sys, bisect
from collections import defaultdict
asm = sys.argv[1]
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 1
srcs = {"mocca_device.h": "mocca_envs_amd/csrc/mocca_device.h", "mocca_kernels.h": "mocca_envs_amd/csrc/mocca_kernels.h"}
funcs = {}
for k, p in srcs.items():
    starts = []
    for i, ln in enumerate(open(p), 1):
        m = re.match(r"^DI\s+[\w:<>\*&\s]+?\b(\w+)\s*\(", ln)
        if m: starts.append((i, m.group(1)))
        m2 = re.match(r"^__global__.*\b(\w+)\s*\(", ln)
        if m2: starts.append((i, m2.group(1)))
    funcs[k] = (starts, [s[0] for s in starts])
def fn(f, L):
    f = f.split("/")[-1]
    if f in funcs:
        st, ls = funcs[f]; i = bisect.bisect_right(ls, L) - 1
        return st[i][1] if i >= 0 else f
    return f
SKIP = {"mocca_step_kernel", "substep", "env_step_impl"}
cur = None
cnt = defaultdict(lambda: [0, 0, 0, 0])
for ln in open(asm):
    if re.match(r"\s*\.loc\s", ln):
        fr = re.findall(r"([^\s:\[\]@;]+):(\d+):\d+", ln.split(";", 1)[1]) if ";" in ln else []
        fr = [(f, int(L)) for f, L in fr if int(L)]
        if fr:
            names = [fn(f, L) for f, L in fr][::-1]   # outermost first
            names = [n for n in names if n not in SKIP] or ["kernel body"]
            # collapse consecutive duplicates
            nn = [names[0]]
            for n in names[1:]:
                if n != nn[-1]: nn.append(n)
            cur = " > ".join(nn[:depth])
        continue
    t = ln.strip()
    if not t or t.startswith((";", ".")) or t.endswith(":"): continue
    op = t.split()[0]
    k = 0 if op.startswith("v_") else 1 if op.startswith("ds_") else 2 if op.startswith("s_") else 3
    cnt[cur or "?"][k] += 1
tot = [sum(v[i] for v in cnt.values()) for i in range(4)]
print("%-60s %6s %6s %6s %6s" % ("phase", "VALU", "LDS", "SALU", "other"))
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1][0]):
    print("%-60s %6d %6d %6d %6d" % (k, *v))
print("%-60s %6d %6d %6d %6d" % ("total", *tot))
