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
if len(sys.argv) > 1 and sys.argv[1] == "--packing":      # mode 2 (bottom of the file): no assembly to read
    sys.argv.insert(1, "/dev/null")
asm = sys.argv[1]
depth = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1
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
if asm != "/dev/null":
    print("%-60s %6s %6s %6s %6s" % ("phase", "VALU", "LDS", "SALU", "other"))
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1][0]):
        print("%-60s %6d %6d %6d %6d" % (k, *v))
    print("%-60s %6d %6d %6d %6d" % ("total", *tot))


# ---- mode 2: `python tools/static_phase_profile.py --packing profiles/<tag>_stamps_custom4096.txt`
# Decision aid for config 5's kernel: per phase of a substep, the MEASURED share of a wave's cycles (tools/stamps.py)
# x the lanes the phase keeps busy, and what two candidate kernels would make of it:
#   A  "residency": today's mapping (one wave per env) shrunk to 5 KB of LDS / 64 VGPRs so that 8 waves per SIMD are resident
#   B  "half-wave": two envs per wave, env e on lanes 32 e .. 32 e + 31 of ONE instruction stream -- phases that use <= 32 lanes serve both envs
#      at the cost of one, phases that use more run twice (or are re-tiled: the kinematics walk's 63 lanes = 3 per body)
LANES = {  # lanes busy in the phase, and whether the phase's instruction count grows with them
    "stage joints": 21, "kinematics walk": 63, "geom points": 44, "collide: terrain": 34, "collide: self pairs": 64, "collide: epilogue": 1,
    "aba: inward levels": 32, "aba: base 6x6": 64, "aba: outward walk": 22, "aba: epilogue": 1, "rows: limit compaction": 42, "rows: build row": 32,
    "rows: ancestor masks": 32, "sweeps: anymask reduction": 32, "sweeps: inward": 32, "sweeps: base + outward": 32, "Delassus build": 32,
    "pgs: warm start": 32, "pgs: iterations": 32, "apply": 27, "solve: epilogue": 1, "integrate": 27}
# (rows / sweeps / Delassus / PGS: lane = row; the compact blob caps rows at 32, the flat-ground walker holds 6 per substep)


def packing(stamps_path):
    phases = []
    for ln in open(stamps_path):
        m = re.match(r"^\s{2}(\S.*?)\s{2,}(\d+)\s+([\d.]+) %", ln)
        if m and m.group(1) in LANES:
            phases.append((m.group(1), int(m.group(2)), LANES[m.group(1)]))
        if ln.startswith("per-wave substep total"):
            break
    tot = sum(t for _, t, _ in phases)
    print("%-28s %8s %7s %6s %18s" % ("phase", "ticks", "share", "lanes", "two envs per wave"))
    packed = 0.0
    for name, t, lanes in phases:
        # <= 32 lanes: both envs in one pass.  The redundant base solve (every lane solves the same 6x6) becomes two solves on half the lanes: 1 pass.
        passes = 1 if (lanes <= 32 or name == "aba: base 6x6") else 2
        if name == "collide: self pairs":
            passes = 5 / 3.0          # 141 pairs = 3 batches of 64 per env, 282 pairs = 5 batches for two
        packed += t * passes
        print("%-28s %8d %6.1f%% %6d %18s" % (name, t, 100.0 * t / tot, lanes, "1 pass" if passes == 1 else "%.2f passes" % passes))
    print("ticks of a substep per WAVE: today %d (one env), two envs per wave %d -> per env %.2f x today's" % (tot, packed, packed / 2 / tot))
    return packed / 2 / tot


if len(sys.argv) > 3 and sys.argv[2] == "--packing":
    packing(sys.argv[3])
