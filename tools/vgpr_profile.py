"""Where does a kernel's VGPR budget go?  Reads device assembly made with -gline-tables-only (hipcc -S --cuda-device-only) and prints, per
source line range of mocca_device.h / mocca_kernels.h, the highest VGPR index any instruction attributed to it touches -- the register
allocator hands out low indices first, so high indices mark the phases that set the kernel's register count (and its waves per SIMD).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Iinclude -Imocca_envs_amd/csrc -S --cuda-device-only -gline-tables-only \
        mocca_envs_amd/csrc/mocca_r32.hip -o /tmp/r32.s
  python tools/vgpr_profile.py /tmp/r32.s _ZN9mocca_r3217mocca_step_kernelI12TopoWalker3DLi0ELb0EEEvNS_8StepArgsE [threshold]
"""
import re
import sys
from collections import defaultdict


def main():
    path, sym = sys.argv[1], sys.argv[2]
    thr = int(sys.argv[3]) if len(sys.argv) > 3 else 96
    files = {}
    inside = False
    cur = (None, 0)
    hi = defaultdict(int)      # (file, line) -> max vgpr index
    cnt = defaultdict(int)     # (file, line) -> instructions
    over = defaultdict(int)    # (file, line) -> instructions touching a register >= thr
    n_inst = 0
    rx1, rx2 = re.compile(r"\bv(\d+)\b"), re.compile(r"\bv\[(\d+):(\d+)\]")
    for ln in open(path):
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
            continue
        if ln.startswith(sym + ":"):
            inside = True
            continue
        if not inside:
            continue
        if ln.startswith(".Lfunc_end") or ln.strip().startswith("s_endpgm") and False:
            break
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        t = ln.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        regs = [int(x) for x in rx1.findall(t)] + [int(b) for _, b in rx2.findall(t)]
        n_inst += 1
        cnt[cur] += 1
        if regs:
            mx = max(regs)
            hi[cur] = max(hi[cur], mx)
            if mx >= thr:
                over[cur] += 1
    print(f"{n_inst} instructions; lines whose instructions touch v{thr}+ (file:line  max-index  touching/total):")
    for k in sorted(over, key=lambda k: (k[0], k[1])):
        print(f"  {k[0]}:{k[1]:5d}  v{hi[k]:3d}  {over[k]:4d}/{cnt[k]}")


if __name__ == "__main__":
    main()
