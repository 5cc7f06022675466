#!/usr/bin/env python3
"""ISA-level chain lengths of the step kernel's phases ("dump the ISA of walk_phase2 and the ABA inward pass, mark the
longest dependent chain in each").  Reads device assembly made with -gline-tables-only (same recipe as tools/static_phase_profile.py), takes
the instructions whose inline chain passes through a source-line range of mocca_device.h, in program order, and reports for each range
  * the instruction mix (VALU / transcendental / DPP / LDS / SALU / s_nop / s_waitcnt),
  * ISSUE time of ONE wave: every instruction at the single-wave issue cost of /opt/skills/guides/MI355X_MICROARCH.md (vector 4 cycles,
    transcendental 8, s_nop n: 4 (n + 1) / 4 ... counted as n + 1 cycles, scalar 4),
  * the LONGEST DEPENDENT CHAIN through registers (VGPR / SGPR / VCC / SCC / EXEC def-use inside the range): its length in instructions
    and in cycles under a latency model -- dependent VALU 6.6 cycles (guide: 4 x 1.66), transcendental 10, DPP source +8 (two wait states
    + the row hop), LDS read 64 cycles from issue to use, LDS write 4, v_readlane / v_readfirstlane -> scalar use 20, scalar 4.
A range whose issue time exceeds its longest chain is ISSUE-bound inside one wave: no amount of interleaving of independent chains inside that
wave can shorten it; only fewer instructions can.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Iinclude -Imocca_envs_amd/csrc -S --cuda-device-only -gline-tables-only \\
        mocca_envs_amd/csrc/mocca_api.hip -o /tmp/api.s
  awk '/^_ZN5mocca17mocca_step_kernelI12TopoWalker3DLi0ELb0EEEvNS_8StepArgsE:/{f=1} f{print} f&&/^\\.Lfunc_end/{exit}' /tmp/api.s > /tmp/k.s
  python tools/isa_chain.py /tmp/k.s          (run from the repo root; result of round 6: profiles/r06_isa_chain.txt)
Static: loops that are not unrolled count once; the substep body appears once.  The walk is instantiated twice in the kernel (FULL inside
the substep, reduced for the observation): the FIRST contiguous occurrence of a range is taken."""
import re
import sys

SRC = "mocca_envs_amd/csrc/mocca_device.h"


def func_lines(name, nxt=None):
    """[first, last] source line of function `name` in mocca_device.h (up to the line before the next top-level DI / template)."""
    L = open(SRC).read().split("\n")
    start = next(i for i, l in enumerate(L, 1) if re.match(r"^DI\s.*\b%s\s*\(" % name, l))
    end = next((i for i in range(start + 1, len(L) + 1) if re.match(r"^(DI\s|template\s*<|// -{10})", L[i - 1])), len(L)) - 1
    return start, end


def stamp_line(n):
    return next(i for i, l in enumerate(open(SRC), 1) if re.search(r"STAMP\(%d\);" % n, l))


REG = re.compile(r"\b([vs])(\d+)\b|\b([vs])\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add(m.group(1) + m.group(2))
        else:
            out.update(m.group(3) + str(k) for k in range(int(m.group(4)), int(m.group(5)) + 1))
    for special in ("vcc", "scc", "exec"):
        if re.search(r"\b%s\b" % special, tok):
            out.add(special)
    return out


TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")


def analyse(insts):
    """insts: list of (op, operand string).  Returns mix, issue cycles, chain (instructions, cycles)."""
    mix = dict(valu=0, trans=0, dpp=0, lds_rd=0, lds_wr=0, salu=0, smem=0, vmem=0, nop_cycles=0, waitcnt=0)
    issue = 0
    ready = {}          # register -> (cycle its value is available, chain length in instructions)
    best = (0.0, 0)
    for op, args in insts:
        parts = [a.strip() for a in args.split(",")] if args else []
        is_store = op.startswith(("ds_write", "ds_store", "global_store", "buffer_store", "flat_store", "s_store"))
        cmp_like = op.startswith(("v_cmp", "v_cmpx", "s_cmp", "s_bitcmp"))
        if op.startswith("s_nop"):
            n = int(parts[0], 0) + 1 if parts else 1
            mix["nop_cycles"] += n; issue += n
            continue
        if op.startswith("s_waitcnt"):
            mix["waitcnt"] += 1
            continue
        if op.startswith(("s_cbranch", "s_branch", "s_barrier", "s_setprio", "s_sleep", "s_endpgm", ";")):
            issue += 4
            continue
        dst = set() if (is_store or not parts) else regs(parts[0])
        src = regs(" ".join(parts if is_store else parts[1:]))
        if cmp_like:
            dst = regs(parts[0]) if op.endswith("_e64") else ({"vcc"} if op.startswith("v_") else {"scc"})
            src = regs(" ".join(parts if not op.endswith("_e64") else parts[1:]))
        if op.startswith(("v_cndmask", "v_addc", "v_subb", "v_div_fmas")) and "vcc" in args:
            src.add("vcc")
        if op.startswith(("s_cselect", "s_addc", "s_subb")):
            src.add("scc")
        if op.startswith(("s_add", "s_sub", "s_and", "s_or", "s_xor", "s_lshl", "s_lshr", "s_ashr", "s_bfe", "s_mul", "s_min", "s_max", "s_andn2", "s_not")):
            dst.add("scc")
        if "dpp" in op or "quad_perm" in args or "row_" in args:
            mix["dpp"] += 1
        if op.startswith(TRANS):
            mix["trans"] += 1; cost, lat = 8, 10.0
        elif op.startswith(("ds_read", "ds_load", "ds_bpermute", "ds_permute", "ds_swizzle")):
            mix["lds_rd"] += 1; cost, lat = 4, 64.0
        elif op.startswith("ds_"):
            mix["lds_wr"] += 1; cost, lat = 4, 4.0
        elif op.startswith(("v_readlane", "v_readfirstlane")):
            mix["valu"] += 1; cost, lat = 4, 20.0
        elif op.startswith("v_"):
            mix["valu"] += 1; cost, lat = 4, 6.6
            if "dpp" in op or "quad_perm" in args or "row_" in args:
                lat += 8.0
        elif op.startswith(("s_load", "s_buffer_load")):
            mix["smem"] += 1; cost, lat = 4, 200.0
        elif op.startswith(("global_", "buffer_", "flat_")):
            mix["vmem"] += 1; cost, lat = 4, 500.0
        else:
            mix["salu"] += 1; cost, lat = 4, 4.0
        issue += cost
        t0, n0 = 0.0, 0
        for r in src:
            if r in ready and ready[r][0] > t0:
                t0, n0 = ready[r]
        t1, n1 = t0 + lat, n0 + 1
        for r in dst:
            ready[r] = (t1, n1)
        if t1 > best[0]:
            best = (t1, n1)
    return mix, issue, best


def main(asm):
    a0, a1 = func_lines("aba_passes")
    ranges = [("walk_phase1 (body frames: 63 lanes, 8 path steps)",) + func_lines("walk_phase1"),
              ("walk_phase2 (S, velocity, c, link inertia, bias force)",) + func_lines("walk_phase2"),
              ("aba_passes: inward levels (8 levels x <= 4 bodies x 8 lanes)", a0, stamp_line(10) - 1),
              ("aba_passes: base 6x6 (gather, Cholesky, solve)", stamp_line(10), stamp_line(11) - 1),
              ("aba_passes: outward walk", stamp_line(11), a1)]
    cur, seqs, done = [], {r[0]: [] for r in ranges}, set()
    last_hit = {r[0]: None for r in ranges}
    idx = 0
    for ln in open(asm):
        if re.match(r"\s*\.loc\s", ln):
            cur = [int(L) for f, L in re.findall(r"([^\s:\[\]@;]+):(\d+):\d+", ln.split(";", 1)[1]) if f.endswith("mocca_device.h")] if ";" in ln else []
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        idx += 1
        op, _, args = t.partition(" ")
        args = args.split(";")[0].strip()
        for name, lo, hi in ranges:
            if name in done:
                continue
            if any(lo <= L <= hi for L in cur):
                seqs[name].append((op, args)); last_hit[name] = idx
            elif last_hit[name] is not None and idx - last_hit[name] > 400:   # the first occurrence has ended (the next one is another instance)
                done.add(name)
    print("range                                                              instr  VALU trans  DPP  LDSr LDSw SALU  nop  wait | issue [cyc]  longest chain: instr, cycles | bound")
    for name, lo, hi in ranges:
        mix, issue, (cyc, n) = analyse(seqs[name])
        tot = sum(mix[k] for k in ("valu", "trans", "lds_rd", "lds_wr", "salu", "smem", "vmem"))
        print("%-66s %5d %5d %5d %4d %5d %4d %4d %4d %5d | %11d  %20d %7.0f | %s"
              % (name + " [%d-%d]" % (lo, hi), tot, mix["valu"], mix["trans"], mix["dpp"], mix["lds_rd"], mix["lds_wr"], mix["salu"] + mix["smem"],
                 mix["nop_cycles"], mix["waitcnt"], issue, n, cyc, "ISSUE" if issue > cyc else "chain"))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/k.s")
