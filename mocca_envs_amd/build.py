"""Build libmocca_hip.so (hand-written HIP for gfx950) in-tree with hipcc."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libmocca_hip.so")
SOURCES = ["mocca_api.hip", "mocca_task.hip", "mocca_r32.hip", "mocca_r64.hip"]   # physics kernels + C ABI; task-layer (INJECT) kernel instances; compact (32-row) step kernels; 64-row accuracy instance
DEPS = SOURCES + ["mocca_kernels.h", "mocca_device.h", "topo_walker3d.h", "topo_cassie.h", "topo_walker2d.h", "topo_crab2d.h", "topo_laikago.h"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, d) for d in DEPS] + [os.path.join(INCLUDE, h) for h in ("mocca.h", "mocca_model.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False, extra_flags=(), out: str = None) -> str:
    """hipcc --offload-arch=gfx950 -O3 -shared -fPIC; cross-compiles without a GPU."""
    if not force and out is None and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # -fno-slp-vectorize: the SLP vectoriser pairs scalar fp32 ops into v_pk_* but pays ~2 v_mov per pair to line
    # up register pairs; measured on this kernel it ADDS 12 % VALU instructions
    # MOCCA_HIPCC_FLAGS: extra compiler flags for experiments (e.g. "-mllvm -amdgpu-sched-strategy=max-ilp"); never set for the product build
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-I" + INCLUDE, "-I" + CSRC] + list(extra_flags) \
        + os.environ.get("MOCCA_HIPCC_FLAGS", "").split()
    if verbose:
        base.insert(1, "-Rpass-analysis=kernel-resource-usage")
    objdir = os.path.join(HERE, "build") if out is None else out + ".objs"
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src in SOURCES:   # the translation units compile in parallel
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = base + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out or LIB] + objs)
    return out or LIB


if __name__ == "__main__":
    # python -m mocca_envs_amd.build [-v] [--out /tmp/libX.so] [--src DIR] [-DFLAG ...]   (diagnostic / A-B builds pass -D flags and --out)
    argv = sys.argv[1:]
    out = argv[argv.index("--out") + 1] if "--out" in argv else None
    if "--src" in argv:   # build another revision's kernel sources (tools/ab.sh)
        CSRC = argv[argv.index("--src") + 1]
    print(build_lib(force=True, verbose="-v" in argv, extra_flags=[a for a in argv if a.startswith("-D")], out=out))
