#!/usr/bin/env python3
"""Copy what tools/round_artifacts.sh <tag> left under gpurun_out/ into profiles/ (tracked): bench lines, PMC summaries, step-kernel
rows of the rocprofv3 kernel stats, timelines; profiles/traffic.json <- <tag>_traffic.json.  usage: tools/refresh_profiles.py r02"""
import glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
for f in sorted(glob.glob(os.path.join(O, tag + "_*"))):
    b = os.path.basename(f)
    if os.path.isdir(f) or b.endswith(".err"):
        continue
    if b.endswith("_kernel_stats.csv"):
        L = open(f).read().split("\n")
        open(os.path.join(P, b), "w").write("\n".join([L[0]] + [l for l in L[1:] if "mocca" in l]) + "\n")
    elif b == tag + "_traffic.json":
        shutil.copy(f, os.path.join(P, "traffic.json"))
    elif b.endswith("_traffic.json"):
        continue
    elif b.endswith(".txt"):
        open(os.path.join(P, b), "w").write("".join(l for l in open(f) if "amdgpu.ids" not in l))
    else:
        shutil.copy(f, os.path.join(P, b))
d = json.load(open(os.path.join(P, tag + "_bench_full.json")))
print("headline", round(d["value"] / 1e6, 2), "M env-steps/s,", round(1e3 * d["roofline"]["kernel_ms"], 1), "us kernel; traffic hash",
      json.load(open(os.path.join(P, "traffic.json")))["kernel_source_sha256"])
