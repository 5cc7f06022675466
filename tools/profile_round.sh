#!/bin/bash
# Round profile of the bench on the GPU box: kernel-trace stats + separate PMC passes (HBM traffic, issue mix).
# Every pass warms up for 200 steps first: the counters are taken at the steady state of the auto-reset workload
# (~4 % of the envs reset per step), not on the first steps after a reset; averages are over the LAST 20 launches.
# usage (via gpurun): [BENCH_EXTRA="--max-rows 32"] tools/profile_round.sh <tag> [env-id] [envs]   -> gpurun_out/<tag>_*; copy what matters to profiles/
# BENCH_EXTRA: further bench.py arguments for every pass (e.g. the compact kernel instance)
tag=${1:-r01}; envid=${2:-Walker3DCustomEnv-v0}; envs=${3:-4096}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_trace -- python3 $R/bench.py --env-id $envid --envs $envs --steps 400 --warmup 100 --no-cpu-baseline --preroll-seconds 0 $BENCH_EXTRA > $O/${tag}_bench.json 2> $O/${tag}_bench.err
cp $(find $O/${tag}_trace -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" "TCC_HIT TCC_MISS TCC_REQ" \
            "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU" \
            "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/${tag}_pmc_$n -- python3 $R/bench.py --env-id $envid --envs $envs --steps 20 --warmup 200 --no-cpu-baseline --preroll-seconds 0 $BENCH_EXTRA > /dev/null 2>&1
done
python3 - $O $tag <<'PY'
import csv, glob, json, sys, collections
O, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/{tag}_pmc_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "mocca_step_kernel" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, v in per.items():   # steady state only: the last 20 launches of the pass (the first 200 are warm-up)
        agg[k] += [x for _, x in sorted(v)[-20:]]
out = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
out["_launches_averaged"] = {k: len(v) for k, v in agg.items()}
json.dump(out, open(f"{O}/{tag}_pmc_summary.json", "w"), indent=1)
print(json.dumps(out))
PY
python3 - $O $tag $R $envs <<'PY'
# profiles/traffic.json candidate: HBM bytes + VALU work per launch, tied to the kernel sources they were measured on
import json, sys
O, tag, R, envs = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
sys.path.insert(0, R)
import bench
p = json.load(open(f"{O}/{tag}_pmc_summary.json"))
t = {"kernel": "mocca_step_kernel", "envs_per_launch": envs, "kernel_source_sha256": bench.kernel_source_hash(),
     "FETCH_SIZE_KB": p["FETCH_SIZE"], "WRITE_SIZE_KB": p["WRITE_SIZE"], "TCC_EA0_RDREQ": p.get("TCC_EA0_RDREQ"),
     "TCC_EA0_WRREQ": p.get("TCC_EA0_WRREQ"),
     "traffic_bytes_per_launch": 1024 * (p["FETCH_SIZE"] + p["WRITE_SIZE"]),
     "traffic_bytes_per_launch_if_fetch_doubled": 1024 * (2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]),
     "valu": {"valu_insts_per_env_step": p["SQ_INSTS_VALU"] / envs,
              "active_lane_ops_per_launch": p["SQ_THREAD_CYCLES_VALU"] / 4.0,
              "active_lanes_per_valu_inst": p["SQ_THREAD_CYCLES_VALU"] / 4.0 / p["SQ_INSTS_VALU"]},
     "method": "rocprofv3 --pmc, separate passes (FETCH_SIZE; WRITE_SIZE; TCC_EA0_*; SQ_*) over `bench.py --steps 20 --warmup 200` "
               "(tools/profile_round.sh): steady state of the auto-reset workload, averaged over the last 20 launches; KB -> bytes x1024. "
               "FETCH_SIZE is reported uncorrected: the gfx950 x2 correction of MI355X_MICROARCH.md is calibrated for 16-B-per-lane "
               "streams, this kernel issues 4-B-per-lane loads (TCC_EA0_RDREQ x 64 B equals FETCH_SIZE); the doubled figure is an upper "
               "bound. active_lane_ops = SQ_THREAD_CYCLES_VALU / 4 (a wave64 VALU instruction occupies each active lane for 4 cycles).",
     "source": f"profiles/{tag}_pmc_summary.json"}
json.dump(t, open(f"{O}/{tag}_traffic.json", "w"), indent=1)
PY
# the raw per-dispatch traces are bulky (gpurun merges at most 64 MiB back): only the summaries above are kept
find $O -maxdepth 1 -type d \( -name "${tag}_pmc_*" -o -name "${tag}_trace" \) -exec rm -rf {} +
