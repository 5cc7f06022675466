#!/bin/bash
# Round profile of the bench on the GPU box: kernel-trace stats + separate PMC passes (HBM traffic, issue mix).
# usage (via gpurun): tools/profile_round.sh <tag> [env-id] [envs]   -> gpurun_out/<tag>_*; copy what matters to profiles/
tag=${1:-r01}; envid=${2:-Walker3DCustomEnv-v0}; envs=${3:-4096}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_trace -- python3 $R/bench.py --env-id $envid --envs $envs --steps 100 --warmup 10 --no-cpu-baseline > $O/${tag}_bench.json 2> $O/${tag}_bench.err
cp $(find $O/${tag}_trace -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" "TCC_HIT TCC_MISS TCC_REQ" \
            "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU" \
            "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/${tag}_pmc_$n -- python3 $R/bench.py --env-id $envid --envs $envs --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
done
python3 - $O $tag <<'PY'
import csv, glob, json, sys, collections
O, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/{tag}_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mocca_step_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
out["_launches_averaged"] = {k: len(v) for k, v in agg.items()}
json.dump(out, open(f"{O}/{tag}_pmc_summary.json", "w"), indent=1)
print(json.dumps(out))
PY
