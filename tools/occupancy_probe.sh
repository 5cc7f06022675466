#!/bin/bash
# How much does the step kernel's throughput at saturation depend on the number of resident waves per SIMD?  The same code with its LDS
# padded (diagnostic build, -DMOCCA_LDS_PAD=<floats>) so that 3 or 2 waves fit a SIMD instead of 4, timed at 8192 and 16384 envs -- the slope
# bounds what a fifth wave (<= 96 VGPRs, <= 8 KB LDS: a re-layout of the ABA view) could buy.  usage: tools/occupancy_probe.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export MOCCA_ALLOW_DIAGNOSTIC_BUILD=1
for pad in 0 800 2600; do   # 10192 B -> 4 waves/SIMD; 13392 B -> 3 (12 per CU: 160 KB / 12 = 13.3 KB); 20592 B -> 7 per CU ~ 1.75/SIMD
  if [ $pad = 0 ]; then unset MOCCA_LIB_PATH; else
    python -m mocca_envs_amd.build --out /tmp/lib_pad$pad.so -DMOCCA_LDS_PAD=$pad > /dev/null || exit 1
    export MOCCA_LIB_PATH=/tmp/lib_pad$pad.so
  fi
  for envs in 4096 8192 16384; do
    python bench.py --envs $envs --steps 200 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_info']; print(json.dumps({'lds_pad_floats': $pad, 'lds_bytes': k['lds_bytes'], 'blocks_per_cu': k['max_blocks_per_cu'], 'envs': $envs, 'kernel_us': round(d['roofline']['kernel_ms']*1000,1), 'env_steps_per_s': round(d['value'])}))"
  done
done
