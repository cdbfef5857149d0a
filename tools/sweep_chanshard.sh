#!/bin/bash
# tools/sweep_chanshard.sh -- emulated per-rank time of the channel-sharded bank (bench_fastddc.py --emulate-world, null transport), fold of few channel rows on / off
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/chanshard
for cfg in "8 64" "8 512" "4 256" "2 128"; do
  set -- $cfg
  for n in 1 0; do
    CSDR_AMD_DDC_NARROW=$n timeout 100 python bench_fastddc.py --emulate-world $1 --shard channels --blocks $2 --steps 100 2>/dev/null | grep '^{' | tail -1 > gpurun_out/chanshard/w$1_b$2_narrow$n.json
    python - gpurun_out/chanshard/w$1_b$2_narrow$n.json <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print("world %d blocks %4d %s: one GPU %.4f ms, slowest rank %.4f ms, compute-only scaling %.2f" % (d["world"], d["blocks_per_global_batch"], sys.argv[1].split("_")[-1][:7], d["t1_ms_single_gpu_same_blocks"], d["t_rank_ms_worst"], d["compute_only_scaling"]))
P
  done
done
