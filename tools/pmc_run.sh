#!/bin/bash
# tools/pmc_run.sh <outdir> <kernel-name-substring> <command...> -- rocprofv3 PMC passes (counters only + --kernel-trace, one small
# counter set per pass) over <command>, then the per-kernel means.  Experiment helper for the GPU box (run through gpurun).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=$1; kern=$2; shift 2
mkdir -p $out
i=0
for set in "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "MeanOccupancyPerCU GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum" \
           "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES" "TA_BUSY_avr TA_FLAT_READ_LDS_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- "$@" > $out/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "$kern" not in k: continue
        k = k.split("(")[0][-48:] + " grid=" + r.get("Grid_Size", "?")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-44s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
