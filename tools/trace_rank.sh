#!/bin/bash
# tools/trace_rank.sh <world> <rank> <shard> <blocks> -- kernel trace of ONE rank's emulated batch (bench_fastddc.py --emulate-world): per-kernel statistics and one
# steady-state batch as a timeline (start / end / duration in us, queue, kernel)
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
R=$PWD; W=${1:-8}; RK=${2:-3}; SH=${3:-channels}; NB=${4:-64}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python $R/bench_fastddc.py --emulate-world $W --rank $RK --shard $SH --blocks $NB --steps 60 > /tmp/tr.log 2>&1; tail -3 /tmp/tr.log
f=$(find /tmp/tr -name "*kernel_stats.csv" | head -1); t=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
cut -d, -f1-4 "$f" | head -16
python3 - "$t" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-60:-20]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f %7.1f  q=%s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
P
