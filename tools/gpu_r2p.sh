#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu -k "wfm or edges or golden or chain or compat" 2>&1 | grep -E "passed|failed|Error|error" | tail -3
timeout 200 python bench.py --steps 300 --no-cpu-baseline --verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wfm', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['verify']['ok'])"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2p_wfm -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
f=$(find gpurun_out/r2p_wfm -name "*kernel_stats.csv" | sort | tail -1); grep "k_wfm" $f | cut -d, -f1-4 | cut -c1-110
