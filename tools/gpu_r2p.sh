#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu -k "nfm or ddc" 2>&1 | grep -E "passed|failed|Error|error" | tail -3
timeout 200 python bench_nfm.py --steps 200 --no-cpu-baseline --verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nfm', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['verify']['ok'])"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2p_nfm -- python bench_nfm.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
f=$(find gpurun_out/r2p_nfm -name "*kernel_stats.csv" | sort | tail -1); grep "k_" $f | cut -d, -f1-4 | cut -c1-110 | head -14
