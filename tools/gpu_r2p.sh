#!/bin/bash
# fastddc A/B on one box: libcsdr_amd_A.so (previous commit) vs the working tree
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_gpu_parity.py -q -x -k "c4 or bank or fastddc" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for rep in 1 2; do
for lib in libcsdr_amd_A.so libcsdr_amd.so; do
  CSDR_AMD_LIB=$PWD/csdr_amd/$lib timeout 200 python bench_fastddc.py --steps 300 --no-cpu-baseline --verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_avg_ms'], d['verify']['ok'], d['verify']['max_rel_rms'])"
done
done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2p_new -- python bench_fastddc.py --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
f=$(find gpurun_out/r2p_new -name "*kernel_stats.csv" | head -1); grep "k_ddc" $f | python -c "
import sys,csv,re
for r in csv.reader(sys.stdin):
    print('   %-28s calls %s avg %.1f us' % (re.search(r'k_ddc_\w+(<[^>]*>)?', r[0]).group(0), r[1], float(r[3])/1e3))"
