#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_configs_gpu.py -q -k "c4 or bank" 2>&1 | grep -E "passed|failed|Error|error" | tail -3
for rep in 1 2; do
for lib in libcsdr_amd_A.so libcsdr_amd.so; do
  CSDR_AMD_LIB=$PWD/csdr_amd/$lib timeout 200 python bench_fastddc.py --steps 300 --no-cpu-baseline --verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['verify']['ok'], d['verify']['max_rel_rms'])"
done
done
