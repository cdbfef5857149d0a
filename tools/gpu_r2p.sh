#!/bin/bash
# A/B on one box: ring pitch + 16 (libcsdr_amd_A.so) vs + 32 in the WFM and NFM front ends
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in libcsdr_amd_A.so libcsdr_amd.so; do
  CSDR_AMD_LIB=$PWD/csdr_amd/$lib timeout 200 python bench.py --steps 300 --no-cpu-baseline --verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wfm $lib', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_avg_ms'], d['verify']['ok'])"
  CSDR_AMD_LIB=$PWD/csdr_amd/$lib timeout 200 python bench_nfm.py --steps 200 --no-cpu-baseline --verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nfm $lib', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_avg_ms'], d['verify']['ok'])"
done
done
