#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
for i in 1 2 3; do
 for lib in default libcsdr_amd_f1.so libcsdr_amd_f2.so libcsdr_amd_f3.so poly; do
  k5=1; if [ "$lib" = default ] || [ "$lib" = poly ]; then unset CSDR_AMD_LIB; else export CSDR_AMD_LIB=$PWD/csdr_amd/$lib; fi
  [ "$lib" = poly ] && k5=0
  CSDR_AMD_FIR_MFMA5=$k5 timeout 100 python bench_fir.py --steps 200 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']; print('%-20s %-12s kernel %.4f ms frac %.4f' % ('$lib', r['kernel'], r['kernel_avg_ms'], r['frac']))"
 done
done
