#!/bin/bash
# A/B on one box: libcsdr_amd_A.so (plain f32 butterflies) vs libcsdr_amd.so (packed), interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "libcsdr_amd_A.so 0" "libcsdr_amd_A.so 2" "libcsdr_amd.so 2" "libcsdr_amd.so 0"; do
  set -- $cfg
  CSDR_AMD_LIB=$PWD/csdr_amd/$1 CSDR_AMD_FFTFILT_LDS_MODE=$2 timeout 200 python bench_fftfilt.py --steps 50 --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 mode $2', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
done
