#!/bin/bash
# A/B on one box: fastddc with / without SLP packing in the transform kernels
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in libcsdr_amd.so libcsdr_amd_A.so; do
  CSDR_AMD_LIB=$PWD/csdr_amd/$lib timeout 200 python bench_fastddc.py --steps 300 --no-cpu-baseline --verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['roofline']['frac'], d['verify']['ok'], d['verify'].get('max_rel_rms'))"
done
done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in libcsdr_amd.so libcsdr_amd_A.so; do
  CSDR_AMD_LIB=$PWD/csdr_amd/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2p_$lib -- python bench_fastddc.py --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  f=$(find gpurun_out/r2p_$lib -name "*kernel_stats.csv" | head -1); echo $lib; grep "k_ddc" $f | cut -d, -f1-4 | cut -c1-120
done
