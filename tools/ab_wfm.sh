#!/bin/bash
# tools/ab_wfm.sh [alt.so] -- A/B of bench.py between this tree's library and another build of it on ONE box, interleaved (boxes and consecutive runs differ by
# several per cent).  alt.so: e.g. a tree with other kernel sources built with `make -C csdr_amd/csrc TARGET=../libcsdr_amd_alt.so OBJDIR=build_alt ../libcsdr_amd_alt.so`
# (an in-tree .so travels to the GPU box).  Run through gpurun:  gpurun -- 'bash tools/ab_wfm.sh csdr_amd/libcsdr_amd_alt.so'
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
alt=${1:-csdr_amd/libcsdr_amd_alt.so}
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['hip_event_ms_per_step_all_kernels'])"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 | pr new20
  [ -f $alt ] && CSDR_AMD_LIB=$PWD/$alt python bench.py --no-cpu-baseline --steps 20 --warmup 5 | pr alt20
done
for i in 1 2; do
  python bench.py --no-cpu-baseline --steps 300 | pr new300
  [ -f $alt ] && CSDR_AMD_LIB=$PWD/$alt python bench.py --no-cpu-baseline --steps 300 | pr alt300
done
