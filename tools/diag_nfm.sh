#!/bin/bash
# tools/diag_nfm.sh -- what bounds k_ddc_mfma: the full kernel, its DMA ring + barriers alone (DDC_DIAG=1) and its math alone (DDC_DIAG=2), front end only, on ONE box.
# The two diagnostic libraries are built in tree first (they travel to the GPU box):
#   make -C csdr_amd/csrc -j8 OBJDIR=build_d1 TARGET=../libcsdr_amd_d1.so EXTRA=-DDDC_DIAG=1 ../libcsdr_amd_d1.so
#   make -C csdr_amd/csrc -j8 OBJDIR=build_d2 TARGET=../libcsdr_amd_d2.so EXTRA=-DDDC_DIAG=2 ../libcsdr_amd_d2.so
cd $GRAFT_REPO_ROOT
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_avg_ms'])"; }
for i in 1 2; do
  python bench_nfm.py --no-cpu-baseline --steps 200 --front-end-only | pr full_kernel
  CSDR_AMD_LIB=$PWD/csdr_amd/libcsdr_amd_d1.so python bench_nfm.py --no-cpu-baseline --steps 200 --front-end-only | pr dma_and_barriers_only
  CSDR_AMD_LIB=$PWD/csdr_amd/libcsdr_amd_d2.so python bench_nfm.py --no-cpu-baseline --steps 200 --front-end-only | pr math_only
done
