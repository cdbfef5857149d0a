cd $GRAFT_REPO_ROOT
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_avg_ms'])"; }
for i in 1 2; do
  python bench_nfm.py --no-cpu-baseline --steps 200 --front-end-only | pr full_kernel
  CSDR_AMD_LIB=$PWD/csdr_amd/libcsdr_amd_d1.so python bench_nfm.py --no-cpu-baseline --steps 200 --front-end-only | pr dma_and_barriers_only
  CSDR_AMD_LIB=$PWD/csdr_amd/libcsdr_amd_d2.so python bench_nfm.py --no-cpu-baseline --steps 200 --front-end-only | pr math_only
done
