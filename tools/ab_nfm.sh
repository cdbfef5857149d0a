cd $GRAFT_REPO_ROOT
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_avg_ms'])"; }
for i in 1 2 3; do
  python bench_nfm.py --no-cpu-baseline --steps 200 | pr whole
  CSDR_AMD_DDC_WHOLE=0 python bench_nfm.py --no-cpu-baseline --steps 200 | pr split
done
