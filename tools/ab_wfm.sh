cd $GRAFT_REPO_ROOT
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['hip_event_ms_per_step_all_kernels'])"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 | pr new20
  CSDR_AMD_LIB=$PWD/csdr_amd/libcsdr_amd_oldwfm.so python bench.py --no-cpu-baseline --steps 20 --warmup 5 | pr old20
done
python bench.py --no-cpu-baseline --steps 300 | pr new300
CSDR_AMD_LIB=$PWD/csdr_amd/libcsdr_amd_oldwfm.so python bench.py --no-cpu-baseline --steps 300 | pr old300
python bench.py --no-cpu-baseline --steps 300 | pr new300
CSDR_AMD_LIB=$PWD/csdr_amd/libcsdr_amd_oldwfm.so python bench.py --no-cpu-baseline --steps 300 | pr old300
