cd $GRAFT_REPO_ROOT
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])"; }
for t in 1025 2047; do for i in 1 2; do
  python bench_fftfilt.py --no-cpu-baseline --no-sweep --taps $t --steps 100 | pr "taps$t default"
  CSDR_AMD_FFTFILT_LDS_MODE=4 python bench_fftfilt.py --no-cpu-baseline --no-sweep --taps $t --steps 100 | pr "taps$t mode4"
  CSDR_AMD_FFTFILT_LDS_MODE=2 python bench_fftfilt.py --no-cpu-baseline --no-sweep --taps $t --steps 100 | pr "taps$t mode2"
done; done
