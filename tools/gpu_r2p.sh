#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "bandpass" 2>&1 | grep -E "passed|failed|Error|error" | tail -3
for rep in 1 2; do
for mode in 0 1 3; do
  CSDR_AMD_FFTFILT_LDS_MODE=$mode timeout 200 python bench_fftfilt.py --steps 100 --no-sweep --no-cpu-baseline --verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mode $mode', d['value'], d['ms_per_step'], d['roofline']['frac'], d['verify']['ok'], d['verify']['max_rel_rms'])"
done
done
for mode in 0 2; do
  CSDR_AMD_FFTFILT_LDS_MODE=$mode timeout 200 python bench_fftfilt.py --steps 100 --no-sweep --no-cpu-baseline --taps 2047 --verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('taps 2047 mode $mode', d['value'], d['ms_per_step'], d['roofline']['frac'], d['verify']['ok'])"
done
