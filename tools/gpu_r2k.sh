#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r2k; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_edges_gpu.py tests/test_cli_gpu.py tests/test_compat_gpu.py tests/test_golden.py -m gpu -q --tb=short -k "fir or chain or am_and_ssb or golden or stream_bank" 2>&1 | tail -8 | cut -c1-250
timeout 300 python bench_fir.py --decimation 50 --tbw 0.005 --streams 64 --verify > $out/bench_fir50.json 2> $out/bench_fir50.err; tail -c 1500 $out/bench_fir50.json
CSDR_AMD_FIR_MFMA_OFF=1 timeout 300 python bench_fir.py --decimation 50 --tbw 0.005 --streams 64 --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('generic:', d['ms_per_step'], d['roofline']['frac'])"
grep -v amdgpu.ids $out/bench_fir50.err | tail -3
