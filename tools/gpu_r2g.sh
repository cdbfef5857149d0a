#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r2g; mkdir -p $out
sel="c4 or bank or fastddc or general"
timeout 600 python -m pytest tests/test_configs_gpu.py tests/test_gpu_parity.py tests/test_cli_gpu.py -m gpu -q --tb=short -k "$sel" > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
for ch in 1 0; do echo "== chains_side=$ch"; CSDR_AMD_DDC_CHAINS=$ch timeout 200 python bench_fastddc.py --steps 500 --no-cpu-baseline 2> $out/b_$ch.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"; done
