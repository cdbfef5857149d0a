#!/bin/bash
# round-2 GPU batch J: the whole GPU suite after the CLI / shifter / FFT changes, per-operator survey, CLI pipe bench
cd $GRAFT_REPO_ROOT; out=gpurun_out/r2j; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q --tb=short > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -25 $out/pytest.log | cut -c1-250
timeout 300 python tools/bench_ops.py > $out/r2j_ops.jsonl 2> $out/ops.err; cut -c1-200 $out/r2j_ops.jsonl
timeout 300 bash tools/bench_cli.sh > $out/cli_bench.txt 2>&1; cat $out/cli_bench.txt
