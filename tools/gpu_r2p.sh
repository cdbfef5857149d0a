#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2t
timeout 800 python tools/bench_ops.py 2>/dev/null > gpurun_out/r2t/r2t_ops.jsonl; wc -l gpurun_out/r2t/r2t_ops.jsonl; cut -c1-150 gpurun_out/r2t/r2t_ops.jsonl | tail -14
