#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -x -m gpu -k "adpcm or f3 or nfm_fused" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
timeout 600 python tools/bench_ops.py 2>/dev/null | grep -i "adpcm"
