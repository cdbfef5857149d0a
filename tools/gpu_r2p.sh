#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_configs_gpu.py -q -k "c1 or c3" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -6
