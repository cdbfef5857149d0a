#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_configs_gpu.py -q -k "random or alternative or fused_forward" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -6
