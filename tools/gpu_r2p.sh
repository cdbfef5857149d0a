#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_cli_gpu.py -q -k "fastddc" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
