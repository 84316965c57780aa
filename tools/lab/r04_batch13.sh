#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b13
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -k "setter or reset_pid or graph" 2>&1 | tail -15
