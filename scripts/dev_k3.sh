#!/bin/bash
# development: parity subset + per-kernel times + traffic of the two scan kernels
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "sgbm" 2>&1 | tail -2
./scripts/gpu_cycle.sh | tail -3
./scripts/gpu_traffic.sh k3 | grep -E "k_horizontal|k_cost"
