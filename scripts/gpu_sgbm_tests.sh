#!/bin/bash
# Runs on the GPU box: the matcher's stage tests, once per value of an environment switch ("VAR=a,b"), compact summary.
mkdir -p gpurun_out
var=${1%%=*}; vals=${1#*=}
for v in ${vals//,/ }; do
    export $var=$v
    timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -k "sgbm" --timeout 200 --timeout-method thread > gpurun_out/sgbm_tests_$v.log 2>&1
    echo "$var=$v rc=$? $(tail -1 gpurun_out/sgbm_tests_$v.log)"
    grep -E "^FAILED" gpurun_out/sgbm_tests_$v.log | head -12
done
