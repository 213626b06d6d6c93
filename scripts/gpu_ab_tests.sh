#!/bin/bash
# Runs on the GPU box: the matcher's GPU tests with the library in the tree, then scripts/gpu_ab.sh over the given builds.
# usage: scripts/gpu_ab_tests.sh <name> <name> ... [-- bench.py arguments]
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -q -x -k "sgbm or speckle or median or full_size or bench_shape or real_pairs or smoke" --timeout 300 --timeout-method thread > gpurun_out/ab_tests.log 2>&1
rc=$?; tail -2 gpurun_out/ab_tests.log; grep -E "^(E |FAILED)" gpurun_out/ab_tests.log | head -8
[ $rc -ne 0 ] && exit $rc
cp laser_3d_reconstruction_amd/libl3d_hip.so /tmp/l3d_keep.so
bash scripts/gpu_ab.sh "$@"; rc=$?
cp /tmp/l3d_keep.so laser_3d_reconstruction_amd/libl3d_hip.so
exit $rc
