#!/bin/bash
# Runs on the GPU box: Steger tests, then scripts/dev_steger.py with each of the builds kept under
# _ab/<name>/libl3d_hip.so, alternating, twice over.  usage: scripts/gpu_steger_ab.sh <name> <name> ...
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests -m gpu -q -k "steger or smoke or real" --timeout 200 --timeout-method thread > gpurun_out/steger_tests.log 2>&1
rc=$?; tail -2 gpurun_out/steger_tests.log; grep -E "^(E |FAILED)" gpurun_out/steger_tests.log | head -8
[ $rc -ne 0 ] && exit $rc
cp laser_3d_reconstruction_amd/libl3d_hip.so /tmp/l3d_keep.so
for rep in 1 2; do for v in "$@"; do
    cp _ab/$v/libl3d_hip.so laser_3d_reconstruction_amd/libl3d_hip.so || exit 1
    echo "== $v"; timeout -k 10 120 python scripts/dev_steger.py || exit 1
done; done
cp /tmp/l3d_keep.so laser_3d_reconstruction_amd/libl3d_hip.so
