#!/bin/bash
# Runs on the GPU box: single-frame extraction latency (scripts/dev_latency_parts.py) with each of the builds kept under
# _ab/<name>/libl3d_hip.so, alternating, twice over.  usage: scripts/gpu_steger_ab1.sh <name> <name> ...
cp laser_3d_reconstruction_amd/libl3d_hip.so /tmp/l3d_keep.so
for rep in 1 2; do for v in "$@"; do
    cp _ab/$v/libl3d_hip.so laser_3d_reconstruction_amd/libl3d_hip.so || exit 1
    echo "== $v"; timeout -k 10 120 python scripts/dev_latency_parts.py 2>&1 | grep -E "extract_points|alone" || exit 1
done; done
cp /tmp/l3d_keep.so laser_3d_reconstruction_amd/libl3d_hip.so
