#!/bin/bash
# Builds laser_3d_reconstruction_amd/libl3d_hip.so for gfx950 (in-tree) and the CPU oracle.
set -e
cd "$(dirname "$0")"
SRC=laser_3d_reconstruction_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
      -Wall -Wno-unused-result -Wno-pass-failed \
      -o laser_3d_reconstruction_amd/libl3d_hip.so \
      $SRC/capi.hip $SRC/sgbm.hip $SRC/wls.hip $SRC/depth.hip $SRC/steger.hip $SRC/simple.hip $SRC/remap.hip $SRC/pointcloud.hip "$@"
make -s -C oracle
