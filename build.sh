#!/bin/bash
# Builds laser_3d_reconstruction_amd/libl3d_hip.so for gfx950 (in-tree) and the CPU oracle.
# Each source is its own translation unit (every kernel is launched from the file that defines it), so the
# files compile in parallel and are linked at the end.
set -e
cd "$(dirname "$0")"
SRC=laser_3d_reconstruction_amd/csrc
OBJ=laser_3d_reconstruction_amd/csrc/_obj
mkdir -p $OBJ
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -Wno-pass-failed"
pids=()
for f in capi sgbm sgbm_paths wls depth steger simple remap pointcloud; do
    hipcc $FLAGS -c -o $OBJ/$f.o $SRC/$f.hip "$@" &
    pids+=($!)
done
# the horizontal kernel alone is scheduled for instruction-level parallelism (see the head of the file)
hipcc $FLAGS ${L3D_HORIZONTAL_FLAGS--mllvm -amdgpu-sched-strategy=max-ilp} -c -o $OBJ/sgbm_horizontal.o $SRC/sgbm_horizontal.hip "$@" &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o laser_3d_reconstruction_amd/libl3d_hip.so $OBJ/*.o
make -s -C oracle
