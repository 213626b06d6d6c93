#!/bin/bash
# Runs on the GPU box: the smoother's tests with the library in the tree, then bench.py of the given configs with each of
# the builds kept under _ab/<name>/libl3d_hip.so, alternating.  usage: scripts/gpu_ab_wls.sh "C3 REFDEFAULT" <name> <name> ...
cfgs=$1; shift
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -q -k "wls or real_pairs or full_size or camera_manager" --timeout 300 --timeout-method thread > gpurun_out/wls_tests.log 2>&1
rc=$?; tail -2 gpurun_out/wls_tests.log; grep -E "^(E |FAILED)" gpurun_out/wls_tests.log | head -8
[ $rc -ne 0 ] && exit $rc
cp laser_3d_reconstruction_amd/libl3d_hip.so /tmp/l3d_keep.so
for c in $cfgs; do for rep in 1 2; do for v in "$@"; do
    cp _ab/$v/libl3d_hip.so laser_3d_reconstruction_amd/libl3d_hip.so || exit 1
    timeout -k 10 280 python bench.py --config $c --steps 8 --warmup 2 --repeats 3 --no-cpu-baseline > gpurun_out/wls_${c}_$v.json 2> gpurun_out/wls_${c}_$v.err || { tail -3 gpurun_out/wls_${c}_$v.err; exit 1; }
    python - $c $v <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/wls_{sys.argv[1]}_{sys.argv[2]}.json"))
k = d["kernel_ms_per_launch"]
print(sys.argv[1], sys.argv[2], "fps %.1f ms/step %.2f verified %s wls stage %.2f ms" % (d["value"], d["ms_per_step"], d["verified"], d["stage_ms_per_step_serialised"]["wls"]), {n: round(v, 3) for n, v in k.items() if "fgs" in n})
PY
done; done; done
cp /tmp/l3d_keep.so laser_3d_reconstruction_amd/libl3d_hip.so
