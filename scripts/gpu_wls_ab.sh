#!/bin/bash
# Runs on the GPU box: the smoother's tests, then bench.py of the given configs with the horizontal pass on the planes as
# they lie (0) and on transposed copies (1).  usage: scripts/gpu_wls_ab.sh "C3 REFDEFAULT"
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -q -k "wls or real_pairs or full_size or camera_manager" --timeout 300 --timeout-method thread > gpurun_out/wls_tests.log 2>&1
rc=$?; tail -2 gpurun_out/wls_tests.log; grep -E "^(E |FAILED)" gpurun_out/wls_tests.log | head -8
[ $rc -ne 0 ] && exit $rc
for c in $1; do for t in 0 1 0 1; do
    L3D_WLS_TRANSPOSE=$t timeout -k 10 280 python bench.py --config $c --steps 8 --warmup 2 --repeats 3 --no-cpu-baseline > gpurun_out/wls_${c}_$t.json 2> gpurun_out/wls_${c}_$t.err || { tail -3 gpurun_out/wls_${c}_$t.err; exit 1; }
    python - $c $t <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/wls_{sys.argv[1]}_{sys.argv[2]}.json"))
print(sys.argv[1], "transpose", sys.argv[2], "fps %.1f ms/step %.2f verified %s wls stage %.2f ms" % (d["value"], d["ms_per_step"], d["verified"], d["stage_ms_per_step_serialised"]["wls"]))
PY
done; done
