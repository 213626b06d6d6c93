#!/bin/bash
# Runs on the GPU box: the matcher's parity tests, then short bench runs of the configurations whose dominant kernel is
# the vertical one; one summary line each.  usage: scripts/gpu_r3_vertical.sh <tag> ["C3 C5 REFDEFAULT"]
tag=$1; cfgs=${2:-"C3 C5 REFDEFAULT"}
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q --timeout 180 --timeout-method thread -k "sgbm or full or stage or matcher or chain" > gpurun_out/r3v_${tag}_tests.log 2>&1 || { tail -30 gpurun_out/r3v_${tag}_tests.log; exit 1; }
tail -1 gpurun_out/r3v_${tag}_tests.log
for c in $cfgs; do
    timeout -k 10 200 python bench.py --config $c --steps 6 --warmup 2 --repeats 3 --no-cpu-baseline --no-also > gpurun_out/r3v_${tag}_$c.json 2> gpurun_out/r3v_${tag}_$c.err || { tail -5 gpurun_out/r3v_${tag}_$c.err; exit 1; }
    python - $c $tag <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r3v_{sys.argv[2]}_{sys.argv[1]}.json"))
print(sys.argv[1], sys.argv[2], "fps %.1f ms %.2f ok %s" % (d["value"], d["ms_per_step"], d["verified"]), {k: round(v, 3) for k, v in d["kernel_ms_per_launch"].items() if "vertical" in k or "horizontal" in k})
PY
done
