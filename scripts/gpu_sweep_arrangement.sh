#!/bin/bash
# Runs on the GPU box: sweeps the stream arrangement of bench.py (pipelines in flight, where the laser extractor runs,
# hardware queues per process) with the library in the tree.  usage: scripts/gpu_sweep_arrangement.sh [bench.py arguments]
mkdir -p gpurun_out
for q in 4 8; do for p in 1 2 3; do for st in side main; do
    GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --steps 30 --warmup 3 --repeats 3 --no-cpu-baseline --no-verify --pipelines $p --steger-stream $st "$@" > gpurun_out/sweep.json 2> gpurun_out/sweep.err || { tail -5 gpurun_out/sweep.err; exit 1; }
    python - $q $p $st <<'PY'
import json, sys
d = json.load(open("gpurun_out/sweep.json"))
print("queues", sys.argv[1], "pipelines", sys.argv[2], "steger", sys.argv[3], "fps %.0f  ms/step %.3f  (min %.0f max %.0f)" % (d["value"], d["ms_per_step"], d["timing"]["value_min"], d["timing"]["value_max"]))
PY
done; done; done
