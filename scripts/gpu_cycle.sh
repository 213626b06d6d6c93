#!/bin/bash
# Runs on the GPU box: GPU parity tests, then a short bench; prints a compact summary.
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -2 gpurun_out/gpu_tests.log
grep -E "^E " gpurun_out/gpu_tests.log | head -5
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/bench_cycle.json 2> gpurun_out/bench_cycle.err || { tail -5 gpurun_out/bench_cycle.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_cycle.json"))
print("fps", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3))
print({k: round(v, 3) for k, v in d["kernel_ms_per_launch"].items()})
print("roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), "pipeline", round(d["pipeline_roofline"]["frac"], 4), "steger Mpix/s", round(d["steger_mpix_per_s"]))
PY
