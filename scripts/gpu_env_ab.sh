#!/bin/bash
# Runs on the GPU box: bench.py for one configuration, alternating between environment settings; one line each.
# usage: scripts/gpu_env_ab.sh <config> "<VAR=value ...>" "<VAR=value ...>" ...   (an empty string = no setting)
cfg=$1; shift
mkdir -p gpurun_out
i=0
for rep in 1 2; do i=0; for e in "$@"; do i=$((i+1))
    env $e timeout -k 10 200 python bench.py --config $cfg --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-also > gpurun_out/envab_${cfg}_$i.json 2> gpurun_out/envab_${cfg}_$i.err || { tail -5 gpurun_out/envab_${cfg}_$i.err; exit 1; }
    python - "$cfg" "$e" $i <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/envab_{sys.argv[1]}_{sys.argv[3]}.json"))
print(sys.argv[1], "[" + sys.argv[2] + "]", "fps %.1f ms %.3f ok %s" % (d["value"], d["ms_per_step"], d["verified"]), {k: round(v, 3) for k, v in d["kernel_ms_per_launch"].items()})
PY
done; done
