#!/bin/bash
# usage: scripts/gpu_stats.sh <tag> [bench.py arguments...]  -> per-kernel averages of a short bench run
tag=$1; shift
ROOT=$GRAFT_REPO_ROOT
out=$ROOT/gpurun_out/stats_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o vm -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $out/bench.json 2> $out/err.log || { tail -5 $out/err.log; exit 1; }
python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$out/vm_kernel_stats.csv")))
for r in rows[:22]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), ("%.1f us" % (float(r["AverageNs"]) / 1e3)).rjust(12), r["Percentage"].rjust(7))
d = json.load(open("$out/bench.json"))
print("fps", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2))
PY
