#!/bin/bash
# Runs on the GPU box: rocprofv3 per-kernel averages of a short bench.py run, kernels whose name matches <pattern>.
# usage: scripts/gpu_kstats.sh <tag> <pattern> [bench.py arguments...]
tag=$1; pat=$2; shift 2
ROOT=$GRAFT_REPO_ROOT
out=$ROOT/gpurun_out/kstats_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o vm -- python3 $ROOT/bench.py --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-verify "$@" > $out/bench.json 2> $out/err.log || { tail -5 $out/err.log; exit 1; }
python3 - <<PY
import csv, re
for r in csv.DictReader(open("$out/vm_kernel_stats.csv")):
    if re.search("$pat", r["Name"]):
        print("$tag", r["Name"][:48].ljust(48), r["Calls"].rjust(5), ("%.1f us" % (float(r["AverageNs"]) / 1e3)).rjust(12), ("min %.1f" % (float(r["MinNs"]) / 1e3)).rjust(12))
PY
rm -f $out/vm_kernel_trace.csv
