#!/bin/bash
# Runs on the GPU box: per-kernel averages of the Steger extractor alone (scripts/dev_steger.py under rocprofv3).
ROOT=$GRAFT_REPO_ROOT
out=$ROOT/gpurun_out/steger_stats
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o vm -- python3 $ROOT/scripts/dev_steger.py > $out/run.log 2> $out/err.log || { tail -5 $out/err.log; exit 1; }
cat $out/run.log
python3 - <<PY
import csv
for r in csv.DictReader(open("$out/vm_kernel_stats.csv")):
    if "steger" in r["Name"] or "fill" in r["Name"]:
        print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), ("%.1f us" % (float(r["AverageNs"]) / 1e3)).rjust(12), ("min %.1f" % (float(r["MinNs"]) / 1e3)).rjust(12))
PY
rm -f $out/vm_kernel_trace.csv
