#!/bin/bash
# Per-kernel times of any python program on the GPU box: scripts/prof_any.sh <tag> <script.py> [args ...]
tag=$1; shift
export TMPDIR=/tmp
out=/tmp/prof_$tag
rm -rf $out
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d $out -o p --output-format csv -- python3 "$@" > gpurun_out/${tag}_prof_stdout.log 2>&1
f=$(find $out -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:18]:
    print("%-96s calls %6s  avg %10.1f us  total %6.2f%%" % (r["Name"][:96], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
