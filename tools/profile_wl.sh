#!/bin/bash
# usage (GPU box): bash tools/profile_wl.sh <tag> <bench args...>   kernel-trace stats only
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-300
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/trace/*/*_kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:28]:
    print("%-70s calls %5s tot_ms %8.3f avg_us %9.1f  %5s%%" % (r["Name"][:70], r["Calls"], int(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
