#!/bin/bash
# usage: bash tools/pmc.sh <tag> "<COUNTERS...>" <python script> [args]   (runs on the GPU box)
# one --pmc pass with --kernel-trace; prints per-kernel average duration and per-launch counter sums
TAG=$1; shift; CTRS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -- python3 $ROOT/"$@" > $OUT/run.log 2>&1
echo rc=$?
python3 - <<PY
import csv,glob,collections
dur=collections.defaultdict(list)
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:44]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:44]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k,v in agg.items():
        if "attn" in k or "gemm" in k or "ln_" in k or "colsum" in k:
            d=dur.get(k,[0]); nl=len(n[k])
            print(k, "launches", nl, "avg_us %.1f" % (sum(d)/max(len(d),1)/1e3), {a: round(b/nl/1e6,3) for a,b in sorted(v.items())})
PY
