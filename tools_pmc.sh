#!/bin/bash
# usage: bash tools_pmc.sh <tag> "<COUNTERS...>" <python script> [args]   (runs on the GPU box)
TAG=$1; shift; CTRS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 $ROOT/"$@" > $OUT/run.log 2>&1
echo rc=$?
python3 - <<PY
import csv,glob,collections
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:40]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    for k,v in agg.items():
        if "attn" in k or "gemm" in k: print(k, {a: round(b/1e6,2) for a,b in sorted(v.items())})
PY
