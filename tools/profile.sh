#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + HBM traffic PMC passes of bench.py.
# usage: bash tools/profile.sh <tag> [bench args...]
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --no-cpu-baseline --no-kernel-timing "$@" > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --no-cpu-baseline --no-kernel-timing "$@" > $OUT/pmc_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $ROOT/bench.py --no-cpu-baseline --no-kernel-timing "$@" > $OUT/pmc_mfma.log 2>&1
tail -1 $OUT/trace.log | cut -c1-300
ls $OUT/*/*
