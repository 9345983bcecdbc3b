#!/bin/bash
# usage: bash tools/build_variant.sh <name> <file.hip> "<-DFLAGS>"   -> vita_amd/libvita_<name>.so = the in-tree library with
# one source rebuilt under extra flags (same-box A/B of kernel variants with VITA_AMD_LIB, see tools/ab.sh)
set -e
cd "$(dirname "$0")/../vita_amd/csrc"
make -s
NAME=$1; SRC=$2; FLAGS=$3
case $SRC in attention*) FLAGS="-fno-honor-nans $FLAGS";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $FLAGS -c $SRC -o /tmp/variant_$NAME.o
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/variant_$NAME.o -o ../libvita_$NAME.so
echo built vita_amd/libvita_$NAME.so
