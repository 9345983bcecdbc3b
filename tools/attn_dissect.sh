# Build timing-only variants of attention.hip (ATT_DBG bit masks, see attention.hip) into vita_amd/libvita_dbg<N>.so.
# usage (CPU box): bash tools/attn_dissect.sh build 1 2 4 8 16 ; (GPU box): bash tools/attn_dissect.sh run 0 1 2 4 8 16
R=${GRAFT_REPO_ROOT:-$PWD}
mode=$1; shift
if [ "$mode" = build ]; then
  for n in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DATT_DBG=$n -c $R/vita_amd/csrc/attention.hip -o /tmp/attention_dbg$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/attention_dbg$n.o $(ls $R/vita_amd/csrc/*.o | grep -v attention.o) -o $R/vita_amd/libvita_dbg$n.so || exit 1
  done
else
  for n in "$@"; do
    lib=$R/vita_amd/libvita_dbg$n.so; [ "$n" = 0 ] && lib=$R/vita_amd/libvita_hip.so
    echo "== ATT_DBG=$n"
    VITA_AMD_LIB=$lib timeout -k 10 120 python3 $R/tools/attn_bench.py attn 1 2>&1 | grep -v amdgpu.ids || exit 1
  done
fi
