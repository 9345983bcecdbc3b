# usage (on the GPU box): bash tools/ab.sh <tools/attn_bench.py args...>  - same-box A/B of the in-tree library builds
R=${GRAFT_REPO_ROOT:-$PWD}
for lib in $R/vita_amd/libvita_base.so $R/vita_amd/libvita_hip.so $R/vita_amd/libvita_exp*.so; do
  [ -f $lib ] || continue
  echo "== $(basename $lib)"
  VITA_AMD_LIB=$lib timeout -k 10 200 python3 $R/tools/attn_bench.py "$@" 2>&1 | grep -v amdgpu.ids || exit 1
done
