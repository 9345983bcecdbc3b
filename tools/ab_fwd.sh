# usage (GPU box): bash tools/ab_fwd.sh [fwd|bwd|both]  - attention timing of every in-tree library build, same box
R=${GRAFT_REPO_ROOT:-$PWD}
for lib in $R/vita_amd/libvita_base.so $R/vita_amd/libvita_hip.so $R/vita_amd/libvita_exp*.so; do
  [ -f $lib ] || continue
  echo "== $(basename $lib)"
  VITA_AMD_LIB=$lib timeout -k 10 200 python3 $R/tools/attn_fwd_time.py "$@" 2>&1 | grep -v amdgpu.ids || exit 1
done
