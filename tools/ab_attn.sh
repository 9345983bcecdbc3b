# usage (GPU box): bash tools/ab_attn.sh  - attention fwd/bwd timings (hand-off path) of every in-tree library build, same box
R=${GRAFT_REPO_ROOT:-$PWD}
for lib in $R/vita_amd/libvita_hip.so $R/vita_amd/libvita_exp*.so $R/vita_amd/libvita_hip.so; do
  [ -f $lib ] || continue
  echo "== $(basename $lib)"
  VITA_AMD_LIB=$lib timeout -k 10 200 python3 $R/tools/attn_bench.py attn 1 2>&1 | grep -v amdgpu.ids || exit 1
done
