# usage (GPU box): bash tools/ab_gemm.sh [M]  - dispatcher-chosen GEMM timings of every in-tree library build, same box
R=${GRAFT_REPO_ROOT:-$PWD}
for lib in $R/vita_amd/libvita_base.so $R/vita_amd/libvita_hip.so $R/vita_amd/libvita_exp*.so; do
  [ -f $lib ] || continue
  echo "== $(basename $lib)"
  VITA_AMD_LIB=$lib GEMM_BENCH_V=${GEMM_BENCH_V:-0} timeout -k 10 200 python3 $R/tools/gemm_bench.py "$@" 2>&1 | grep -v amdgpu.ids || exit 1
done
