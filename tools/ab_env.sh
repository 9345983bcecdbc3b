# usage (GPU box): bash tools/ab_env.sh "<ENV=VAL ...>" [bench args]  - bench.py with / without an environment setting, alternating, same box
R=${GRAFT_REPO_ROOT:-$PWD}
SET=$1; shift
for rep in 1 2 3; do
  for mode in off on; do
    if [ $mode = on ]; then E="$SET"; else E=""; fi
    L=$(env $E timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 3 "$@" 2>&1 | grep '"metric"') || exit 1
    echo "$mode [$E] $(echo "$L" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); k=d.get("kernels",{}); print(d["ms_per_step"], {n: v["ms_per_step"] for n, v in k.items()})')"
  done
done
