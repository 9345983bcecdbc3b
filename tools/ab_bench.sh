# usage (GPU box): bash tools/ab_bench.sh "<ENV=.. ENV=..>" "<ENV=..>" ...   - one bench.py run per environment setting, same box
R=${GRAFT_REPO_ROOT:-$PWD}
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 280 python3 $R/bench.py --no-cpu-baseline --steps ${STEPS:-10} --warmup 3 ${BENCH_ARGS:-} 2>&1 | grep -v amdgpu.ids | \
    python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d.get('kernels', {})
        print('ms/step', d['ms_per_step'], 'value', d['value'], ' | '.join(f\"{n.replace('gemm_kernel','g').replace('_kernels','').replace('_kernel','')} {v['ms_per_step']}\" for n, v in k.items()))
    else:
        print(l.rstrip())
" || exit 1
done
