set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab1 -- python3 $R/tools_attn_bench.py attn 1 > $R/gpurun_out/ab1.log 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab0 -- python3 $R/tools_attn_bench.py attn 0 > $R/gpurun_out/ab0.log 2>&1
timeout -k 10 120 python3 $R/tools_attn_bench.py ln > $R/gpurun_out/ln.log 2>&1
cat $R/gpurun_out/ab1.log $R/gpurun_out/ab0.log $R/gpurun_out/ln.log | grep -v amdgpu.ids
for d in ab1 ab0; do head -6 $R/gpurun_out/$d/*/*kernel_stats.csv | cut -c1-60,100-220; done
