mkdir -p gpurun_out/r3k
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3k/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3k/stats.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_rocprof.py gpurun_out/r3k/stats/s_kernel_stats.csv 70 > gpurun_out/r3k/kernel_stats.md 2>&1
find gpurun_out/r3k -name "*kernel_trace.csv" -delete
grep -i "pgf\|dsw_scalars" gpurun_out/r3k/kernel_stats.md | head -12
