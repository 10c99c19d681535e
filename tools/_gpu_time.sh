mkdir -p gpurun_out/tm
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tm/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/tm/stats.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_rocprof.py gpurun_out/tm/stats/s_kernel_stats.csv 70 > gpurun_out/tm/kernel_stats.md
find gpurun_out/tm -name "*kernel_trace.csv" -delete
grep -i "$1" gpurun_out/tm/kernel_stats.md | head -${2:-12}
