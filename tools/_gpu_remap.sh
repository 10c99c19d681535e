mkdir -p gpurun_out/rm
timeout 600 python -m pytest tests/test_remap.py tests/test_tracer_advection.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rm/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-op-timing --tracers 4 --remap > $GRAFT_REPO_ROOT/gpurun_out/rm/stats.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_rocprof.py gpurun_out/rm/stats/s_kernel_stats.csv 80 > gpurun_out/rm/kernel_stats_full_dynamics.md
find gpurun_out/rm -name "*kernel_trace.csv" -delete
grep -i "remap\|tracer\|dsw_scalars_t<2, ., true" gpurun_out/rm/kernel_stats_full_dynamics.md | head -20
tail -1 gpurun_out/rm/stats.log | cut -c1-200
