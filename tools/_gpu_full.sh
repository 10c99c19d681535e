mkdir -p gpurun_out/full
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/full/pytest.log 2>&1; tail -3 gpurun_out/full/pytest.log
timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/full/bench.log 2>&1; tail -1 gpurun_out/full/bench.log | cut -c1-2200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/full/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-op-timing --tracers 4 --remap > $GRAFT_REPO_ROOT/gpurun_out/full/stats.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_rocprof.py gpurun_out/full/stats/s_kernel_stats.csv 80 > gpurun_out/full/kernel_stats_full_dynamics.md
find gpurun_out/full -name "*kernel_trace.csv" -delete
grep -i "remap\|tracer\|dsw_scalars_t<2, ., true\|fv3_k3<4" gpurun_out/full/kernel_stats_full_dynamics.md | head -30
