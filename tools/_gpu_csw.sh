mkdir -p gpurun_out/csw1
timeout 600 python -m pytest tests/test_parity.py tests/test_baseline_configs.py -m gpu -x -q -k "c_sw or c768_layout" > gpurun_out/csw1/pytest.log 2>&1; tail -3 gpurun_out/csw1/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/csw1/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/csw1/stats.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_rocprof.py gpurun_out/csw1/stats/s_kernel_stats.csv 60 > gpurun_out/csw1/kernel_stats.md
find gpurun_out/csw1 -name "*kernel_trace.csv" -delete
grep -i "c_sw\|csw" gpurun_out/csw1/kernel_stats.md | head -12
tail -1 gpurun_out/csw1/stats.log | cut -c1-300
