mkdir -p gpurun_out/r3i
python -m pytest tests/test_parity.py -q -m gpu -x -k "fused_nh_p_grad" 2>&1 | tail -2
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3i/bench.log 2>&1
tail -1 gpurun_out/r3i/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],2), round(d['acoustic_step_ms'],2), d['finite'], d['state_checksum']['w'], d['state_checksum']['u'], {k: round(v,2) for k,v in d['operators_ms_per_substep'].items()})"
python bench.py --emulate-share 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3i/share8.log 2>&1
tail -1 gpurun_out/r3i/share8.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('share 8', d['acoustic_step_ms'], d['finite'], {k: round(v,2) for k,v in d['operators_ms_per_substep'].items()})"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3i/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3i/stats.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_rocprof.py gpurun_out/r3i/stats/s_kernel_stats.csv 70 > gpurun_out/r3i/kernel_stats.md 2>&1
find gpurun_out/r3i -name "*kernel_trace.csv" -delete
grep -i "pgf\|nh_p_grad\|launch_frame\|fv3_k3n" gpurun_out/r3i/kernel_stats.md | head
