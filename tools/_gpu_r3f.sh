mkdir -p gpurun_out/r3f
python -m pytest tests/test_parity.py -q -m gpu -x -k "vorticity_del_n or height_del_n or del_n_chains" 2>&1 | tail -3
for v in "" "FV3_DZ_DELN=arrays" "FV3_DSW_VORT_DELN=arrays"; do
env $v python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3f/bench_$v.log 2>&1
tail -1 gpurun_out/r3f/bench_$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$v]', round(d['value'],2), round(d['acoustic_step_ms'],2), d['finite'], d['state_checksum']['w'], {k: round(v,2) for k,v in d['operators_ms_per_substep'].items()})"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3f/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3f/stats.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_rocprof.py gpurun_out/r3f/stats/s_kernel_stats.csv 60 > gpurun_out/r3f/kernel_stats.md 2>&1
find gpurun_out/r3f -name "*kernel_trace.csv" -delete
head -45 gpurun_out/r3f/kernel_stats.md
