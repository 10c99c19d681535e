# round 3: del-n chains inside the scalar marches -- parity on the GPU, A/B timing, kernel stats
mkdir -p gpurun_out/r3b
python -m pytest tests/test_parity.py -q -m gpu -x -k "del_n_chains or fused_scalar or native_and or c768 or forms" 2>&1 | tail -5 > gpurun_out/r3b/pytest_gpu.log
cat gpurun_out/r3b/pytest_gpu.log
for m in fused arrays; do
FV3_DSW_DELN=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3b/bench_$m.log 2>&1
tail -1 gpurun_out/r3b/bench_$m.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$m', d['value'], d['acoustic_step_ms'], d['state_checksum']['w'], d['state_checksum']['pt'], {k: round(v,2) for k,v in d['operators_ms_per_substep'].items()})"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3b/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3b/stats.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_rocprof.py gpurun_out/r3b/stats/s_kernel_stats.csv 60 > gpurun_out/r3b/kernel_stats.md 2>&1
find gpurun_out/r3b -name "*kernel_trace.csv" -delete
head -40 gpurun_out/r3b/kernel_stats.md
