mkdir -p gpurun_out/r3l
timeout 600 python -m pytest tests/test_parity.py -q -m gpu -x -k "fused_nh_p_grad" 2>&1 | grep -E "passed|failed|Error|fault" | tail -3
for v in "" "FV3_NH_PGF=staged"; do
timeout 600 env $v python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3l/bench_$v.log 2>&1
tail -1 gpurun_out/r3l/bench_$v.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print('[$v]', round(d['value'],2), round(d['acoustic_step_ms'],2), d['finite'], d['state_checksum']['w'], d['state_checksum']['u'], {k: round(v,2) for k,v in d['operators_ms_per_substep'].items()})
except Exception as e: print('no json')"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3l/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3l/stats.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_rocprof.py gpurun_out/r3l/stats/s_kernel_stats.csv 70 > gpurun_out/r3l/kernel_stats.md 2>&1
find gpurun_out/r3l -name "*kernel_trace.csv" -delete
grep -i "pgf" gpurun_out/r3l/kernel_stats.md | head -12
