mkdir -p gpurun_out/r3j
timeout 600 python -m pytest tests/test_parity.py -q -m gpu -x -k "fused_nh_p_grad or del_n_chains" 2>&1 | grep -E "passed|failed|Error|fault" | tail -3
for v in "FV3_Q4_KB=0" "FV3_NH_PGF=staged" ""; do
timeout 600 env $v python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3j/bench_$v.log 2>&1
echo "[$v] rc=$?"; grep -E "fault|Error" gpurun_out/r3j/bench_$v.log | head -2
tail -1 gpurun_out/r3j/bench_$v.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print(round(d['value'],2), round(d['acoustic_step_ms'],2), d['finite'], d['state_checksum']['w'], d['state_checksum']['u'], {k: round(v,2) for k,v in d['operators_ms_per_substep'].items()})
except Exception as e: print('no json')"
done
