mkdir -p gpurun_out/r3m
timeout 900 python -m pytest tests/test_operator_parity.py tests/test_parity.py -q -m gpu -x -k "riem or native_and or c768" 2>&1 | grep -E "passed|failed|Error|fault" | tail -3
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3m/bench.log 2>&1
tail -1 gpurun_out/r3m/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],2), round(d['acoustic_step_ms'],2), d['finite'], d['state_checksum']['w'], d['state_checksum']['u'], {k: round(v,2) for k,v in d['operators_ms_per_substep'].items()})"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3m/bench20.log 2>&1
tail -1 gpurun_out/r3m/bench20.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('20 steps:', round(d['value'],2), round(d['acoustic_step_ms'],2), d['finite'], {k: round(v,2) for k,v in d['operators_ms_per_substep'].items()})"
