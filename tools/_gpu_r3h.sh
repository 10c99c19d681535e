mkdir -p gpurun_out/r3h
python -m pytest tests/test_parity.py tests/test_operator_parity.py -q -m gpu -x -k "fused_nh_p_grad or nh_p_grad or frame_first or native_and or c768" 2>&1 | tail -3
for v in "" "FV3_NH_PGF=staged"; do
env $v python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3h/bench_$v.log 2>&1
tail -1 gpurun_out/r3h/bench_$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$v]', round(d['value'],2), round(d['acoustic_step_ms'],2), d['finite'], d['state_checksum']['w'], d['state_checksum']['u'], {k: round(v,2) for k,v in d['operators_ms_per_substep'].items()})"
done
python bench.py --emulate-share 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3h/share8.log 2>&1
tail -1 gpurun_out/r3h/share8.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('share 8', d['acoustic_step_ms'], d['finite'], {k: round(v,2) for k,v in d['operators_ms_per_substep'].items()})"
