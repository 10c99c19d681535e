# round 3, first GPU call: GPU tests, bench with and without the scalar ping-pong
mkdir -p gpurun_out/r3a
python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r3a/pytest_gpu.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3a/bench_pp.log 2>&1
FV3_PINGPONG=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3a/bench_nopp.log 2>&1
cat gpurun_out/r3a/pytest_gpu.log
for f in gpurun_out/r3a/bench_pp.log gpurun_out/r3a/bench_nopp.log; do tail -1 $f | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['acoustic_step_ms'], d['state_checksum'], {k: round(v,2) for k,v in d['operators_ms_per_substep'].items()})"; done
