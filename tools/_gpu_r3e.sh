mkdir -p gpurun_out/r3e
python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r3e/pytest_gpu.log; cat gpurun_out/r3e/pytest_gpu.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3e/bench.log 2>&1
tail -1 gpurun_out/r3e/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['acoustic_step_ms'], d['finite'], d['state_checksum']['w'], {k: round(v,2) for k,v in d['operators_ms_per_substep'].items()})"
python tools/halo_overlap_experiment.py --share 8 --delays 0,250,500,1000 --json gpurun_out/r3e/overlap_share8.json 2>&1 | tee gpurun_out/r3e/overlap_share8.md | tail -8
python tools/fp32_drift.py --json gpurun_out/r3e/fp32_drift.json 2>&1 | tee gpurun_out/r3e/fp32_drift.md | tail -8
