mkdir -p gpurun_out/r3d
python -m pytest tests/test_parity.py -q -m gpu -x -k "frame_first or native_and or del_n" 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3d/bench.log 2>&1
tail -1 gpurun_out/r3d/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['acoustic_step_ms'], d['finite'], d['state_checksum']['w'], {k: round(v,2) for k,v in d['operators_ms_per_substep'].items()})"
python tools/halo_overlap_experiment.py --share 8 --delays 0,250,500,1000 --json gpurun_out/r3d/overlap_share8.json 2>&1 | tee gpurun_out/r3d/overlap_share8.md
