# round 3: profiles of the fused-del-n build + the emulated per-GPU shares
bash tools/collect_profiles.sh r03_fd --steps 3 --warmup 1
for n in 8 4 2; do
python bench.py --emulate-share $n --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_fd/share$n.log 2>&1
tail -1 gpurun_out/r03_fd/share$n.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('share $n', d['acoustic_step_ms'], d['finite'], d['halo_transport'], {k: round(v,2) for k,v in d['operators_ms_per_substep'].items()})"
done
cat gpurun_out/r03_fd/traffic.md | head -8
