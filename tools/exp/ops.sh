# Per-operator times (HIP-event pairs of the bench line) of a few run-time variants, one line each: the whole problem, the one-launch edge_profile switch, the emulated 1/8 share and its segment lengths.
cd ${GRAFT_REPO_ROOT:-.}
run() { name=$1; shift; env "$@" > gpurun_out/ops_$name.log 2>&1; grep '^{' gpurun_out/ops_$name.log | tail -1 | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
print('$name', round(l['acoustic_step_ms'], 3), l['finite'], {k: round(v, 3) for k, v in l['operators_ms_per_substep'].items()})
"; }
run whole X=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
run whole_ep1 FV3_EP_ONE_LAUNCH=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
run share8 X=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --emulate-share 8
run share8_seg96 FV3_SEG=96 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --emulate-share 8
run share8_seg48 FV3_SEG=48 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --emulate-share 8
run whole_b X=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
run whole_ep1_b FV3_EP_ONE_LAUNCH=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
