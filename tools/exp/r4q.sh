#!/bin/bash
# round-4 experiment Q: scalar-base addressing (32-bit level-carrying offsets) in the wave Riemann solvers (lib tag a64 = -DFV3_RIEM_ADDR64: 64-bit vector addresses)
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4q
mkdir -p "$out"
cd "$R"
timeout 900 python3 -m pytest tests/test_parity.py tests/test_operator_parity.py tests/test_gpu_invariants.py -m gpu -x -q -k "riem or acoustic or nh or invariant or toggle or form" > "$out/pytest.log" 2>&1; tail -3 "$out/pytest.log"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > "$out/bench_$name.log" 2>&1; echo "== $name"; tail -1 "$out/bench_$name.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
o = l['operators_ms_per_substep']
print(l['value'], l['acoustic_step_ms'], l['state_checksum']['w'], {k: round(o[k], 2) for k in ('riem_solver_c', 'riem_solver3', 'c_sw', 'd_sw')})
"; }
run a64 FV3_LIB_TAG=a64
run s32 X=1
run a642 FV3_LIB_TAG=a64
run s32b X=1
