#!/bin/bash
# round-4 experiment U: gam of the wave Riemann solvers in the accumulation registers through a computed branch (FV3_RIEM_REGS=0: scratch field)
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4u
mkdir -p "$out"
cd "$R"
timeout 900 python3 -m pytest tests/test_parity.py tests/test_operator_parity.py tests/test_gpu_invariants.py -m gpu -x -q -k "riem or acoustic or nh or invariant or toggle or form" > "$out/pytest.log" 2>&1; grep -E "passed|failed" "$out/pytest.log" | tail -2
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > "$out/bench_$name.log" 2>&1; echo "== $name"; tail -1 "$out/bench_$name.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
o = l['operators_ms_per_substep']
print(l['value'], l['acoustic_step_ms'], repr(l['state_checksum']['w']), repr(l['state_checksum']['delz']), {k: round(o[k], 2) for k in ('riem_solver_c', 'riem_solver3', 'c_sw', 'd_sw')})
"; }
run mem FV3_RIEM_REGS=0
run acc X=1
run memb FV3_RIEM_REGS=0
run accb X=1
