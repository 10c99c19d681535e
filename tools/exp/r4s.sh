#!/bin/bash
# round-4 experiment S: scalar-base addressing (FV3_EL) in the fused nh_p_grad march (lib tag pgf0 = plain element accesses there)
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4s
mkdir -p "$out"
cd "$R"
timeout 900 python3 -m pytest tests/test_parity.py tests/test_operator_parity.py -m gpu -x -q -k "nh_p_grad or acoustic" > "$out/pytest.log" 2>&1; grep -E "passed|failed" "$out/pytest.log" | tail -2
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > "$out/bench_$name.log" 2>&1; echo "== $name"; tail -1 "$out/bench_$name.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
o = l['operators_ms_per_substep']
print(l['value'], l['acoustic_step_ms'], l['state_checksum']['u'], {k: round(o[k], 2) for k in ('c_sw', 'd_sw', 'nh_p_grad', 'riem_solver3')})
"; }
run pgf0 FV3_LIB_TAG=pgf0
run el X=1
run pgf0b FV3_LIB_TAG=pgf0
run elb X=1
