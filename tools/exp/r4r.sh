#!/bin/bash
# round-4 experiment R: scalar-base addressing (FV3_EL) in the transport marches: default = tp4 + tp2d, elA = tp4 only, el0 = neither
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4r
mkdir -p "$out"
cd "$R"
timeout 900 python3 -m pytest tests/test_parity.py tests/test_gpu_invariants.py -m gpu -x -q -k "d_sw or update_dz_d or fxadv or acoustic or toggle or form or invariant" > "$out/pytest.log" 2>&1; grep -E "passed|failed" "$out/pytest.log" | tail -2
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > "$out/bench_$name.log" 2>&1; echo "== $name"; tail -1 "$out/bench_$name.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
o = l['operators_ms_per_substep']
print(l['value'], l['acoustic_step_ms'], l['state_checksum']['u'], {k: round(o[k], 2) for k in ('c_sw', 'd_sw', 'update_dz_d', 'riem_solver3')})
"; }
run el0 FV3_LIB_TAG=el0
run elA FV3_LIB_TAG=elA
run elAB X=1
run el0b FV3_LIB_TAG=el0
run elAb FV3_LIB_TAG=elA
run elABb X=1
