#!/bin/bash
# round-4 experiment X: the layer-mean pressure of the first 48 levels in the second accumulation-register bank (lib tag pm2) vs through memory
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4x
mkdir -p "$out"
cd "$R"
FV3_LIB_TAG=pm2 timeout 900 python3 -m pytest tests/test_parity.py tests/test_operator_parity.py -m gpu -x -q -k "riem or acoustic" > "$out/pytest.log" 2>&1; grep -E "passed|failed" "$out/pytest.log" | tail -2
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > "$out/bench_$name.log" 2>&1; echo "== $name"; tail -1 "$out/bench_$name.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
o = l['operators_ms_per_substep']
print(l['value'], l['acoustic_step_ms'], repr(l['state_checksum']['w']), {k: round(o[k], 2) for k in ('riem_solver_c', 'riem_solver3', 'nh_p_grad', 'c_sw', 'd_sw')})
"; }
run base X=1
run pm2 FV3_LIB_TAG=pm2
run baseb X=1
run pm2b FV3_LIB_TAG=pm2
