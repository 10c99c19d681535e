#!/bin/bash
# round-4 experiment V: full GPU suite on the build with the accumulation-register columns (fp64 + fp32) and the unscaled division in nh_p_grad
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4v
mkdir -p "$out"
cd "$R"
timeout 1500 python3 -m pytest tests -m gpu -x -q > "$out/pytest.log" 2>&1; grep -E "passed|failed|error" "$out/pytest.log" | tail -3
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > "$out/bench_$name.log" 2>&1; echo "== $name"; tail -1 "$out/bench_$name.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
o = l['operators_ms_per_substep']
print(l['value'], l['acoustic_step_ms'], repr(l['state_checksum']['u']), repr(l['state_checksum']['w']), {k: round(o[k], 2) for k in ('riem_solver_c', 'riem_solver3', 'nh_p_grad', 'c_sw', 'd_sw')})
"; }
run a X=1
run b X=1
