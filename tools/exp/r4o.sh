#!/bin/bash
# round-4 experiment O: c_sw interior march with 8 waves (levels) per workgroup sharing the metric rows (lib tag nw8) vs 4
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4o
mkdir -p "$out"
cd "$R"
FV3_LIB_TAG=nw8 timeout 900 python3 -m pytest tests/test_parity.py -m gpu -x -q -k "c_sw" > "$out/pytest.log" 2>&1; tail -3 "$out/pytest.log"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > "$out/bench_$name.log" 2>&1; echo "== $name"; tail -1 "$out/bench_$name.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
o = l['operators_ms_per_substep']
print(l['value'], l['acoustic_step_ms'], l['state_checksum']['u'], {k: round(o[k], 2) for k in ('c_sw', 'd_sw', 'update_dz_d', 'nh_p_grad')})
"; }
run nw4 X=1
run nw8 FV3_LIB_TAG=nw8
run nw4b X=1
run nw8b FV3_LIB_TAG=nw8
run nw4c X=1
run nw8c FV3_LIB_TAG=nw8
