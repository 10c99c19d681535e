#!/bin/bash
# round-4 experiment W: levels in flight of the wave Riemann solvers' heavy sweeps (FV3_RIEM_U = 2 / 4 / 8) now that gam is in registers
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4w
mkdir -p "$out"
cd "$R"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > "$out/bench_$name.log" 2>&1; echo "== $name"; tail -1 "$out/bench_$name.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
o = l['operators_ms_per_substep']
print(l['value'], l['acoustic_step_ms'], repr(l['state_checksum']['w']), {k: round(o[k], 2) for k in ('riem_solver_c', 'riem_solver3', 'nh_p_grad', 'c_sw', 'd_sw')})
"; }
run u4 X=1
run u2 FV3_LIB_TAG=u2
run u8 FV3_LIB_TAG=u8
run u4b X=1
run u2b FV3_LIB_TAG=u2
run u8b FV3_LIB_TAG=u8
