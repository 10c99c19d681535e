#!/bin/bash
# round-4: the other configurations on the final build
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4y
mkdir -p "$out"
cd "$R"
run() { name=$1; shift; timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$out/bench_$name.log" 2>&1; echo "== $name"; tail -1 "$out/bench_$name.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
o = l['operators_ms_per_substep']
print(l['value'], l['ms_per_step'], l['acoustic_step_ms'], l['config']['workload'][:60], {k: round(o[k], 2) for k in ('riem_solver_c', 'riem_solver3', 'nh_p_grad', 'c_sw', 'd_sw', 'update_dz_d')})
"; }
run fp32 --precision 32
run l127_fp32 --precision 32 --nz 127
run l127_fp64 --nz 127
run c384 --config c384
run c192 --config c192
run share8 --emulate-share 8
run share4 --emulate-share 4
run tracers_remap --tracers 4 --remap
