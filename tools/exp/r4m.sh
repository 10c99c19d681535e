#!/bin/bash
# round-4 experiment M: gam (and, lib tag rp, PM) of the wave Riemann solvers in registers (FV3_RIEM_REGS=0: scratch fields)
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4m
mkdir -p "$out"
cd "$R"
timeout 900 python3 -m pytest tests/test_parity.py tests/test_operator_parity.py -m gpu -x -q -k "riem or acoustic or nh" > "$out/pytest.log" 2>&1; tail -3 "$out/pytest.log"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
FV3_RIEM_REGS=0 $B > "$out/bench_mem.log" 2>&1
$B > "$out/bench_regs.log" 2>&1
FV3_LIB_TAG=rp $B > "$out/bench_rp.log" 2>&1
FV3_RIEM_REGS=0 $B > "$out/bench_mem2.log" 2>&1
$B > "$out/bench_regs2.log" 2>&1
FV3_LIB_TAG=rp $B > "$out/bench_rp2.log" 2>&1
for f in bench_mem bench_regs bench_rp bench_mem2 bench_regs2 bench_rp2; do echo "== $f"; tail -1 "$out/$f.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
o = l['operators_ms_per_substep']
print(l['value'], l['acoustic_step_ms'], l['state_checksum']['w'], l['state_checksum']['delz'], {k: round(o[k], 2) for k in ('riem_solver_c', 'riem_solver3', 'd_sw', 'c_sw')})
"; done
