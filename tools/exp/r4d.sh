#!/bin/bash
# round-4 experiment D: PPM order as a compile-time constant in the two-tracer marches (FV3_HORD_CONST=0: run-time form), same box
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4d
mkdir -p "$out"
cd "$R"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
FV3_HORD_CONST=0 $B > "$out/bench_hc0.log" 2>&1
$B > "$out/bench_hc6.log" 2>&1
FV3_HORD_CONST=0 $B > "$out/bench_hc0b.log" 2>&1
$B > "$out/bench_hc6b.log" 2>&1
for f in bench_hc0 bench_hc6 bench_hc0b bench_hc6b; do echo "== $f"; tail -1 "$out/$f.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
print(l['value'], l['acoustic_step_ms'], {k: round(v, 2) for k, v in l['operators_ms_per_substep'].items()})
"; done
