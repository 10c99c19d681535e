#!/bin/bash
# round-4 experiment H: -ffp-contract=fast after the wait-placement fixes (same box A/B)
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4h
mkdir -p "$out"
cd "$R"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
$B > "$out/bench_off.log" 2>&1
FV3_LIB_TAG=fast $B > "$out/bench_fast.log" 2>&1
$B > "$out/bench_off2.log" 2>&1
FV3_LIB_TAG=fast $B > "$out/bench_fast2.log" 2>&1
for f in bench_off bench_fast bench_off2 bench_fast2; do echo "== $f"; tail -1 "$out/$f.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
print(l['value'], l['acoustic_step_ms'], {k: round(v, 2) for k, v in l['operators_ms_per_substep'].items()})
"; done
