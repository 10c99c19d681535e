#!/bin/bash
# round-4 experiment G: FA instantiations of tp2d_stream_t as rolled loops (no spills) vs FV3_TP2D_FA=0, same box
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4g
mkdir -p "$out"
cd "$R"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
FV3_TP2D_FA=0 $B > "$out/bench_nofa.log" 2>&1
$B > "$out/bench_fa.log" 2>&1
FV3_TP2D_FA=0 $B > "$out/bench_nofa2.log" 2>&1
$B > "$out/bench_fa2.log" 2>&1
for f in bench_nofa bench_fa bench_nofa2 bench_fa2; do echo "== $f"; tail -1 "$out/$f.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
print(l['value'], l['acoustic_step_ms'], {k: round(v, 2) for k, v in l['operators_ms_per_substep'].items()})
"; done
timeout 900 python3 -m pytest tests/test_parity.py tests/test_operator_parity.py tests/test_baseline_configs.py -m gpu -q -x 2>&1 | tail -4
