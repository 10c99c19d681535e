#!/bin/bash
# round-4 experiment F: unconditional stores (sink) + optional inputs one step ahead; FA instantiations of tp2d_stream_t
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4f
mkdir -p "$out"
cd "$R"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
FV3_LIB_TAG=nou $B > "$out/bench_nou.log" 2>&1
$B > "$out/bench_ust.log" 2>&1
FV3_TP2D_FA=0 $B > "$out/bench_ust_nofa.log" 2>&1
FV3_LIB_TAG=nou $B > "$out/bench_nou2.log" 2>&1
$B > "$out/bench_ust2.log" 2>&1
for f in bench_nou bench_ust bench_ust_nofa bench_nou2 bench_ust2; do echo "== $f"; tail -1 "$out/$f.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
print(l['value'], l['acoustic_step_ms'], {k: round(v, 2) for k, v in l['operators_ms_per_substep'].items()})
"; done
timeout 900 python3 -m pytest tests/test_parity.py tests/test_operator_parity.py tests/test_baseline_configs.py -m gpu -q -x 2>&1 | tail -4
