#!/bin/bash
# round-4 experiment E: rare-path loads waited for inside their own branch (FV3_LANDED) vs the build without (-DFV3_NO_LANDED), same box
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4e
mkdir -p "$out"
cd "$R"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
FV3_LIB_TAG=nol $B > "$out/bench_nol.log" 2>&1
$B > "$out/bench_landed.log" 2>&1
FV3_LIB_TAG=nol $B > "$out/bench_nol2.log" 2>&1
$B > "$out/bench_landed2.log" 2>&1
for f in bench_nol bench_landed bench_nol2 bench_landed2; do echo "== $f"; tail -1 "$out/$f.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
print(l['value'], l['acoustic_step_ms'], {k: round(v, 2) for k, v in l['operators_ms_per_substep'].items()})
"; done
timeout 600 python3 -m pytest tests/test_parity.py -m gpu -q -x 2>&1 | tail -4
