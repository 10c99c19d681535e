#!/bin/bash
# round-4 experiment A: baseline vs -ffp-contract=fast (same box), in-kernel stamps of the marches, parity of the fast build
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4a
mkdir -p "$out"
cd "$R"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
$B > "$out/bench_base.log" 2>&1
FV3_LIB_TAG=fast $B > "$out/bench_fast.log" 2>&1
$B > "$out/bench_base2.log" 2>&1
FV3_LIB_TAG=fast $B > "$out/bench_fast2.log" 2>&1
FV3_LIB_TAG=stamps timeout 300 python3 tools/exp/stamps.py --out "$out/stamps.md" > "$out/stamps.log" 2>&1
FV3_LIB_TAG=fast timeout 900 python3 -m pytest tests -m gpu -q --deselect tests/test_kernel_budgets.py 2>&1 | tail -40 > "$out/pytest_fast.log"
for f in bench_base bench_fast bench_base2 bench_fast2; do echo "== $f"; tail -1 "$out/$f.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
print(l['value'], l['acoustic_step_ms'], {k: round(v, 2) for k, v in l['operators_ms_per_substep'].items()})
"; done
cat "$out/stamps.md"
tail -15 "$out/pytest_fast.log"
