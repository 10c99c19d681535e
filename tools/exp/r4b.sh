#!/bin/bash
# round-4 experiment B: fast log / exp in the Riemann solvers vs libm (same box), in-kernel stamps of the marches
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4b
mkdir -p "$out"
cd "$R"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
$B > "$out/bench_fastmath.log" 2>&1
FV3_LIB_TAG=libm $B > "$out/bench_libm.log" 2>&1
FV3_LIB_TAG=stamps timeout 400 python3 tools/exp/stamps.py --config c384 --out "$out/stamps_c384.md" > "$out/stamps_c384.log" 2>&1
echo "stamps rc $?"
FV3_LIB_TAG=stamps timeout 400 python3 tools/exp/stamps.py --config c768 --out "$out/stamps_c768.md" > "$out/stamps_c768.log" 2>&1
echo "stamps rc $?"
timeout 900 python3 -m pytest tests/test_parity.py tests/test_operator_parity.py tests/test_driver.py tests/test_baseline_configs.py -m gpu -q -x 2>&1 | tail -15 > "$out/pytest.log"
for f in bench_fastmath bench_libm; do echo "== $f"; tail -1 "$out/$f.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
print(l['value'], l['acoustic_step_ms'], {k: round(v, 2) for k, v in l['operators_ms_per_substep'].items()})
"; done
tail -12 "$out/stamps_c384.log"
tail -12 "$out/stamps_c768.log"
tail -8 "$out/pytest.log"
