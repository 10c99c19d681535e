#!/bin/bash
# round-4 experiment J: the wind branch of d_sw on the auxiliary stream beside the scalar marches (FV3_DSW_WIND_OVERLAP=0: program order)
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4j
mkdir -p "$out"
cd "$R"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
FV3_DSW_WIND_OVERLAP=0 $B > "$out/bench_serial.log" 2>&1
$B > "$out/bench_overlap.log" 2>&1
FV3_DSW_WIND_OVERLAP=0 $B > "$out/bench_serial2.log" 2>&1
$B > "$out/bench_overlap2.log" 2>&1
for f in bench_serial bench_overlap bench_serial2 bench_overlap2; do echo "== $f"; tail -1 "$out/$f.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
print(l['value'], l['acoustic_step_ms'], l['state_checksum']['u'], {k: round(v, 2) for k, v in l['operators_ms_per_substep'].items()})
"; done
timeout 1200 python3 -m pytest tests/test_parity.py tests/test_gpu_invariants.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error" | tail -5
