#!/bin/bash
# round-4 experiment L: the q_con + pt march recomputes the new air mass (vs -DFV3_NO_DN_RECOMP: loads it);
# FV3_DSW_EDGE_OVERLAP=1: the transposed tile-edge marches beside the interior ones
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4l
mkdir -p "$out"
cd "$R"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
FV3_LIB_TAG=nodn $B > "$out/bench_nodn.log" 2>&1
$B > "$out/bench_dn.log" 2>&1
FV3_DSW_EDGE_OVERLAP=1 $B > "$out/bench_edge.log" 2>&1
FV3_LIB_TAG=nodn $B > "$out/bench_nodn2.log" 2>&1
$B > "$out/bench_dn2.log" 2>&1
FV3_DSW_EDGE_OVERLAP=1 $B > "$out/bench_edge2.log" 2>&1
for f in bench_nodn bench_dn bench_edge bench_nodn2 bench_dn2 bench_edge2; do echo "== $f"; tail -1 "$out/$f.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
print(l['value'], l['acoustic_step_ms'], l['state_checksum']['u'], l['state_checksum']['pt'], {k: round(v, 2) for k, v in l['operators_ms_per_substep'].items()})
"; done
timeout 900 python3 -m pytest tests/test_parity.py tests/test_tracer_advection.py -m gpu -x -q > "$out/pytest.log" 2>&1; tail -3 "$out/pytest.log"
FV3_DSW_EDGE_OVERLAP=1 timeout 900 python3 -m pytest tests/test_parity.py -m gpu -x -q -k "d_sw or acoustic" > "$out/pytest_edge.log" 2>&1; tail -3 "$out/pytest_edge.log"
