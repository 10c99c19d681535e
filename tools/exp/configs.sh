#!/bin/bash
# The other configurations on the current build (DESIGN §5 "other configurations"): fp32, L127, C384 / C192, the emulated per-GPU shares, the
# body of step_dynamics.  One summary line each.
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/${FV3_CONFIGS_TAG:-r06_configs}
mkdir -p "$out"
cd "$R"
run() {
  name=$1; shift
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$out/$name.log" 2>&1
  echo "== $name ($*)"
  grep '^{' "$out/$name.log" | tail -1 | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
o = l['operators_ms_per_substep']
print(round(l['value'], 2), 'SDPD', round(l['acoustic_step_ms'], 3), 'ms/sub-step', l['finite'], {k: round(v, 2) for k, v in o.items() if v >= 1.0})
"
}
run fp32_l79 --precision 32
run fp32_l127 --precision 32 --nz 127
run fp64_l127 --nz 127
run c384 --config c384
run c192 --config c192
run share8 --emulate-share 8 --steps 3 --warmup 1
run share4 --emulate-share 4
run dynamics --tracers 4 --remap
