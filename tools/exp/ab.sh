#!/bin/bash
# Same-box A/B of build / run-time variants on the GPU box (through gpurun): optional GPU parity tests of ONE variant first, then every
# variant benched twice in alternation, one summary line each (whole-job value, ms per acoustic sub-step, state checksums, operator times).
#   usage: bash tools/exp/ab.sh <tag> "<pytest -k expression or empty>" "<variant for the tests: ENV=V ... or empty>" name1:ENV=V,ENV=V name2: ...
#   e.g.:  FV3_LIB_TAG=nofma FV3_EXTRA_FLAGS=-DFV3_MATH_NO_FMA python -m pace_amd.build      (here: builds libfv3_mi355x_f64.nofma.so)
#          gpurun -- 'bash tools/exp/ab.sh p "riem or acoustic" "" nofma:FV3_LIB_TAG=nofma fma:'
# The variants of round 4 and what they measured: tools/exp/EXPERIMENTS.md.
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; kexpr=$2; tenv=$3; shift 3
out=$R/gpurun_out/ab_$tag
mkdir -p "$out"
cd "$R"
if [ -n "$kexpr" ]; then
  env $tenv X_=1 timeout 1500 python3 -m pytest tests -m gpu -x -q -k "$kexpr" > "$out/pytest.log" 2>&1
  grep -E "passed|failed|error" "$out/pytest.log" | tail -2
fi
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
for rep in ${AB_REPS:-a b}; do
  for v in "$@"; do
    name=${v%%:*}; envs=${v#*:}
    env ${envs//,/ } X_=1 $B > "$out/bench_${name}_$rep.log" 2>&1
    echo "== $name ($rep)"
    tail -1 "$out/bench_${name}_$rep.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
o = l['operators_ms_per_substep']
c = l['state_checksum']
print(round(l['value'], 2), round(l['acoustic_step_ms'], 3), repr(c['u']), repr(c['w']), repr(c['delz']), {k: round(v, 2) for k, v in o.items() if v >= 2.5})
"
  done
done
