#!/bin/bash
# A/B of run-time variants on the emulated 1/8 share and the whole problem:  bash tools/exp/share_ab.sh <tag> name1:ENV=V,ENV=V name2: ...
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
out=$R/gpurun_out/sab_$tag
mkdir -p "$out"
cd "$R"
for rep in ${AB_REPS:-a b}; do
  for v in "$@"; do
    name=${v%%:*}; envs=${v#*:}
    for cfg in "share8:--emulate-share 8" "whole:"; do
      cn=${cfg%%:*}; args=${cfg#*:}
      env ${envs//,/ } X_=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-op-timing $args > "$out/${cn}_${name}_$rep.log" 2>&1
      echo "== $name $cn ($rep) $(grep '^{' "$out/${cn}_${name}_$rep.log" | tail -1 | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
c = l['state_checksum']
print(round(l['value'], 2), round(l['acoustic_step_ms'], 3), l['finite'], repr(c['u']), repr(c['w']))
")"
    done
  done
done
