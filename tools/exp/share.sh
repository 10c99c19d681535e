#!/bin/bash
# The emulated per-GPU shares (and two small full configurations) launched kernel by kernel and replayed from a HIP graph (bench.py --graph), same box.
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r06_share
mkdir -p "$out"
cd "$R"
run() {
  name=$1; shift
  python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-op-timing "$@" > "$out/$name.log" 2>&1
  echo "== $name ($*)"
  grep '^{' "$out/$name.log" | tail -1 | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
c = l['state_checksum']
print(round(l['value'], 2), 'SDPD', round(l['acoustic_step_ms'], 3), 'ms/sub-step', l['finite'], l.get('graph_replay'), repr(c['u']), repr(c['w']))
" || tail -5 "$out/$name.log"
}
run whole
run share8 --emulate-share 8
run share8_graph --emulate-share 8 --graph
run share4 --emulate-share 4
run share4_graph --emulate-share 4 --graph
run share2 --emulate-share 2
run share2_graph --emulate-share 2 --graph
run c192 --config c192
run c192_graph --config c192 --graph
