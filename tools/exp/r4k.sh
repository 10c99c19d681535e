#!/bin/bash
# round-4 experiment K: level-major launch geometry of the corner-KE and divergence-damping marches (FV3_KE_KB=0: plane-major)
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4k
mkdir -p "$out"
cd "$R"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
FV3_KE_KB=0 $B > "$out/bench_plane.log" 2>&1
$B > "$out/bench_level.log" 2>&1
FV3_KE_KB=0 $B > "$out/bench_plane2.log" 2>&1
$B > "$out/bench_level2.log" 2>&1
for f in bench_plane bench_level bench_plane2 bench_level2; do echo "== $f"; tail -1 "$out/$f.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
print(l['value'], l['acoustic_step_ms'], l['state_checksum']['u'], {k: round(v, 2) for k, v in l['operators_ms_per_substep'].items()})
"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o s -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$out/stats.log" 2>&1
cd "$R"; python3 tools/summarize_rocprof.py "$out/stats/s_kernel_stats.csv" 40 2>/dev/null | grep -E "ke_stream|divdamp_stream"
rm -rf "$out/stats"
