#!/bin/bash
# Per-kernel times of build / run-time variants on the GPU box (through gpurun): optional GPU parity tests of the default build first, then one
# `rocprofv3 --kernel-trace --stats` run of a short bench per variant; prints the average duration of the kernels whose (demangled) name
# matches <pattern> and the bench line's operator times / state checksums.
#   usage: bash tools/exp/kstat.sh <tag> "<pytest -k expression or empty>" "<egrep pattern>" name1:ENV=V,ENV=V name2: ...
#   e.g.:  gpurun -- 'bash tools/exp/kstat.sh abl "pair_march" "pair_march|dsw_scalars" new: old:FV3_DSW_MARCH=old abl1:FV3_LIB_TAG=abl1'
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; kexpr=$2; pat=$3; shift 3
out=$R/gpurun_out/ks_$tag
mkdir -p "$out"
cd "$R"
if [ -n "$kexpr" ]; then
  timeout 1500 python3 -m pytest tests -m gpu -x -q -k "$kexpr" > "$out/pytest.log" 2>&1
  grep -E "passed|failed|error" "$out/pytest.log" | tail -3
fi
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  (
    for e in ${envs//,/ }; do export "$e"; done
    cd /tmp && export TMPDIR=/tmp
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/st_$name" -o s -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench_$name.log" 2>&1
  )
  echo "== $name"
  f=$(find "$out/st_$name" -name "s_kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then
    python3 tools/summarize_rocprof.py "$f" 70 > "$out/kernel_stats_$name.md" 2>&1
    grep -E "$pat" "$out/kernel_stats_$name.md" | head -24
  fi
  grep '^{' "$out/bench_$name.log" | tail -1 | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
o = l['operators_ms_per_substep']
c = l['state_checksum']
print(round(l['value'], 2), round(l['acoustic_step_ms'], 3), repr(c['u']), repr(c['w']), repr(c['delz']), {k: round(v, 2) for k, v in o.items() if v >= 2.5})
"
  find "$out" -name "*kernel_trace.csv" -delete
done
