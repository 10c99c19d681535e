#!/bin/bash
# timeline of one d_sw call (tools/dsw_timeline.py) for run-time variants:  bash tools/exp/dswtl.sh <tag> name1:ENV=V,ENV=V name2: ...
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
out=$R/gpurun_out/tl_$tag
mkdir -p "$out"
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  (
    for e in ${envs//,/ }; do export "$e"; done
    cd /tmp && export TMPDIR=/tmp
    timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$out/$name" -o t -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-op-timing ${BENCH_ARGS:-} > "$out/bench_$name.log" 2>&1
  )
  f=$(find "$out/$name" -name "t_kernel_trace.csv" | head -1)
  echo "== $name"
  python3 "$R/tools/dsw_timeline.py" "$f" ${TL_WHICH:-14} ${TL_MODE:-} ${TL_PER:-} | tee "$out/timeline_$name.md"
  find "$out/$name" -name "*kernel_trace.csv" -delete
done
