#!/bin/bash
# The two PMC traffic passes of tools/collect_profiles.sh alone (FETCH_SIZE / WRITE_SIZE of one regular sub-step -> traffic.md + traffic_d_sw.json in gpurun_out/$1).
set -u
ulimit -c 0
tag=${1:-r05_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export FV3_ACC_STORE=0 FV3_GZ_FIRST=copy
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -o p -- python3 "$R/bench.py" --steps 1 --warmup 0 --k-split 1 --n-split 1 --no-cpu-baseline --no-op-timing > "$out/pmc_$c.log" 2>&1
done
cd "$R"
export FV3_PMC_NOTE="counter pass with FV3_ACC_STORE=0: a regular (accumulating) sub-step; the first sub-step of a call reads four fields (10.4 GB) less"
python3 tools/pmc_traffic.py "$out/pmc_FETCH_SIZE/p_counter_collection.csv" "$out/pmc_WRITE_SIZE/p_counter_collection.csv" 70 2348252160 fxadv "fv3_d_sw_out#" "$out/traffic_d_sw.json" > "$out/traffic.md" 2>&1
find "$out" -name "*kernel_trace.csv" -delete
find "$out" -name "*counter_collection.csv" -delete
head -4 "$out/traffic.md"
