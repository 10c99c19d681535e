#!/bin/bash
# Run on the GPU box (through gpurun): GPU tests, the default bench line, a rocprofv3 kernel-trace
# summary of the same command, and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel
# trace only -- MI355X_MICROARCH.md, HBM section).  Everything lands in gpurun_out/$1/; the summaries
# are then copied into profiles/ by hand.
#   usage: tools/collect_profiles.sh <tag> [bench args...]
set -u
tag=${1:-final}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p "$out"
cd "$R"
python -m pytest tests -q -m gpu > "$out/pytest_gpu_full.log" 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" "$out/pytest_gpu_full.log" | tail -5 > "$out/pytest_gpu.log"
python bench.py "$@" > "$out/bench.log" 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o s -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$out/stats.log" 2>&1
# (the counter passes profile ONE sub-step, which is the first of its call: with FV3_ACC_STORE=0 it runs the form the other five sub-steps of a call
#  run -- the four flux accumulators read and written; the first sub-step of a call reads four fields, 10.4 GB, less)
#  FV3_GZ_FIRST=copy keeps the gz -> zh copy of the reference's order in the profiled sub-step: fv3_copy is the kernel the counters are calibrated on.)
export FV3_ACC_STORE=0 FV3_GZ_FIRST=copy
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -o p -- python3 "$R/bench.py" --steps 1 --warmup 0 --k-split 1 --n-split 1 --no-cpu-baseline --no-op-timing > "$out/pmc_$c.log" 2>&1
done
cd "$R"
export FV3_PMC_NOTE="counter pass with FV3_ACC_STORE=0: a regular (accumulating) sub-step; the first sub-step of a call reads four fields (10.4 GB) less"
python tools/summarize_rocprof.py "$out/stats/s_kernel_stats.csv" 60 > "$out/kernel_stats.md" 2>&1
cp "$out/stats/s_kernel_stats.csv" "$out/kernel_stats.csv" 2>/dev/null
python tools/pmc_traffic.py "$out/pmc_FETCH_SIZE/p_counter_collection.csv" "$out/pmc_WRITE_SIZE/p_counter_collection.csv" 70 2348252160 fxadv "fv3_d_sw_out#" "$out/traffic_d_sw.json" > "$out/traffic.md" 2>&1
unset FV3_ACC_STORE FV3_GZ_FIRST FV3_PMC_NOTE
find "$out" -name "*kernel_trace.csv" -delete
find "$out" -name "*counter_collection.csv" -delete
cat "$out/pytest_gpu.log"
tail -1 "$out/bench.log" | cut -c1-2500
