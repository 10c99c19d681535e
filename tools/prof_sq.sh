#!/bin/bash
# SQ-counter passes + kernel stats of one acoustic sub-step of the bench workload (run on the GPU box through gpurun).
#   usage: tools/prof_sq.sh <tag> [env assignments for the bench, e.g. FV3_DSW_SCALARS=separate]
set -u
tag=${1:-sq}
shift || true
for kv in "$@"; do export "$kv"; done
export FV3_ACC_STORE=0  # (the profiled sub-step is the first of its call: run the accumulating form the other five sub-steps run, as tools/collect_profiles.sh does)
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --k-split 1 --n-split 1 --no-cpu-baseline --no-op-timing"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o s -- $B > "$out/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d "$out/pmc_a" -o p -- $B > "$out/pmc_a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$out/pmc_b" -o p -- $B > "$out/pmc_b.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_c" -o p -- $B > "$out/pmc_c.log" 2>&1
cd "$R"
python tools/summarize_rocprof.py "$out/stats/s_kernel_stats.csv" 40 > "$out/kernel_stats.md" 2>&1
python tools/pmc_sq.py "$out/pmc_a/p_counter_collection.csv" "" 30 > "$out/sq_a.md" 2>&1
python tools/pmc_sq.py --valu-window "$out/pmc_a/p_counter_collection.csv" fxadv "${DSW_LAST:-fv3_d_sw_out#}" "$out/valu_d_sw.json" "$out/pmc_c/p_counter_collection.csv" > "$out/valu_d_sw.log" 2>&1
python tools/pmc_sq.py "$out/pmc_b/p_counter_collection.csv" "" 30 > "$out/sq_b.md" 2>&1
find "$out" -name "*kernel_trace.csv" -delete
find "$out" -name "*counter_collection.csv" -delete
head -30 "$out/kernel_stats.md"
head -14 "$out/sq_a.md"
head -14 "$out/sq_b.md"
