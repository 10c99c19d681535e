#!/bin/bash
set -u
tag=${1:-cswprof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --k-split 1 --n-split 1 --no-cpu-baseline --no-op-timing"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d "$out/pmc_a" -o p -- $B > "$out/pmc_a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$out/pmc_b" -o p -- $B > "$out/pmc_b.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -o p -- $B > "$out/pmc_$c.log" 2>&1
done
cd "$R"
python tools/pmc_sq.py "$out/pmc_a/p_counter_collection.csv" "" 60 > "$out/sq_a.md" 2>&1
python tools/pmc_sq.py "$out/pmc_b/p_counter_collection.csv" "" 60 > "$out/sq_b.md" 2>&1
python tools/pmc_traffic.py "$out/pmc_FETCH_SIZE/p_counter_collection.csv" "$out/pmc_WRITE_SIZE/p_counter_collection.csv" 70 2348252160 fxadv "fv3_d_sw#" "$out/traffic_d_sw.json" > "$out/traffic.md" 2>&1
find "$out" -name "*kernel_trace.csv" -delete
find "$out" -name "*counter_collection.csv" -delete
head -4 "$out/sq_a.md"; grep -i "csw\|c_sw#1[23]" "$out/sq_a.md"
head -4 "$out/sq_b.md"; grep -i "csw\|c_sw#1[23]" "$out/sq_b.md"
grep -i "csw\|c_sw#1[23]\|^| kernel" "$out/traffic.md"
