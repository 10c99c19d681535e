#!/bin/bash
# Instruction mix of the kernels of one acoustic sub-step (run on the GPU box through gpurun): how many of a wave's VALU
# instructions are fp64 arithmetic (add / mul / fma / transcendental) and how many are moves, selects, lane reads and integer
# address arithmetic -- the part a marching kernel can lose without changing its results.
#   usage: tools/prof_mix.sh <tag> [env assignments for the bench]
set -u
tag=${1:-mix}
shift || true
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --k-split 1 --n-split 1 --no-cpu-baseline --no-op-timing"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d "$out/pmc_a" -o p -- $B > "$out/pmc_a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d "$out/pmc_b" -o p -- $B > "$out/pmc_b.log" 2>&1
cd "$R"
python tools/pmc_sq.py "$out/pmc_a/p_counter_collection.csv" "" 30 > "$out/mix_a.md" 2>&1
python tools/pmc_sq.py "$out/pmc_b/p_counter_collection.csv" "" 30 > "$out/mix_b.md" 2>&1
find "$out" -name "*kernel_trace.csv" -delete
find "$out" -name "*counter_collection.csv" -delete
grep -v "at::native\|rocclr\|rocblas" "$out/mix_a.md" | head -24
grep -v "at::native\|rocclr\|rocblas" "$out/mix_b.md" | head -24
tail -3 "$out/pmc_a.log"
