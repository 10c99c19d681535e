#!/bin/bash
# round-4: GPU suite + bench + kernel stats + PMC traffic + SQ counters of the current build
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
bash tools/collect_profiles.sh r04_final --steps 5 --warmup 2
bash tools/prof_sq.sh r04_sq
cat gpurun_out/r04_sq/valu_d_sw.log
