#!/bin/bash
# rounds 5 - 6: GPU suite + bench + kernel stats + PMC traffic + SQ counters (+ shader clock) of the current build
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
bash tools/collect_profiles.sh ${FV3_ROUND_TAG:-r06_final} --steps 5 --warmup 2
bash tools/prof_sq.sh ${FV3_ROUND_TAG_SQ:-r06_sq}
python3 -m pytest tests/test_baseline_configs.py -m gpu -q -s -k "fp32_build" 2>&1 | grep -E "fp32 vs|passed|failed" > gpurun_out/${FV3_ROUND_TAG:-r06_final}/fp32_errors.log
cat gpurun_out/${FV3_ROUND_TAG:-r06_final}/fp32_errors.log
cat gpurun_out/${FV3_ROUND_TAG_SQ:-r06_sq}/valu_d_sw.log
