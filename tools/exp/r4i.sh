#!/bin/bash
# round-4 experiment I: in-kernel stamps of the four marches on the current build; GPU suite
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4i
mkdir -p "$out"
cd "$R"
FV3_LIB_TAG=stamps timeout 600 python3 tools/exp/stamps.py --config c768 --out "$out/stamps_c768.md" > "$out/stamps_c768.log" 2>&1
echo "stamps rc $?"; tail -3 "$out/stamps_c768.log"
cat "$out/stamps_c768.md"
timeout 1500 python3 -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -5
