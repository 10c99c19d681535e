#!/bin/bash
# round-4 experiment C: in-kernel stamps of the marches (bench workload), then the GPU suite on the current build
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4c
mkdir -p "$out"
cd "$R"
FV3_LIB_TAG=stamps timeout 400 python3 tools/exp/stamps.py --config c768 --out "$out/stamps_c768.md" > "$out/stamps_c768.log" 2>&1
echo "stamps rc $?"
tail -12 "$out/stamps_c768.log"
timeout 1200 python3 -m pytest tests -m gpu -q 2>&1 | tail -15 > "$out/pytest.log"
tail -8 "$out/pytest.log"
