#!/bin/bash
# kernel timeline of one acoustic call of the emulated 1/8 share (which kernels run when, on which stream; gaps between them)
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r3p
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/t -o s -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-op-timing --emulate-share ${1:-8} > $out/log 2>&1
ls -la $out/t
