#!/bin/bash
# round-4 experiment N: rows per marching wave (FV3_SEG) and levels per XCD walk (FV3_Q4_KB) re-measured on the final marches
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r4n
mkdir -p "$out"
cd "$R"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > "$out/bench_$name.log" 2>&1; echo "== $name"; tail -1 "$out/bench_$name.log" | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
o = l['operators_ms_per_substep']
print(l['value'], l['acoustic_step_ms'], l['state_checksum']['u'], {k: round(o[k], 2) for k in ('c_sw', 'd_sw', 'update_dz_d', 'nh_p_grad')})
"; }
run base X=1
run seg128 FV3_SEG=128
run seg192 FV3_SEG=192
run seg64 FV3_SEG=64
run kb8 FV3_Q4_KB=8
run kb32 FV3_Q4_KB=32
run base2 X=1
run seg128b FV3_SEG=128
