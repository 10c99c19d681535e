cd ${GRAFT_REPO_ROOT:-.}
run() { name=$1; shift; env "$@" > gpurun_out/dbg_$name.log 2>&1; grep '^{' gpurun_out/dbg_$name.log | tail -1 | python3 -c "
import sys, json
l = json.loads(sys.stdin.readline())
print('$name', round(l['acoustic_step_ms'], 3), l['finite'], repr(l['state_checksum']['u']))
"; }
run fused31 X=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-op-timing --emulate-share 8
run staged31 FV3_DSW_WINDSTAGE=staged python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-op-timing --emulate-share 8
run staged42 FV3_DSW_WINDSTAGE=staged python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-op-timing --emulate-share 8
run fused42acc0 FV3_ACC_STORE=0 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-op-timing --emulate-share 8
