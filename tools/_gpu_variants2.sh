# usage: _gpu_variants2.sh <grep-pattern> "<global extra flags 1>" ...   (FV3_EXTRA_FLAGS variants; prints bench op timings too)
pat=$1; shift
mkdir -p gpurun_out/var
for fl in "$@"; do
  env "FV3_EXTRA_FLAGS=$fl" python -m pace_amd.build --precision 64 > gpurun_out/var/build.log 2>&1 || { tail -5 gpurun_out/var/build.log; continue; }
  echo "== [$fl]"
  python bench.py --no-cpu-baseline --steps 2 --warmup 1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); o=d['operators_ms_per_substep']; print('SDPD', round(d['value'],2), 'sub-step', round(d['acoustic_step_ms'],2), {k: round(v,2) for k,v in o.items() if v>1})"
done
