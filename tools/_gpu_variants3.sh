# usage: _gpu_variants3.sh "<env assignments for the build 1>" ...   (e.g. "FV3_FLAGS_fv3_tp4=-ffp-contract=fast FV3_FLAGS_fv3_tp2d=-ffp-contract=fast"; "-" = default)
for fl in "$@"; do
  if [ "$fl" = "-" ]; then python -m pace_amd.build --precision 64 > /tmp/build.log 2>&1; else env $fl python -m pace_amd.build --precision 64 > /tmp/build.log 2>&1; fi || { tail -5 /tmp/build.log; continue; }
  echo "== [$fl]"
  python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); o=d['operators_ms_per_substep']; print('SDPD', round(d['value'],2), 'sub-step', round(d['acoustic_step_ms'],2), {k: round(v,2) for k,v in o.items() if v>2.5})"
done
