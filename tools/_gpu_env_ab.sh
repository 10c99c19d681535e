# usage: _gpu_env_ab.sh "ENV1=.." "ENV2=.." ...  (bench op timings per environment; "-" = default)
for e in "$@"; do
  echo "== [$e]"
  if [ "$e" = "-" ]; then python bench.py --no-cpu-baseline --steps 2 --warmup 1 > /tmp/b.log 2>/dev/null; else env $e python bench.py --no-cpu-baseline --steps 2 --warmup 1 > /tmp/b.log 2>/dev/null; fi
  tail -1 /tmp/b.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); o=d['operators_ms_per_substep']; print('SDPD', round(d['value'],2), 'sub-step', round(d['acoustic_step_ms'],2), {k: round(v,2) for k,v in o.items() if v>2.5})"
done
