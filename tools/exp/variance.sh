#!/bin/bash
# Round-5 study of c_sw's run-to-run drift on one box: N consecutive bench processes, rocm-smi sampled twice a second beside
# each (shader / memory clock, power, edge / junction / HBM temperature), per-operator times of every run -> gpurun_out/${FV3_VARIANCE_TAG:-r06_variance}/summary.md
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/${FV3_VARIANCE_TAG:-r06_variance}
mkdir -p "$out"
cd "$R"
N=${1:-5}
rocm-smi --showclocks --showpower --showtemp > "$out/smi_idle.txt" 2>&1
for i in $(seq 1 $N); do
  ( while true; do echo "== $(date +%s.%N)"; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature"; sleep 0.5; done ) > "$out/smi_$i.txt" 2>&1 &
  sp=$!
  python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > "$out/bench_$i.log" 2>&1
  kill $sp 2>/dev/null
  wait $sp 2>/dev/null
done
python3 - "$out" "$N" <<'PY'
import json, re, sys, statistics as st
out, n = sys.argv[1], int(sys.argv[2])
rows = []
for i in range(1, n + 1):
    line = [l for l in open(f"{out}/bench_{i}.log") if l.startswith("{")]
    if not line:
        continue
    l = json.loads(line[-1])
    o = l["operators_ms_per_substep"]
    smi = open(f"{out}/smi_{i}.txt").read()
    def vals(pat):
        return [float(x) for x in re.findall(pat, smi)]
    sclk = vals(r"sclk clock level: \d+: \((\d+)Mhz\)")
    mclk = vals(r"mclk clock level: \d+: \((\d+)Mhz\)")
    pw = vals(r"Power \(W\): ([0-9.]+)")
    tj = vals(r"Temperature \(Sensor junction\) \(C\): ([0-9.]+)")
    th = vals(r"Temperature \(Sensor (?:HBM|memory)[^)]*\) \(C\): ([0-9.]+)")
    # (the last two thirds of the samples: the timed steps; the first third is grid generation on the host)
    def tail(v):
        v = v[len(v) // 3:]
        return (st.mean(v), max(v)) if v else (float("nan"), float("nan"))
    rows.append((i, l["acoustic_step_ms"], o.get("c_sw"), o.get("d_sw"), o.get("riem_solver3"), tail(sclk), tail(mclk), tail(pw), tail(tj), tail(th)))
with open(f"{out}/summary.md", "w") as f:
    f.write("| run | ms / sub-step | c_sw | d_sw | riem_solver3 | sclk MHz (mean / max) | mclk MHz | power W (mean / max) | junction C | HBM C |\n|---|---:|---:|---:|---:|---|---|---|---|---|\n")
    for r in rows:
        f.write(f"| {r[0]} | {r[1]:.2f} | {r[2]:.2f} | {r[3]:.2f} | {r[4]:.2f} | {r[5][0]:.0f} / {r[5][1]:.0f} | {r[6][0]:.0f} | {r[7][0]:.0f} / {r[7][1]:.0f} | {r[8][0]:.0f} / {r[8][1]:.0f} | {r[9][0]:.0f} / {r[9][1]:.0f} |\n")
print(open(f"{out}/summary.md").read())
PY
head -40 "$out/smi_idle.txt"
