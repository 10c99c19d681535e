#!/bin/bash
# L2-miss read bytes of the scalar marches with the level-major launch geometry (FV3_Q4_KB=16) against the default
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r3n
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --k-split 1 --n-split 1 --no-cpu-baseline --no-op-timing"
for v in 0 16; do
  export FV3_Q4_KB=$v
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f$v -o p -- $B > $out/f$v.log 2>&1
  python3 - <<PY
import csv, collections
acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
for r in csv.DictReader(open("$out/f$v/p_counter_collection.csv")):
    n = r["Kernel_Name"]
    if "dsw_scalars" not in n: continue
    key = n[n.index("dsw_scalars_t"):][:40]
    a = acc[key]; a[0] += float(r["Counter_Value"]); a[1] += 1; a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, a in sorted(acc.items()):
    print("Q4_KB=$v", k, "calls", a[1], "FETCH_SIZE raw KB x2 -> GB", round(a[0] * 2 * 1024 / 1e9 / a[1], 2) if a[0] < 1e9 else round(a[0]*2/1e9/a[1],2), "ms", round(a[2] / a[1] / 1e6, 3))
PY
  find $out -name "*.csv" -delete
done
