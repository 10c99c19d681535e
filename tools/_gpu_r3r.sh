#!/bin/bash
# resident waves per CU of the transposed tile-edge marches vs their time and L2-miss bytes
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r3r
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --k-split 1 --n-split 1 --no-cpu-baseline --no-op-timing"
for v in 0 3 2 1; do
  export FV3_EDGE_WAVES_PER_CU=$v
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f$v -o p -- $B > $out/f$v.log 2>&1
  python3 - <<PY
import csv, collections
acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
for r in csv.DictReader(open("$out/f$v/p_counter_collection.csv")):
    n = r["Kernel_Name"]
    if "dsw_scalars" not in n: continue
    key = n[n.index("dsw_scalars_t"):][:36]
    if ", 2, false, true" not in key: continue
    a = acc[key]; a[0] += float(r["Counter_Value"]); a[1] += 1; a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, a in sorted(acc.items()):
    print("waves/CU=$v", k, "FETCH GB", round(a[0] * 2 * 1024 / 1e9 / a[1], 2), "ms", round(a[2] / a[1] / 1e6, 3))
PY
  find $out -name "*.csv" -delete
done
