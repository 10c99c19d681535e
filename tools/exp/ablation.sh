#!/bin/bash
# Round-5 ablation of the marching kernels (profiles/r05_ablation.md): per-kernel times of the product build, of the loads-and-stores-only build
# (-DPX_ABL=1: every load and store of a row step kept, the arithmetic replaced by one sum) and of the arithmetic-only build (-DPX_ABL=2: the
# loads inside the march replaced by registers), each from one `rocprofv3 --kernel-trace --stats` run of a short bench; then the shader clock the
# product build ran at (GRBM_GUI_ACTIVE over the kernels' durations, a separate --pmc pass).  Variant libraries are built in the build container:
#   for n in 1 2; do FV3_LIB_TAG=abl$n FV3_FLAGS_fv3_tp4x=-DPX_ABL=$n FV3_FLAGS_fv3_tp2x=-DPX_ABL=$n python -m pace_amd.build --precision 64; done
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r05_ablation
mkdir -p "$out"
cd "$R"
bash tools/exp/kstat.sh r05abl "" "pair_march|single_march" full: loads_stores:FV3_LIB_TAG=abl1 arithmetic:FV3_LIB_TAG=abl2 > "$out/kstat.log" 2>&1
cat "$out/kstat.log"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_clk" -o p -- python3 "$R/bench.py" --steps 1 --warmup 0 --k-split 1 --n-split 2 --no-cpu-baseline --no-op-timing > "$out/pmc_clk.log" 2>&1
cd "$R"
python3 - "$out/pmc_clk/p_counter_collection.csv" <<'PY'
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    m = re.search(r"(pair_march_t<\d>|single_march_t<\d, \w+>|csw_fused_stream|fv3_riem_solver3|fv3_riem_solver_c|fxadv|nh_pgf_fused)", r["Kernel_Name"])
    if not m:
        continue
    e = acc[m.group(1)]
    e[0] += float(r["Counter_Value"])
    e[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    e[2] += 1
print("| kernel | launches | mean ms | GRBM_GUI_ACTIVE cycles / ns = shader clock (GHz) |\n|---|---:|---:|---:|")
for k, (c, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"| {k} | {n} | {ns / n / 1e6:.3f} | {c / ns:.3f} |")
PY
find "$out" -name "*counter_collection.csv" -delete
