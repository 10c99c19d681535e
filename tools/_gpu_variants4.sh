# usage: _gpu_variants4.sh <grep-pattern> "<FV3_EXTRA_FLAGS 1>" ...   (rocprof kernel-time sums per build variant)
pat=$1; shift
mkdir -p gpurun_out/var
for fl in "$@"; do
  env "FV3_EXTRA_FLAGS=$fl" python -m pace_amd.build --precision 64 > gpurun_out/var/build.log 2>&1 || { tail -5 gpurun_out/var/build.log; continue; }
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/var/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-op-timing > $GRAFT_REPO_ROOT/gpurun_out/var/stats.log 2>&1 )
  python tools/summarize_rocprof.py gpurun_out/var/stats/s_kernel_stats.csv 90 > gpurun_out/var/kernel_stats.md
  echo "== [$fl]"; grep -i "$pat" gpurun_out/var/kernel_stats.md | awk -F'|' '{n+=$3; t+=$5} END {print "launches", n, "total ms", t, "per c_sw call", t/24}'
  rm -rf gpurun_out/var/stats
done
