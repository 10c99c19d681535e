# usage: _gpu_variants.sh <file-stem> <grep-pattern> "<flags1>" "<flags2>" ...   (timing of build variants; results may be wrong)
stem=$1; pat=$2; shift 2
mkdir -p gpurun_out/var
for fl in "$@"; do
  env "FV3_FLAGS_$stem=$fl" python -m pace_amd.build --precision 64 > gpurun_out/var/build.log 2>&1 || { tail -5 gpurun_out/var/build.log; continue; }
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/var/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/var/stats.log 2>&1 )
  python tools/summarize_rocprof.py gpurun_out/var/stats/s_kernel_stats.csv 70 > gpurun_out/var/kernel_stats.md
  echo "== [$fl]"; grep -i "$pat" gpurun_out/var/kernel_stats.md | head -3
  rm -rf gpurun_out/var/stats
done
