# Kernel timeline (rocprofv3 --kernel-trace -> tools/share_timeline.py) of the emulated 1/8 share and of the whole problem: launches per sub-step, idle time between kernels, short launches.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r06_timeline
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for cfg in "share8:--emulate-share 8" "whole:"; do
  name=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --output-format csv -d "$out/$name" -o t -- python3 "$R/bench.py" $args --steps 3 --warmup 1 --no-cpu-baseline --no-op-timing > "$out/$name.log" 2>&1
  f=$(find "$out/$name" -name "t_kernel_trace.csv" | head -1)
  echo "== $name"
  python3 "$R/tools/share_timeline.py" "$f" 48
  find "$out/$name" -name "*kernel_trace.csv" -delete
done
