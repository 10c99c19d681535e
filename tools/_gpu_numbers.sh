mkdir -p gpurun_out/num
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], "SDPD", round(d["value"],2), "ms/step", round(d["ms_per_step"],1), "sub-step", round(d["acoustic_step_ms"],2), "finite", d["finite"])'
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --tracers 4 --remap | tail -1 | python -c "$P" "c768 f64 tracers4+remap"
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --remap | tail -1 | python -c "$P" "c768 f64 remap"
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --tracers 4 | tail -1 | python -c "$P" "c768 f64 tracers4"
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --precision 32 | tail -1 | python -c "$P" "c768 f32"
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --precision 32 --nz 127 | tail -1 | python -c "$P" "c768 L127 f32"
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --nz 127 | tail -1 | python -c "$P" "c768 L127 f64"
python bench.py --no-cpu-baseline --steps 3 --warmup 1 --config c384 | tail -1 | python -c "$P" "c384"
python bench.py --no-cpu-baseline --steps 3 --warmup 1 --config c272 | tail -1 | python -c "$P" "c272"
python bench.py --no-cpu-baseline --steps 3 --warmup 1 --config c192 | tail -1 | python -c "$P" "c192"
