#!/usr/bin/env python3
"""How much of the halo-exchange time does the sequencer hide?  One process plays 1 of N GPUs alone (bench.py --emulate-share:
the multi-process plans, pack / unpack kernels and start / wait protocol, messages looped back on the device) and every update
is given an artificial transfer time (FV3_LOOPBACK_DELAY_US: a device-side stall on the communication stream).  Three
sequencer forms per delay:
  overlap   the product path: exchanges on the communication stream, delp / pt / q_con started inside d_sw, frame-first passes for
            uc / vc and u / v / w
  no-frame  the same without the frame-first passes (FV3_FRAME_FIRST=0)
  exposed   everything on the compute stream (FV3_HALO_STREAM=0): every microsecond of delay is paid
Prints ms per acoustic sub-step; (exposed - overlap) is what the overlap hides.
    python tools/halo_overlap_experiment.py [--share 8] [--config c768] [--delays 0,250,500,1000] [--steps 2]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(share, config, delay, env_extra, steps):
    env = dict(os.environ, FV3_LOOPBACK_DELAY_US=str(delay), **env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emulate-share", str(share), "--config", config, "--steps", str(steps), "--warmup", "1",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-2000:])
    d = json.loads(out.stdout.strip().splitlines()[-1])
    return d["acoustic_step_ms"], d["finite"], d["operators_ms_per_substep"].get("halo", 0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--share", type=int, default=8)
    ap.add_argument("--config", default="c768")
    ap.add_argument("--delays", default="0,250,500,1000")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    forms = {"overlap": {}, "no-frame": {"FV3_FRAME_FIRST": "0"}, "exposed": {"FV3_HALO_STREAM": "0", "FV3_FRAME_FIRST": "0"}}
    rows = []
    print(f"| delay per update (us) | " + " | ".join(f"{k} ms/sub-step" for k in forms) + " | hidden by the overlap (ms) | of the injected |")
    print("|---:|" + "---:|" * (len(forms) + 2))
    base = {}
    for delay in [float(x) for x in a.delays.split(",")]:
        res = {k: run(a.share, a.config, delay, e, a.steps) for k, e in forms.items()}
        if delay == 0:
            base = {k: v[0] for k, v in res.items()}
        injected = res["exposed"][0] - base.get("exposed", res["exposed"][0])
        hidden = res["exposed"][0] - res["overlap"][0] - (base.get("exposed", 0) - base.get("overlap", 0))
        rows.append({"delay_us": delay, **{k: v[0] for k, v in res.items()}, "finite": all(v[1] for v in res.values()), "hidden_ms": hidden, "injected_ms": injected})
        print(f"| {delay:g} | " + " | ".join(f"{res[k][0]:.2f}" for k in forms) + f" | {hidden:.2f} | {100 * hidden / injected if injected > 0 else 0:.0f} % |", flush=True)
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
