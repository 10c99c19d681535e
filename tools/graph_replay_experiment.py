#!/usr/bin/env python3
"""How much of a model step is launch-bound: the acoustic calls of one step captured into a HIP graph and replayed, against the same calls launched
kernel by kernel (experiment R5-28 of tools/exp/EXPERIMENTS.md).  One process, device-local halo transport (every sub-domain on this GPU), no
per-operator event pairs.  The sequencer re-creates the same launches on every call once its lazy allocations have happened (two warm-up steps),
forks / joins its auxiliary stream with events -- which a capture follows -- and reads no device value on the host, so a step is capturable as it is.

    python tools/graph_replay_experiment.py --config c192 [--steps 10]

Prints one line per configuration: ms per acoustic sub-step launched directly / replayed from the graph, and the state checksum of both (bitwise equal)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pace_amd.harness import CONFIGS, DycoreHarness  # noqa: E402


def run(config, steps, nz=0):
    kw = dict(CONFIGS[config])
    if nz:
        kw["nz"] = nz
    sub = kw["k_split"] * kw["n_split"]
    res = {}
    for mode in ("direct", "graph"):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            h = DycoreHarness(device="cuda:0", dtype=torch.float64, **kw)
            for _ in range(2):
                h.step()
            s.synchronize()
            g = None
            if mode == "graph":
                g = torch.cuda.CUDAGraph()
                g.capture_begin()
                h.step()
                g.capture_end()
                s.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                if g is not None:
                    g.replay()
                else:
                    h.step()
            s.synchronize()
            dt = time.perf_counter() - t0
            chk = {k: float(getattr(h.state, k).storage.double().sum().item()) for k in ("delp", "pt", "u", "w")}
        res[mode] = (dt / steps / sub * 1e3, chk)
        del h, g
        torch.cuda.empty_cache()
    same = res["direct"][1] == res["graph"][1]
    print(f"{config}{' L' + str(nz) if nz else ''}: {res['direct'][0]:.3f} ms / sub-step launched directly, {res['graph'][0]:.3f} replayed from a graph of one step "
          f"({sub} sub-steps); states {'bitwise equal' if same else 'DIFFER: ' + repr(res)}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", action="append", default=[])
    ap.add_argument("--steps", type=int, default=8)
    a = ap.parse_args()
    for c in a.config or ["c192"]:
        run(c, a.steps)
