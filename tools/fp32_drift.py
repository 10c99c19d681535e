#!/usr/bin/env python3
"""fp32 build (PACE_FLOAT_PRECISION=32) against the fp64 build over SEVERAL acoustic sub-steps: per-field field-scale relative
difference max|a32 - a64| / max|a64| after 1, 2, 6 and 12 sub-steps (one call with n_split = N each, same initial state), C96 L127
by default.  The fp64 HIP build stands in for the fp64 oracle here (they agree to 1e-11, tests/test_parity.py; the numpy oracle
needs ~20 s per sub-step at this size).
    python tools/fp32_drift.py [--nx 96] [--nz 127] [--json out.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

FIELDS = ("delp", "pt", "u", "v", "w", "delz", "q_con")


def run(nx, nz, n_split, dtype, backend):
    from pace_amd._testing import harness_for
    from pace_amd.harness import DycoreHarness

    h = harness_for(backend)(nx, nz=nz, layout=(1, 1), dt_atmos=18.75 * n_split, k_split=1, n_split=n_split, dtype=dtype, noise=0.01)
    h.step()
    h.synchronize()
    out = {}
    for n in FIELDS:
        q = getattr(h.state, n)
        out[n] = [q.numpy(i)[3 : 3 + nx, 3 : 3 + nx, :nz].astype(np.float64) for i in range(6)]
    return out


def drift_table(nx=96, nz=127, splits=(1, 2, 6, 12), backend="hip:gfx950"):
    table = {}
    for ns in splits:
        a64 = run(nx, nz, ns, torch.float64, backend)
        a32 = run(nx, nz, ns, torch.float32, backend)
        row = {}
        for n in FIELDS:
            sc = max(np.abs(b).max() for b in a64[n])
            row[n] = float(max(np.abs(a - b).max() for a, b in zip(a32[n], a64[n])) / sc)
        table[ns] = row
    return table


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=96)
    ap.add_argument("--nz", type=int, default=127)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    t = drift_table(a.nx, a.nz)
    print("| sub-steps | " + " | ".join(FIELDS) + " |")
    print("|---:|" + "---:|" * len(FIELDS))
    for ns, row in t.items():
        print(f"| {ns} | " + " | ".join(f"{row[n]:.1e}" for n in FIELDS) + " |")
    if a.json:
        json.dump(t, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
