#!/usr/bin/env python3
"""Round-off growth of the ORACLE at the reference's savepoints, calibrated the way the reference calibrates its own
thresholds, next to the numbers the reference holds.

The reference validates this path against external savepoint data with per-variable thresholds that it derives from itself:
10 trials of one ``step_dynamics`` whose input state is perturbed at round-off level (``ndsl.testing.perturb``: every value
times 1 + U(-1e-16, 1e-16)), threshold = 10 x the largest difference to the first trial
[REF tests/savepoint/test_checkpoints.py:118-128,161-195]; the results for the C12 6-rank baroclinic case are committed in
[REF tests/savepoint/thresholds/fv_dynamics.yaml:2-170].  Those magnitudes say how strongly one C_SW / D_SW call amplifies
last-bit noise in each variable -- a property of the ALGORITHM.  Running the same procedure on the oracle (C12 L79,
baroclinic wave, 6 ranks, one acoustic sub-step) and landing orders of magnitude away from the reference's numbers for a
variable would point at a restatement error in what feeds that variable.  Since round 2 the trial also runs the oracle's
vertical remap after the acoustic call and compares the `Remapping-Out` variables the same way.  It does not lift "parity unpinned" (the
reference's data stay external); it is the only reference-held NUMBER there is to hold the oracle against.

    python tests/threshold_study.py [--write]      # --write: refresh tests/golden/threshold_study_c12.json

Lives under tests/: it executes the oracle.  tests/golden/reference_thresholds_fv_dynamics.json is the down-selected
reference data (made by tools/make_fixtures.py --thresholds where /root/reference exists).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
# savepoint variable -> (recorded operator, oracle argument index); c_sw(D, delp, pt, u, v, w, uc, vc, ua, va, ut, vt, divgd, omga, dt2)
C_SW_OUT = {"delpd": 0, "ptd": 1, "ud": 2, "vd": 3, "wd": 4, "ucd": 5, "vcd": 6, "uad": 7, "vad": 8, "utd": 9, "vtd": 10, "divgdd": 11}
# d_sw(D, cfg, col, delpc, delp, pt, u, v, w, uc, vc, ua, va, divgd, mfx, mfy, cx, cy, crx, cry, xfx, yfx, q_con, zh, heat, diss, dt)
D_SW_OUT = {"delpcd": 2, "delpd": 3, "ptd": 4, "ud": 5, "vd": 6, "wd": 7, "ucd": 8, "vcd": 9, "uad": 10, "vad": 11, "divgdd": 12, "mfxd": 13, "mfyd": 14, "xfxd": 19, "yfxd": 20}


# The ABSOLUTE thresholds are compared (a difference in the variable's own unit).  The reference's "relative" numbers are
# pointwise |a - b| / |b| maxima and depend on how close to zero its input data happen to come (the oracle's zonally
# symmetric analytic state has exact zeros in uc / vc / fluxes), so they are not a property of the algorithm alone.
NOT_COMPARABLE = {
    "uad": "the reference's INPUT already differs by 640 in uad / vad (C_SW-In threshold): its A-grid winds hold fill values",
    "vad": "see uad",
    "mfxd": "reference threshold is 'anything' (relative 100: the accumulated mass fluxes are not reproducible in its own trials)",
    "mfyd": "see mfxd",
    "delpcd": "dead work array after d_sw: the two implementations leave different scratch data in it",
    "wd": "w == 0 in the analytic state and c_sw / d_sw keep it 0; the reference's serialized state has w != 0",
}
# (Remapping-Out: every variable listed is compared; the reference run is the moist nwat = 6 configuration with the saturation
#  adjustment inside its remap, the oracle's is dry -- orders of magnitude are what is being compared)



def perturb(states, rng):
    """ndsl.testing.perturb: round-off level multiplicative noise on every float array, in place."""
    for s in states:
        for a in s.values():
            if isinstance(a, np.ndarray) and a.dtype == np.float64:
                a *= 1.0 + rng.uniform(-1e-16, 1e-16, size=a.shape)


def one_trial(seed, perturbed):
    from fv3_oracle.dyn_core import OracleAcousticDynamics
    from pace_amd.config import AcousticDynamicsConfig
    from pace_amd.constants import get_constants
    from pace_amd.grid import make_grid
    from pace_amd.init import baroclinic_state
    from pace_amd.topology import CubedSpherePartitioner
    from test_operator_parity import Recorder

    nx, nz = 12, 79
    part = CubedSpherePartitioner(nx, (1, 1))
    cfg = AcousticDynamicsConfig(npx=nx + 1, npy=nx + 1, npz=nz, layout=(1, 1), n_split=1, k_split=1)
    grids = [make_grid(part, r, nz=nz) for r in range(6)]
    init = [baroclinic_state(g) for g in grids]
    phis = [s.pop("phis") for s in init]
    if perturbed:
        perturb(init, np.random.default_rng(seed))
    dyn = OracleAcousticDynamics(part, grids, cfg, get_constants(), phis)
    with Recorder() as rec:
        dyn(init, 225.0, 1)
    out = {}
    for sp, op, table in (("C_SW-Out", "c_sw", C_SW_OUT), ("D_SW-Out", "d_sw", D_SW_OUT)):
        for var, idx in table.items():
            out[f"{sp}/{var}"] = [np.asarray(c["outs"][idx]).copy() for c in rec.calls[op][:6]]
    # the vertical remap that closes step_dynamics (k_split = 1: the reference's single remap is its last step, which leaves the
    # temperature in pt -- compared with our pt * pkz)  [REF tests/savepoint/thresholds/fv_dynamics.yaml:227-326]
    from fv3_oracle import remap as o_remap

    c = get_constants()
    for r, D in enumerate(dyn.doms):
        s = init[r]
        C = D.sl(1, D.nx, 1, D.ny)
        # what the acoustic call hands to the tracer advection: the accumulated Courant numbers [REF fv_dynamics.yaml:328-343]
        out.setdefault("Tracer2D1L-In/cxd", []).append(s["cxd"][D.sl(1, D.nx + 1, 1, D.ny)][:, :, :nz].copy())
        out.setdefault("Tracer2D1L-In/cyd", []).append(s["cyd"][D.sl(1, D.nx, 1, D.ny + 1)][:, :, :nz].copy())
        # the state the whole acoustic call (incl. the Riemann solvers, the height updates, the pressure gradient) leaves: Remapping-In
        for var in ("delp", "delz", "pe", "peln", "pk", "pt", "w"):
            kk = nz + 1 if var in ("pe", "peln", "pk") else nz
            out.setdefault(f"Remapping-In/{var}", []).append(s[var][C][:, :, :kk].copy())
        out.setdefault("Remapping-In/u", []).append(s["u"][D.sl(1, D.nx, 1, D.ny + 1)][:, :, :nz].copy())
        out.setdefault("Remapping-In/v", []).append(s["v"][D.sl(1, D.nx + 1, 1, D.ny)][:, :, :nz].copy())
        out.setdefault("Remapping-In/wsd", []).append(dyn.tmp[r]["wsd"][C].copy())
        o_remap.lagrangian_to_eulerian(D, c, s, dyn.tmp[r]["wsd"].copy(), [])
        for var in ("delp", "delz", "pe", "peln", "pk", "pkz", "w"):
            kk = nz + 1 if var in ("pe", "peln", "pk") else nz
            out.setdefault(f"Remapping-Out/{var}", []).append(s[var][C][:, :, :kk].copy())
        out.setdefault("Remapping-Out/pt", []).append((s["pt"][C][:, :, :nz] * s["pkz"][C][:, :, :nz]).copy())
        out.setdefault("Remapping-Out/u", []).append(s["u"][D.sl(1, D.nx, 1, D.ny + 1)][:, :, :nz].copy())
        out.setdefault("Remapping-Out/v", []).append(s["v"][D.sl(1, D.nx + 1, 1, D.ny)][:, :, :nz].copy())
    return out


def study(n_trials=10, factor=10.0):
    first = one_trial(0, True)
    growth = {k: [0.0, 0.0] for k in first}
    for t in range(1, n_trials):
        cur = one_trial(t, True)
        for k in first:
            for a, b in zip(cur[k], first[k]):
                d = np.abs(a - b)
                growth[k][0] = max(growth[k][0], float(d.max()))
                nz_ = np.abs(b) > 0
                if nz_.any():
                    growth[k][1] = max(growth[k][1], float((d[nz_] / np.abs(b[nz_])).max()))
    return {k: {"absolute": factor * v[0], "relative": factor * v[1]} for k, v in growth.items()}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--trials", type=int, default=10)
    a = ap.parse_args(argv)
    ours = study(a.trials)
    ref = json.load(open(os.path.join(GOLD, "reference_thresholds_fv_dynamics.json")))
    print(f"{'savepoint/variable':22s} {'oracle abs':>11s} {'ref abs':>11s}  log10(oracle / ref)   note")
    rows = {}
    for k in sorted(ours):
        o, r = ours[k], ref.get(k)
        if r is None:
            continue
        var = k.split("/")[1]
        note = NOT_COMPARABLE.get(var, "")
        ratio = np.log10(o["absolute"] / r["absolute"]) if o["absolute"] > 0 and r["absolute"] > 0 else float("nan")
        rows[k] = dict(oracle_absolute=o["absolute"], reference_absolute=r["absolute"], log10_ratio=None if np.isnan(ratio) else float(ratio), comparable=not note)
        print(f"{k:22s} {o['absolute']:11.2e} {r['absolute']:11.2e}  {ratio:+6.1f}               {note}")
    if a.write:
        json.dump(rows, open(os.path.join(GOLD, "threshold_study_c12.json"), "w"), indent=1, sort_keys=True)
    return rows


if __name__ == "__main__":
    main()
