#!/usr/bin/env python3
"""What the fp32 STORAGE of the interface heights costs w, and what a per-level reference height would buy (round-4 review, item 7;
DESIGN §2 "fp32").  The fp64 oracle runs one acoustic sub-step three times from the same state:

  exact      every array in fp64;
  plain      the interface heights (gz of the C-grid half step, zh) rounded to fp32 wherever an operator stores them -- the fp32 build's
             storage, with every other array and all arithmetic left in fp64;
  offset     the same rounding applied to the DEVIATION from a per-level reference height z_ref(k) (the hydrostatic height of interface k in
             an isothermal 250 K column: a constant per level, so the 2-D transport of update_dz_d commutes with it), i.e. the heights
             stored as fp32 perturbations.

and prints max |w - w_exact| / max |w_exact| for both.  This isolates the one error source the analysis blames (everything else exact), so
the ratio plain / offset is the gain the eight-kernel change "heights as perturbations" could deliver at most.

    python tools/fp32_height_study.py [--nx 24] [--nz 79]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=24)
    ap.add_argument("--nz", type=int, default=79)
    ap.add_argument("--sub-steps", type=int, default=1)
    ap.add_argument("--solver32", action="store_true", help="also: the two Riemann solvers evaluated in fp32 on otherwise exact inputs")
    a = ap.parse_args()
    from fv3_oracle import nh as _nh
    from helpers import oracle_cube

    orig = {n: getattr(_nh, n) for n in ("update_dz_c", "riem_solver_c", "update_dz_d", "riem_solver3")}
    res = {}
    modes = ("exact", "plain", "offset") + (("solver32", "solver32_h64") if a.solver32 else ())
    for mode in modes:
        part, cfg, grids, ost, phis, odyn = oracle_cube(a.nx, (1, 1), a.nz, dict(n_split=a.sub_steps))
        g = odyn.c.GRAV
        # reference height of interface k: isothermal 250 K column under the reference pressures of the hybrid coordinate
        ak, bk = np.asarray(grids[0].ak), np.asarray(grids[0].bk)
        p_ref = np.maximum(ak + bk * 1.0e5, 1.0e-3)
        zref = -(odyn.c.RDGAS * 250.0 / g) * np.log(p_ref / 1.0e5)

        def rnd(z, scale=1.0):
            if mode == "exact":
                return
            ref = (zref * scale)[None, None, :] if mode == "offset" else 0.0
            z[...] = (z - ref).astype(np.float32).astype(np.float64) + ref

        def wrap(name, idx, scale):
            f = orig[name]

            def w(*args, **kw):
                out = f(*args, **kw)
                rnd(args[idx], scale)
                return out

            return w

        # (gz of the C-grid half step is a height here, as in the library's zh -> gz form; riem_solver_c stores gz = g * height)
        _nh.update_dz_c = wrap("update_dz_c", 5, 1.0)
        _nh.riem_solver_c = wrap("riem_solver_c", 9, g)
        _nh.update_dz_d = wrap("update_dz_d", 5, 1.0)
        _nh.riem_solver3 = wrap("riem_solver3", 11, 1.0)
        if mode in ("solver32", "solver32_h64"):
            # the two Riemann solvers evaluated in fp32 (every array argument cast to float32, results copied back): the arithmetic of the
            # fp32 build's solvers on otherwise exact inputs.  solver32_h64: the same with the interface heights (and the thickness formed
            # from them) left in fp64 inside the solver -- what a mixed-precision solver would do.
            def cast_wrap(name, hidx):
                f = orig[name]

                def w(*args, **kw):
                    args = list(args)
                    keep = {}
                    for i, x in enumerate(args):
                        if isinstance(x, np.ndarray) and x.dtype == np.float64 and not (mode == "solver32_h64" and i == hidx):
                            keep[i] = x
                            args[i] = x.astype(np.float32)
                    out = f(*args, **kw)
                    for i, x in keep.items():
                        x[...] = args[i]
                    return out

                return w

            _nh.riem_solver_c = cast_wrap("riem_solver_c", 9)
            _nh.riem_solver3 = cast_wrap("riem_solver3", 11)
        try:
            odyn(ost, 18.75 * a.sub_steps, 1)
        finally:
            for n, f in orig.items():
                setattr(_nh, n, f)
        res[mode] = ost
    nz = a.nz
    print(f"C{a.nx} L{nz}, {a.sub_steps} acoustic sub-step(s), heights up to {zref.max() / 1e3:.0f} km; field-scale relative error against the all-fp64 run")
    for name in ("w", "delz", "pt", "u"):
        sc = max(np.abs(s[name][3:-4, 3:-4, :nz]).max() for s in res["exact"])
        row = {}
        for mode in modes[1:]:
            row[mode] = max(np.abs(x[name][3:-4, 3:-4, :nz] - y[name][3:-4, 3:-4, :nz]).max() for x, y in zip(res[mode], res["exact"])) / sc
        gain = row["plain"] / row["offset"] if row["offset"] > 0 else float("inf")
        extra = "".join(f"   {m} {row[m]:.2e}" for m in modes[3:])
        print(f"  {name:5s} fp32 heights {row['plain']:.2e}   fp32 perturbation heights {row['offset']:.2e}   gain {gain:.1f} x" + extra)


if __name__ == "__main__":
    main()
