#!/usr/bin/env python3
"""Side-by-side debug mode: the oracle and the library, operator by operator, on the same inputs -- the analogue of the
reference's ``pair_debug`` run mode [REF driver/pace/driver/driver.py:83-87], which steps two backends together and
compares after every stencil.

    python tests/pair_debug.py [--nx 12] [--nz 8] [--layout 1] [--backend hip:gfx950|hostemu] [--n-split 2] [--tol 1e-10]

The oracle runs one acoustic call with every operator recorded (inputs before / outputs after); each operator is then
replayed ALONE through the C ABI on the recorded inputs, and the report lists, per operator and output field, the
field-scale relative difference (max |a - b| / max |b|) of the worst rank -- so a deviation is attributed to the operator
that produces it instead of surfacing several operators later.  Lives under tests/ because it executes the oracle
(checker code); exit status 1 when any difference exceeds --tol.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

from helpers import oracle_cube  # noqa: E402
from test_operator_parity import Dev, Recorder  # noqa: E402

# operator -> (C entry, argument plan).  An argument is ("f", i) = field from oracle argument i, ("s", i) = scalar,
# ("c", value) = constant; outputs: (name, oracle argument index, region, level count or None)
U = lambda D: D.sl(1, D.nx, 1, D.ny + 1)  # noqa: E731
V = lambda D: D.sl(1, D.nx + 1, 1, D.ny)  # noqa: E731
CELLS = lambda D: D.sl(1, D.nx, 1, D.ny)  # noqa: E731
RING = lambda D: D.sl(0, D.nx + 1, 0, D.ny + 1)  # noqa: E731
CORN = lambda D: D.sl(1, D.nx + 1, 1, D.ny + 1)  # noqa: E731


def plans(nz):
    F, S = (lambda i: ("f", i)), (lambda i: ("s", i))
    return [
        ("c_sw", "c_sw", [F(0), F(1), F(2), F(3), F(4), F(5), F(6), F(7), F(8), F(9), F(10), F(11), F(12), ("new", None), ("new", None), S(13)],
         [("uc", 5, V, nz), ("vc", 6, U, nz), ("ua", 7, RING, nz), ("va", 8, RING, nz), ("ut", 9, V, nz), ("vt", 10, U, nz), ("divgd", 11, CORN, nz), ("omga", 12, CELLS, nz)]),
        ("update_dz_c", "update_dz_c", [F(1), F(2), F(3), F(4), F(5), S(6)], [("gz", 4, CELLS, None), ("ws3", 5, CELLS, None)]),
        ("riem_solver_c", "riem_solver_c", [S(0), F(1), S(2), F(3), F(4), F(5), F(6), F(7), F(8), F(9), F(10)], [("gz", 8, CELLS, None), ("pef", 9, CELLS, None)]),
        ("p_grad_c", "p_grad_c", [F(2), F(3), F(4), F(5), F(6), S(7)], [("uc", 2, V, nz), ("vc", 3, U, nz)]),
        ("d_sw", "d_sw", [F(i) for i in range(2, 25)] + [S(25)],
         [("delp", 3, CELLS, nz), ("pt", 4, CELLS, nz), ("u", 5, U, nz), ("v", 6, V, nz), ("w", 7, CELLS, nz), ("q_con", 21, CELLS, nz), ("mfx", 13, V, nz), ("mfy", 14, U, nz),
          ("crx", 17, V, nz), ("cry", 18, U, nz), ("xfx", 19, V, nz), ("yfx", 20, U, nz), ("heat_source", 23, CELLS, nz)]),
        ("update_dz_d", "update_dz_d", [F(3), F(4), F(5), F(6), F(7), F(8), F(9), S(10)], [("zh", 4, CELLS, None), ("wsd", 9, CELLS, None)]),
        ("riem_solver3", "riem_solver3", [S(0), S(1), F(2), S(3), F(4), F(5), F(6), F(7), F(8), F(9), F(10), F(11), F(12), F(13), F(14), F(15), F(16)],
         [("w", 16, CELLS, nz), ("delz", 6, CELLS, nz), ("zh", 10, CELLS, None), ("ppe", 12, CELLS, None), ("pk3", 13, CELLS, None), ("pe", 11, CELLS, None), ("peln", 15, CELLS, None)]),
        ("nh_p_grad", "nh_p_grad", [F(0), F(1), F(2), F(3), F(4), F(5), S(6), S(7), S(8)], [("u", 0, U, nz), ("v", 1, V, nz)]),
        ("ray_fast", "ray_fast", [F(1), F(2), F(3), S(6), S(7)], [("u", 1, U, nz), ("v", 2, V, nz), ("w", 3, CELLS, nz)]),
        ("del2_cubed", "del2_cubed", [F(0), S(1), ("c", 3)], [("heat_source", 0, CELLS, nz)]),
        ("apply_diffusive_heating", "apply_diffusive_heating", [F(0), F(1), F(2), F(3), F(4), S(5)], [("pt", 4, CELLS, nz)]),
    ]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--nx", type=int, default=12)
    ap.add_argument("--nz", type=int, default=8)
    ap.add_argument("--layout", type=int, default=1)
    ap.add_argument("--backend", default=None)
    ap.add_argument("--n-split", type=int, default=1)
    ap.add_argument("--tol", type=float, default=1e-10)
    a = ap.parse_args(argv)
    import torch

    backend = a.backend or ("hip:gfx950" if torch.cuda.is_available() else "hostemu")
    if backend == "hostemu":
        from pace_amd import build

        build.build(64, hostemu=True, verbose=False)
    part, cfg, grids, ost, phis, odyn = oracle_cube(a.nx, (a.layout, a.layout), a.nz, dict(n_split=a.n_split))
    with Recorder() as rec:
        odyn(ost, 225.0, 1)
    dv = Dev(backend, grids, cfg)
    nr = part.total_ranks
    worst_all = 0.0
    print(f"pair_debug: C{a.nx} L{a.nz} layout {a.layout}x{a.layout} ({nr} ranks), n_split {a.n_split}, backend {backend}")
    print(f"{'operator':26s} {'sub-step':>8s}  " + "field: field-scale relative difference (worst rank)")
    for oname, entry, args, outs in plans(a.nz):
        calls = rec.calls.get(oname, [])
        for it in range(len(calls) // nr):
            cl = calls[it * nr : (it + 1) * nr]
            if oname == "c_sw":  # the oracle returns (delpc, ptc) instead of taking them
                for c in cl:
                    c["outs"] = list(c["outs"]) + [c["ret"][0], c["ret"][1]]
            qs, call_args = {}, []
            for kind, i in args:
                if kind == "f":
                    if cl[0]["ins"][i] is None:
                        qs[i] = dv.q([np.zeros(dv.sf.sizer.storage_shape[:2] + (a.nz + 1,))] * nr)
                    else:
                        qs[i] = dv.q([c["ins"][i] for c in cl])
                    call_args.append(qs[i].fref)
                elif kind == "s":
                    v = cl[0]["ins"][i]
                    call_args.append(int(bool(v)) if isinstance(v, (bool, np.bool_)) else float(v))
                elif kind == "new":
                    q = dv.q([np.zeros(dv.sf.sizer.storage_shape[:2] + (a.nz + 1,))] * nr)
                    qs[("new", len(call_args))] = q
                    call_args.append(q.fref)
                else:
                    call_args.append(i)
            dv.sf.call(entry, *call_args)
            if oname == "c_sw":
                outs_ = outs + [("delpc", 14, CELLS, a.nz), ("ptc", 15, CELLS, a.nz)]
                news = [q for k, q in qs.items() if isinstance(k, tuple)]
                qs[14], qs[15] = news[0], news[1]
            else:
                outs_ = outs
            cells = []
            for name, i, region, kk in outs_:
                worst = 0.0
                for r, c in enumerate(cl):
                    want = np.asarray(c["outs"][i])
                    got = qs[i].numpy(r)
                    R = region(c["D"])
                    if want.ndim == 3 and want.shape[2] == 1:
                        want = want[:, :, 0]
                    if got.ndim == 3 and want.ndim == 3:
                        n = min(want.shape[2], got.shape[2]) if kk is None else kk
                        g_, w_ = got[R][:, :, :n], want[R][:, :, :n]
                    else:
                        g_, w_ = got[R], want[R]
                    sc = np.abs(w_).max()
                    e = np.abs(g_ - w_).max()
                    worst = max(worst, e / sc if sc > 0 else e)
                worst_all = max(worst_all, worst)
                cells.append(f"{name}: {worst:.1e}" + (" <<<" if worst > a.tol else ""))
            print(f"{oname:26s} {it:8d}  " + "  ".join(cells))
    print(f"worst difference {worst_all:.2e} (tolerance {a.tol:g})")
    return 1 if worst_all > a.tol else 0


if __name__ == "__main__":
    sys.exit(main())
