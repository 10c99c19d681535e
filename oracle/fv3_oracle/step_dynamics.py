"""ORACLE (test infrastructure, never the shipped path) -- the body of ``DynamicalCore.step_dynamics`` for the dycore-only
configuration: ``k_split`` x [AcousticDynamics, tracer advection, Lagrangian-to-Eulerian remap] over all ranks of the cube
[REF driver/pace/driver/driver.py:494-504, 641; savepoints FVDynamics-In / -Out, Tracer2D1L-In / -Out, Remapping-In / -Out
of tests/savepoint/thresholds/fv_dynamics.yaml:171-360].  No physics, no moist thermodynamics (see remap_oracle.c).  The
sequence is the one of ``pace_amd.harness.DycoreHarness.step``; ``record`` receives the savepoints in the reference's names.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np

from . import remap as _remap
from . import tracer_2d_1l as _t2


def step_dynamics(odyn, consts, states: List[Dict[str, np.ndarray]], tracers: List[Dict[str, np.ndarray]], dt_atmos: float, k_split: int, hord_tr: int = 8,
                  record: Optional[Callable[[str, int, Dict[str, np.ndarray]], None]] = None):
    """``odyn``: an OracleAcousticDynamics; ``states[r]`` / ``tracers[r]``: the per-rank oracle arrays ([i, j, k], nz + 1 levels),
    updated in place.  ``record(savepoint, rank, {name: array})`` is called at the reference's checkpoints."""
    nz = odyn.doms[0].nz
    nr = len(states)
    rec = record or (lambda *a: None)
    dt = dt_atmos / k_split
    for r in range(nr):
        rec("FVDynamics-In", r, {**{k: states[r][k] for k in ("u", "v", "w", "delz", "ua", "va", "uc", "vc")}, **{f"tracer_{n}": q for n, q in tracers[r].items()}})
    for k in range(k_split):
        dp1 = [s["delp"].copy() for s in states]
        odyn(states, dt, k + 1)
        V = lambda a: np.ascontiguousarray(a[:, :, :nz])  # noqa: E731
        flux = {n: [V(s[n]) for s in states] for n in ("mfxd", "mfyd", "cxd", "cyd")}
        dp1n = [V(a) for a in dp1]
        trn = [{n: V(q) for n, q in t.items()} for t in tracers]
        for name in (trn[0] if trn else {}):
            odyn.ex.scalar([t[name] for t in trn])
        for r in range(nr):
            rec("Tracer2D1L-In", r, {"dp1": dp1n[r], "mfxd": flux["mfxd"][r], "mfyd": flux["mfyd"][r], "cxd": flux["cxd"][r], "cyd": flux["cyd"][r],
                                     **{f"tracer_{n}": q for n, q in trn[r].items()}})
        _t2.tracer_2d_1l(odyn.doms, trn, dp1n, flux["mfxd"], flux["mfyd"], flux["cxd"], flux["cyd"], hord_tr, halo_update=lambda fs: odyn.ex.scalar(fs))
        for r in range(nr):
            rec("Tracer2D1L-Out", r, {"dp1": dp1n[r], "mfxd": flux["mfxd"][r], "mfyd": flux["mfyd"][r], "cxd": flux["cxd"][r], "cyd": flux["cyd"][r],
                                      **{f"tracer_{n}": q for n, q in trn[r].items()}})
            for n in flux:
                states[r][n][:, :, :nz] = flux[n][r]
            for n, q in trn[r].items():
                tracers[r][n][:, :, :nz] = q
        for r, D in enumerate(odyn.doms):
            s = states[r]
            wsd = odyn.tmp[r]["wsd"]
            names = ("cappa", "delp", "delz", "pe", "peln", "pk", "pkz", "pt", "u", "v", "w")
            rec("Remapping-In", r, {**{n: s[n] for n in names}, "wsd": np.asarray(wsd).reshape(s["delp"].shape[:2]), **{f"tracer_{n}": q for n, q in tracers[r].items()}})
            tl = [np.ascontiguousarray(q) for q in tracers[r].values()]
            ps = _remap.lagrangian_to_eulerian(D, consts, s, np.asarray(wsd).copy(), tl)
            for (n, q), a in zip(tracers[r].items(), tl):
                q[...] = a
            rec("Remapping-Out", r, {**{n: s[n] for n in names}, "ps": ps, **{f"tracer_{n}": q for n, q in tracers[r].items()}})
    for r in range(nr):
        rec("FVDynamics-Out", r, {**{k: states[r][k] for k in ("u", "v", "w", "delz", "ua", "va", "uc", "vc")}, **{f"tracer_{n}": q for n, q in tracers[r].items()}})
