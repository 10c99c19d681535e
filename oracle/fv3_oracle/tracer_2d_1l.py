"""ORACLE (test infrastructure, never the shipped path) -- sub-cycled 2-D tracer advection, ``tracer_2d_1l``.

Restates the reference operator ``pyFV3.stencils.tracer_2d_1l.TracerAdvection`` (constructed with a
``FiniteVolumeTransport`` and called as ``tracer_advection(tracers, dp1, mfxd, mfyd, cxd, cyd)``
[REF examples/notebooks/functions.py:34, 916-951, 1037-1044]; savepoint ``Tracer2D1L-In/Out`` with the variables
``cxd, cyd, dp1, mfxd, mfyd`` [REF tests/savepoint/thresholds/fv_dynamics.yaml:328-360]; ``hord_tr: 8``
[REF driver/examples/configs/baroclinic_c12.yaml:60]) from the published algorithm (GFDL_atmos_cubed_sphere
``fv_tracer2d.F90: tracer_2d_1L``; Lin & Rood 1996; Putman & Lin 2007).  PARITY UNPINNED like the rest of the oracle
(util.py header).

    cx, cy, mfx, mfy: Courant numbers / mass fluxes accumulated over the acoustic sub-steps of one remapping interval (d_sw)
    xfx = cx * dxa(upwind) * dy * sin_sg(upwind),  yfx likewise                       (area swept through a face)
    cmax = max over cells and levels of  max(|cx|, |cy|) + 1 - sin_sg5 ;  all-reduce MAX over the ranks
    n_split = int(1 + cmax);  if n_split > 1: cx, cy, xfx, yfx, mfx, mfy are scaled by 1 / n_split (in place)
    ra_x = area + xfx - xfx[i+1],  ra_y = area + yfx - yfx[j+1]
    n_split times:  dp2 = dp1 + (mfx - mfx[i+1] + mfy - mfy[j+1]) * rarea
                    every tracer q:  (fx, fy) = fv_tp_2d(q, cx, cy, xfx, yfx, ra_x, ra_y; mfx, mfy)
                                     q = (q * dp1 + (fx - fx[i+1] + fy - fy[j+1]) * rarea) / dp2
                    not the last time: dp1 = dp2, halo update of the tracers
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np

from .fvtp2d import fv_tp_2d
from .util import Dom


def flux_compute(D: Dom, cx, cy, xfx, yfx):
    """Area fluxes from the accumulated Courant numbers (cx on i = is..ie+1, j = jsd..jed; cy on i = isd..ied, j = js..je+1)."""
    S, m = D.sl, D.m
    is_, ie, js, je, isd, ied, jsd, jed = D.is_, D.ie, D.js, D.je, D.isd, D.ied, D.jsd, D.jed
    R, Rm = S(is_, ie + 1, jsd, jed), S(is_ - 1, ie, jsd, jed)
    xfx[R] = np.where(cx[R] > 0.0, cx[R] * m.dxa[Rm] * m.dy[R] * m.sin_sg3[Rm], cx[R] * m.dxa[R] * m.dy[R] * m.sin_sg1[R])
    R, Rm = S(isd, ied, js, je + 1), S(isd, ied, js - 1, je)
    yfx[R] = np.where(cy[R] > 0.0, cy[R] * m.dya[Rm] * m.dx[R] * m.sin_sg4[Rm], cy[R] * m.dya[R] * m.dx[R] * m.sin_sg2[R])


def cmax_local(D: Dom, cx, cy) -> float:
    """max over the compute cells of max(|cx|, |cy|) + 1 - sin_sg5."""
    R = D.sl(D.is_, D.ie, D.js, D.je)
    return float(np.max(np.maximum(np.abs(cx[R]), np.abs(cy[R])) + 1.0 - D.m.sin_sg5[R]))


def tracer_2d_1l(doms: List[Dom], tracers: List[Dict[str, np.ndarray]], dp1: List[np.ndarray], mfx, mfy, cx, cy, hord: int,
                 halo_update: Optional[Callable[[List[np.ndarray]], None]] = None) -> int:
    """All ranks of the cube in one process (like OracleAcousticDynamics): per-rank lists of [i, j, k] arrays.
    ``tracers[r]`` maps names to fields; dp1, mfx, mfy, cx, cy are modified in place like the reference's
    (Tracer2D1L-Out holds them).  ``halo_update(list of per-rank arrays)`` fills the tracer halos between sub-cycles.
    Returns n_split."""
    nr = len(doms)
    xfx = [np.zeros_like(a) for a in cx]
    yfx = [np.zeros_like(a) for a in cy]
    cmax = 0.0
    for r, D in enumerate(doms):
        flux_compute(D, cx[r], cy[r], xfx[r], yfx[r])
        cmax = max(cmax, cmax_local(D, cx[r], cy[r]))  # (the all-reduce MAX of the reference)
    n_split = int(1.0 + cmax)
    if n_split > 1:
        frac = 1.0 / n_split
        for r, D in enumerate(doms):
            S = D.sl
            Rx, Ry = S(D.is_, D.ie + 1, D.jsd, D.jed), S(D.isd, D.ied, D.js, D.je + 1)
            for a in (cx[r], xfx[r]):
                a[Rx] = a[Rx] * frac
            for a in (cy[r], yfx[r]):
                a[Ry] = a[Ry] * frac
            Rx, Ry = S(D.is_, D.ie + 1, D.js, D.je), S(D.is_, D.ie, D.js, D.je + 1)
            mfx[r][Rx] = mfx[r][Rx] * frac
            mfy[r][Ry] = mfy[r][Ry] * frac
    ra = []
    for r, D in enumerate(doms):
        S, m = D.sl, D.m
        ra_x, ra_y = np.zeros_like(cx[r]), np.zeros_like(cx[r])
        R = S(D.is_, D.ie, D.jsd, D.jed)
        ra_x[R] = m.area[R] + xfx[r][R] - xfx[r][S(D.is_ + 1, D.ie + 1, D.jsd, D.jed)]
        R = S(D.isd, D.ied, D.js, D.je)
        ra_y[R] = m.area[R] + yfx[r][R] - yfx[r][S(D.isd, D.ied, D.js + 1, D.je + 1)]
        ra.append((ra_x, ra_y))
    names = list(tracers[0])
    for it in range(n_split):
        for r, D in enumerate(doms):
            S, m = D.sl, D.m
            C, Ce, Cn = S(D.is_, D.ie, D.js, D.je), S(D.is_ + 1, D.ie + 1, D.js, D.je), S(D.is_, D.ie, D.js + 1, D.je + 1)
            dp2 = dp1[r][C] + (mfx[r][C] - mfx[r][Ce] + mfy[r][C] - mfy[r][Cn]) * m.rarea[C]
            for n in names:
                q = tracers[r][n]
                fx, fy = fv_tp_2d(D, q, cx[r], cy[r], xfx[r], yfx[r], ra[r][0], ra[r][1], hord, mfx=mfx[r], mfy=mfy[r])
                q[C] = (q[C] * dp1[r][C] + (fx[C] - fx[Ce] + fy[C] - fy[Cn]) * m.rarea[C]) / dp2
            if it < n_split - 1:
                dp1[r][C] = dp2
        if it < n_split - 1 and halo_update is not None:
            for n in names:
                halo_update([tracers[r][n] for r in range(nr)])
    return n_split
