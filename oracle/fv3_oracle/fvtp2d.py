"""ORACLE (test infrastructure) -- Lin-Rood 2-D flux-form transport ``fv_tp_2d`` and the
del-n damping fluxes  [SURVEY A.4 / A.4.3; FV3 tp_core fv_tp_2d, deln_flux; sw_core
del6_vt_flux; pyFV3 ``fvtp2d.FiniteVolumeTransport``, ``delnflux.DelnFlux(NoSG)``
-- ctor/call shape in REF examples/notebooks/functions.py:935-951].
All arrays may carry a trailing k axis; the operators are level-independent.
"""
from __future__ import annotations

import numpy as np

from .ppm import xppm, yppm
from .util import Dom, copy_corners


def _cc(D, q, direction):
    if D.sw or D.se or D.ne or D.nw:
        copy_corners(D, q, direction)


def del6_vt_flux(D: Dom, nord: int, damp: float, q):
    """del-(2*nord+2) damping fluxes of q (FV3 del6_vt_flux). Returns fx2, fy2, d2."""
    S = D.sl
    m = D.m
    is_, ie, js, je = D.is_, D.ie, D.js, D.je
    d2 = np.zeros_like(q)
    fx2 = np.zeros_like(q)
    fy2 = np.zeros_like(q)
    i1, i2, j1, j2 = is_ - 1 - nord, ie + 1 + nord, js - 1 - nord, je + 1 + nord
    d2[S(i1, i2, j1, j2)] = damp * q[S(i1, i2, j1, j2)]
    if nord > 0:
        _cc(D, d2, 1)
    fx2[S(is_ - nord, ie + nord + 1, js - nord, je + nord)] = m.del6_v[S(is_ - nord, ie + nord + 1, js - nord, je + nord)] * (
        d2[S(is_ - nord - 1, ie + nord, js - nord, je + nord)] - d2[S(is_ - nord, ie + nord + 1, js - nord, je + nord)]
    )
    if nord > 0:
        _cc(D, d2, 2)
    fy2[S(is_ - nord, ie + nord, js - nord, je + nord + 1)] = m.del6_u[S(is_ - nord, ie + nord, js - nord, je + nord + 1)] * (
        d2[S(is_ - nord, ie + nord, js - nord - 1, je + nord)] - d2[S(is_ - nord, ie + nord, js - nord, je + nord + 1)]
    )
    for n in range(1, nord + 1):
        nt = nord - n
        R = S(is_ - nt - 1, ie + nt + 1, js - nt - 1, je + nt + 1)
        d2[R] = (
            fx2[R] - fx2[S(is_ - nt, ie + nt + 2, js - nt - 1, je + nt + 1)] + fy2[R] - fy2[S(is_ - nt - 1, ie + nt + 1, js - nt, je + nt + 2)]
        ) * m.rarea[R]
        _cc(D, d2, 1)
        R = S(is_ - nt, ie + nt + 1, js - nt, je + nt)
        fx2[R] = m.del6_v[R] * (d2[R] - d2[S(is_ - nt - 1, ie + nt, js - nt, je + nt)])
        _cc(D, d2, 2)
        R = S(is_ - nt, ie + nt, js - nt, je + nt + 1)
        fy2[R] = m.del6_u[R] * (d2[R] - d2[S(is_ - nt, ie + nt, js - nt - 1, je + nt)])
    return fx2, fy2, d2


def deln_flux(D: Dom, nord: int, damp: float, q, fx, fy, mass=None):
    """Add del-n damping fluxes of q to (fx, fy) in place (FV3 deln_flux)."""
    S = D.sl
    is_, ie, js, je = D.is_, D.ie, D.js, D.je
    if mass is None:
        fx2, fy2, _ = del6_vt_flux(D, nord, damp, q)
        fx[S(is_, ie + 1, js, je)] += fx2[S(is_, ie + 1, js, je)]
        fy[S(is_, ie, js, je + 1)] += fy2[S(is_, ie, js, je + 1)]
    else:
        fx2, fy2, _ = del6_vt_flux(D, nord, 1.0, q)
        damp2 = 0.5 * damp
        fx[S(is_, ie + 1, js, je)] += damp2 * (mass[S(is_ - 1, ie, js, je)] + mass[S(is_, ie + 1, js, je)]) * fx2[S(is_, ie + 1, js, je)]
        fy[S(is_, ie, js, je + 1)] += damp2 * (mass[S(is_, ie, js - 1, je)] + mass[S(is_, ie, js, je + 1)]) * fy2[S(is_, ie, js, je + 1)]


def fv_tp_2d(D: Dom, q, crx, cry, xfx, yfx, ra_x, ra_y, hord=6, mfx=None, mfy=None, mass=None, nord=None, damp_c=None):
    """2-D transport fluxes (fx at i=is..ie+1, j=js..je; fy at i=is..ie, j=js..je+1).

    ``q`` gets its cube-corner halo overwritten (copy_corners), like the reference.
    """
    S = D.sl
    m = D.m
    is_, ie, js, je, isd, ied, jsd, jed = D.is_, D.ie, D.js, D.je, D.isd, D.ied, D.jsd, D.jed
    _cc(D, q, 2)
    fy2 = yppm(D, q, cry, isd, ied, hord)
    fyy = np.zeros_like(q)
    R = S(isd, ied, js, je + 1)
    fyy[R] = yfx[R] * fy2[R]
    q_i = np.zeros_like(q)
    R = S(isd, ied, js, je)
    q_i[R] = (q[R] * m.area[R] + fyy[R] - fyy[S(isd, ied, js + 1, je + 1)]) / ra_y[R]
    fx = xppm(D, q_i, crx, js, je, hord)

    _cc(D, q, 1)
    fx2 = xppm(D, q, crx, jsd, jed, hord)
    fx1 = np.zeros_like(q)
    R = S(is_, ie + 1, jsd, jed)
    fx1[R] = xfx[R] * fx2[R]
    q_j = np.zeros_like(q)
    R = S(is_, ie, jsd, jed)
    q_j[R] = (q[R] * m.area[R] + fx1[R] - fx1[S(is_ + 1, ie + 1, jsd, jed)]) / ra_x[R]
    fy = yppm(D, q_j, cry, is_, ie, hord)

    Rx = S(is_, ie + 1, js, je)
    Ry = S(is_, ie, js, je + 1)
    if mfx is not None:
        fx[Rx] = 0.5 * (fx[Rx] + fx2[Rx]) * mfx[Rx]
        fy[Ry] = 0.5 * (fy[Ry] + fy2[Ry]) * mfy[Ry]
        if nord is not None and damp_c is not None and mass is not None and damp_c > 1.0e-4:
            damp = (damp_c * D.grid.da_min) ** (nord + 1)
            deln_flux(D, nord, damp, q, fx, fy, mass=mass)
    else:
        fx[Rx] = 0.5 * (fx[Rx] + fx2[Rx]) * xfx[Rx]
        fy[Ry] = 0.5 * (fy[Ry] + fy2[Ry]) * yfx[Ry]
        if nord is not None and damp_c is not None and damp_c > 1.0e-4:
            damp = (damp_c * D.grid.da_min) ** (nord + 1)
            deln_flux(D, nord, damp, q, fx, fy)
    # keep only the defined ranges
    out_fx = np.zeros_like(q)
    out_fy = np.zeros_like(q)
    out_fx[Rx] = fx[Rx]
    out_fy[Ry] = fy[Ry]
    return out_fx, out_fy
