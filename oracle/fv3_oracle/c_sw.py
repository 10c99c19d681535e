"""ORACLE (test infrastructure) -- C-grid half step ``c_sw`` with ``d2a2c_vect`` and
``divergence_corner``  [SURVEY A.2; FV3 sw_core.F90 c_sw / d2a2c_vect /
divergence_corner; pyFV3 ``c_sw.CGridShallowWaterDynamics``, ``d2a2c_vect.DGrid2AGrid2CGridVectors``;
checkpoint variables REF tests/savepoint/thresholds/fv_dynamics.yaml:2-75].
"""
from __future__ import annotations

import numpy as np

from .util import Dom, fill_4corners

A1 = 0.5625
A2 = -0.0625
C1 = -2.0 / 14.0
C2 = 11.0 / 14.0
C3 = 5.0 / 14.0
BIG = 1.0e30


def _edge_interpolate4(ua, dxa):
    """ua, dxa: lists of 4 consecutive cells straddling the edge (2 outside, 2 inside)."""
    u1, u2, u3, u4 = ua
    d1, d2, d3, d4 = dxa
    return 0.5 * (((2.0 * d2 + d1) * u2 - d2 * u1) / (d1 + d2) + ((2.0 * d3 + d4) * u3 - d3 * u4) / (d3 + d4))


def d2a2c_vect(D: Dom, u, v, ua, va, uc, vc, ut, vt):
    """D-grid winds -> A-grid (ua, va) and C-grid (uc, vc) + contravariant (ut, vt); in place."""
    S = D.sl
    o = D.o
    m = D.m
    is_, ie, js, je, npx, npy = D.is_, D.ie, D.js, D.je, D.npx, D.npy
    isd, ied, jsd, jed = D.isd, D.ied, D.jsd, D.jed
    W, E, Sd, N = D.west, D.east, D.south, D.north
    utmp = np.full_like(u, BIG)
    vtmp = np.full_like(u, BIG)

    # 4th-order interior (wider than strictly needed so ua/va are defined on is-2..ie+2)
    jlo = 4 if Sd else js - 2
    jhi = npy - 4 if N else je + 2
    ilo = 4 if W else isd
    ihi = npx - 4 if E else ied
    R = S(ilo, ihi, jlo, jhi)
    utmp[R] = A2 * (u[S(ilo, ihi, jlo - 1, jhi - 1)] + u[S(ilo, ihi, jlo + 2, jhi + 2)]) + A1 * (u[R] + u[S(ilo, ihi, jlo + 1, jhi + 1)])
    jlo_v = 4 if Sd else jsd
    jhi_v = npy - 4 if N else jed
    ilo_v = 4 if W else is_ - 2
    ihi_v = npx - 4 if E else ie + 2
    R = S(ilo_v, ihi_v, jlo_v, jhi_v)
    vtmp[R] = A2 * (v[S(ilo_v - 1, ihi_v - 1, jlo_v, jhi_v)] + v[S(ilo_v + 2, ihi_v + 2, jlo_v, jhi_v)]) + A1 * (v[R] + v[S(ilo_v + 1, ihi_v + 1, jlo_v, jhi_v)])

    def two_pt(i0, i1, j0, j1):
        R = S(i0, i1, j0, j1)
        utmp[R] = 0.5 * (u[R] + u[S(i0, i1, j0 + 1, j1 + 1)])
        vtmp[R] = 0.5 * (v[R] + v[S(i0 + 1, i1 + 1, j0, j1)])

    # within 3 cells of a tile edge: 2-point averages
    if Sd:
        two_pt(isd, ied, jsd, 3)
    if N:
        two_pt(isd, ied, npy - 3, jed)
    jm0 = 4 if Sd else jsd
    jm1 = npy - 4 if N else jed
    if W:
        two_pt(isd, 3, jm0, jm1)
    if E:
        two_pt(npx - 3, ied, jm0, jm1)

    # contravariant components at cell centres
    R = S(is_ - 2, ie + 2, js - 2, je + 2)
    ua[R] = (utmp[R] - vtmp[R] * m.cosa_s[R]) * m.rsin2[R]
    va[R] = (vtmp[R] - utmp[R] * m.cosa_s[R]) * m.rsin2[R]

    def U(i, j):
        return (i + o, j + o)

    # ---- A -> C, x direction: fix utmp in the corner halo
    if D.sw:
        for i in range(-2, 1):
            utmp[U(i, 0)] = -vtmp[U(0, 1 - i)]
    if D.se:
        for i in range(0, 3):
            utmp[U(npx + i, 0)] = vtmp[U(npx, i + 1)]
    if D.ne:
        for i in range(0, 3):
            utmp[U(npx + i, npy)] = -vtmp[U(npx, je - i)]
    if D.nw:
        for i in range(-2, 1):
            utmp[U(i, npy)] = vtmp[U(0, je + i)]

    ifirst = 3 if W else is_ - 1
    ilast = npx - 2 if E else ie + 2
    R = S(ifirst, ilast, js - 1, je + 1)
    uc[R] = A2 * (utmp[S(ifirst - 2, ilast - 2, js - 1, je + 1)] + utmp[S(ifirst + 1, ilast + 1, js - 1, je + 1)]) + A1 * (
        utmp[S(ifirst - 1, ilast - 1, js - 1, je + 1)] + utmp[R]
    )
    ut[R] = (uc[R] - v[R] * m.cosa_u[R]) * m.rsin_u[R]

    if D.sw:
        ua[U(-1, 0)] = -va[U(0, 2)]
        ua[U(0, 0)] = -va[U(0, 1)]
    if D.se:
        ua[U(npx, 0)] = va[U(npx, 1)]
        ua[U(npx + 1, 0)] = va[U(npx, 2)]
    if D.ne:
        ua[U(npx, npy)] = -va[U(npx, npy - 1)]
        ua[U(npx + 1, npy)] = -va[U(npx, npy - 2)]
    if D.nw:
        ua[U(-1, npy)] = va[U(0, npy - 2)]
        ua[U(0, npy)] = va[U(0, npy - 1)]

    def col(a, i):
        return a[i + o : i + o + 1, js - 1 + o : je + 1 + o + 1]

    def setcol(a, i, val):
        a[i + o : i + o + 1, js - 1 + o : je + 1 + o + 1] = val

    if W:
        setcol(uc, 0, C1 * col(utmp, -2) + C2 * col(utmp, -1) + C3 * col(utmp, 0))
        ut1 = _edge_interpolate4([col(ua, i) for i in (-1, 0, 1, 2)], [col(m.dxa, i) for i in (-1, 0, 1, 2)])
        setcol(ut, 1, ut1)
        setcol(uc, 1, np.where(ut1 > 0.0, ut1 * col(m.sin_sg3, 0), ut1 * col(m.sin_sg1, 1)))
        setcol(uc, 2, C1 * col(utmp, 3) + C2 * col(utmp, 2) + C3 * col(utmp, 1))
        setcol(ut, 0, (col(uc, 0) - col(v, 0) * col(m.cosa_u, 0)) * col(m.rsin_u, 0))
        setcol(ut, 2, (col(uc, 2) - col(v, 2) * col(m.cosa_u, 2)) * col(m.rsin_u, 2))
    if E:
        setcol(uc, npx - 1, C1 * col(utmp, npx - 3) + C2 * col(utmp, npx - 2) + C3 * col(utmp, npx - 1))
        utn = _edge_interpolate4([col(ua, i) for i in (npx - 2, npx - 1, npx, npx + 1)], [col(m.dxa, i) for i in (npx - 2, npx - 1, npx, npx + 1)])
        setcol(ut, npx, utn)
        setcol(uc, npx, np.where(utn > 0.0, utn * col(m.sin_sg3, npx - 1), utn * col(m.sin_sg1, npx)))
        setcol(uc, npx + 1, C3 * col(utmp, npx) + C2 * col(utmp, npx + 1) + C1 * col(utmp, npx + 2))
        setcol(ut, npx - 1, (col(uc, npx - 1) - col(v, npx - 1) * col(m.cosa_u, npx - 1)) * col(m.rsin_u, npx - 1))
        setcol(ut, npx + 1, (col(uc, npx + 1) - col(v, npx + 1) * col(m.cosa_u, npx + 1)) * col(m.rsin_u, npx + 1))

    # ---- y direction
    if D.sw:
        for j in range(-2, 1):
            vtmp[U(0, j)] = -utmp[U(1 - j, 0)]
    if D.nw:
        for j in range(0, 3):
            vtmp[U(0, npy + j)] = utmp[U(j + 1, npy)]
    if D.se:
        for j in range(-2, 1):
            vtmp[U(npx, j)] = utmp[U(ie + j, 0)]
    if D.ne:
        for j in range(0, 3):
            vtmp[U(npx, npy + j)] = -utmp[U(ie - j, npy)]
    if D.sw:
        va[U(0, -1)] = -ua[U(2, 0)]
        va[U(0, 0)] = -ua[U(1, 0)]
    if D.se:
        va[U(npx, 0)] = ua[U(npx - 1, 0)]
        va[U(npx, -1)] = ua[U(npx - 2, 0)]
    if D.ne:
        va[U(npx, npy)] = -ua[U(npx - 1, npy)]
        va[U(npx, npy + 1)] = -ua[U(npx - 2, npy)]
    if D.nw:
        va[U(0, npy)] = ua[U(1, npy)]
        va[U(0, npy + 1)] = ua[U(2, npy)]

    jfirst = 3 if Sd else js - 1
    jlast = npy - 2 if N else je + 2
    R = S(is_ - 1, ie + 1, jfirst, jlast)
    vc[R] = A2 * (vtmp[S(is_ - 1, ie + 1, jfirst - 2, jlast - 2)] + vtmp[S(is_ - 1, ie + 1, jfirst + 1, jlast + 1)]) + A1 * (
        vtmp[S(is_ - 1, ie + 1, jfirst - 1, jlast - 1)] + vtmp[R]
    )
    vt[R] = (vc[R] - u[R] * m.cosa_v[R]) * m.rsin_v[R]

    def row(a, j):
        return a[is_ - 1 + o : ie + 1 + o + 1, j + o : j + o + 1]

    def setrow(a, j, val):
        a[is_ - 1 + o : ie + 1 + o + 1, j + o : j + o + 1] = val

    if Sd:
        vt1 = _edge_interpolate4([row(va, j) for j in (-1, 0, 1, 2)], [row(m.dya, j) for j in (-1, 0, 1, 2)])
        setrow(vt, 1, vt1)
        setrow(vc, 1, np.where(vt1 > 0.0, vt1 * row(m.sin_sg4, 0), vt1 * row(m.sin_sg2, 1)))
        setrow(vc, 0, C1 * row(vtmp, -2) + C2 * row(vtmp, -1) + C3 * row(vtmp, 0))
        setrow(vt, 0, (row(vc, 0) - row(u, 0) * row(m.cosa_v, 0)) * row(m.rsin_v, 0))
        setrow(vc, 2, C1 * row(vtmp, 3) + C2 * row(vtmp, 2) + C3 * row(vtmp, 1))
        setrow(vt, 2, (row(vc, 2) - row(u, 2) * row(m.cosa_v, 2)) * row(m.rsin_v, 2))
    if N:
        vtn = _edge_interpolate4([row(va, j) for j in (npy - 2, npy - 1, npy, npy + 1)], [row(m.dya, j) for j in (npy - 2, npy - 1, npy, npy + 1)])
        setrow(vt, npy, vtn)
        setrow(vc, npy, np.where(vtn > 0.0, vtn * row(m.sin_sg4, npy - 1), vtn * row(m.sin_sg2, npy)))
        setrow(vc, npy - 1, C1 * row(vtmp, npy - 3) + C2 * row(vtmp, npy - 2) + C3 * row(vtmp, npy - 1))
        setrow(vt, npy - 1, (row(vc, npy - 1) - row(u, npy - 1) * row(m.cosa_v, npy - 1)) * row(m.rsin_v, npy - 1))
        setrow(vc, npy + 1, C1 * row(vtmp, npy + 2) + C2 * row(vtmp, npy + 1) + C3 * row(vtmp, npy))
        setrow(vt, npy + 1, (row(vc, npy + 1) - row(u, npy + 1) * row(m.cosa_v, npy + 1)) * row(m.rsin_v, npy + 1))


def divergence_corner(D: Dom, u, v, ua, va, divg_d):
    """Corner divergence of the D-grid wind (FV3 divergence_corner); writes divg_d on is..ie+1, js..je+1."""
    S = D.sl
    o = D.o
    m = D.m
    is_, ie, js, je, npx, npy = D.is_, D.ie, D.js, D.je, D.npx, D.npy
    uf = np.zeros_like(u)
    vf = np.zeros_like(u)
    R = S(is_ - 1, ie + 1, js, je + 1)
    Rm = S(is_ - 1, ie + 1, js - 1, je)
    uf[R] = (u[R] - 0.25 * (va[Rm] + va[R]) * (m.cos_sg4[Rm] + m.cos_sg2[R])) * m.dyc[R] * 0.5 * (m.sin_sg4[Rm] + m.sin_sg2[R])
    for flag, j in ((D.south, 1), (D.north, npy)):
        if flag:
            R1 = S(is_ - 1, ie + 1, j, j)
            R1m = S(is_ - 1, ie + 1, j - 1, j - 1)
            uf[R1] = u[R1] * m.dyc[R1] * 0.5 * (m.sin_sg4[R1m] + m.sin_sg2[R1])
    R = S(is_, ie + 1, js - 1, je + 1)
    Rm = S(is_ - 1, ie, js - 1, je + 1)
    vf[R] = (v[R] - 0.25 * (ua[Rm] + ua[R]) * (m.cos_sg3[Rm] + m.cos_sg1[R])) * m.dxc[R] * 0.5 * (m.sin_sg3[Rm] + m.sin_sg1[R])
    for flag, i in ((D.west, 1), (D.east, npx)):
        if flag:
            R1 = S(i, i, js - 1, je + 1)
            R1m = S(i - 1, i - 1, js - 1, je + 1)
            vf[R1] = v[R1] * m.dxc[R1] * 0.5 * (m.sin_sg3[R1m] + m.sin_sg1[R1])
    R = S(is_, ie + 1, js, je + 1)
    divg_d[R] = vf[S(is_, ie + 1, js - 1, je)] - vf[R] + uf[S(is_ - 1, ie, js, je + 1)] - uf[R]
    if D.sw:
        divg_d[1 + o, 1 + o] -= vf[1 + o, 0 + o]
    if D.se:
        divg_d[npx + o, 1 + o] -= vf[npx + o, 0 + o]
    if D.ne:
        divg_d[npx + o, npy + o] += vf[npx + o, npy + o]
    if D.nw:
        divg_d[1 + o, npy + o] += vf[1 + o, npy + o]
    divg_d[R] = m.rarea_c[R] * divg_d[R]


def c_sw(D: Dom, delp, pt, u, v, w, uc, vc, ua, va, ut, vt, divgd, omga, dt2, nord=1):
    """Half-step C-grid update.  Same argument order as the reference operator
    (``CGridShallowWaterDynamics.__call__``); returns (delpc, ptc).  ``omga`` receives wc.
    All arrays [i, j, k] over the nz layers passed in."""
    S = D.sl
    o = D.o
    m = D.m
    is_, ie, js, je, npx, npy = D.is_, D.ie, D.js, D.je, D.npx, D.npy
    delpc = np.zeros_like(delp)
    ptc = np.zeros_like(delp)

    d2a2c_vect(D, u, v, ua, va, uc, vc, ut, vt)
    if nord > 0:
        divergence_corner(D, u, v, ua, va, divgd)

    R = S(is_ - 1, ie + 2, js - 1, je + 1)
    Rm = S(is_ - 2, ie + 1, js - 1, je + 1)
    ut[R] = np.where(ut[R] > 0.0, dt2 * ut[R] * m.dy[R] * m.sin_sg3[Rm], dt2 * ut[R] * m.dy[R] * m.sin_sg1[R])
    R = S(is_ - 1, ie + 1, js - 1, je + 2)
    Rm = S(is_ - 1, ie + 1, js - 2, je + 1)
    vt[R] = np.where(vt[R] > 0.0, dt2 * vt[R] * m.dx[R] * m.sin_sg4[Rm], dt2 * vt[R] * m.dx[R] * m.sin_sg2[R])

    # first-order upwind transport of delp, pt, w
    for q in (delp, pt, w):
        fill_4corners(D, q, 1)
    R = S(is_ - 1, ie + 2, js - 1, je + 1)
    Rm = S(is_ - 2, ie + 1, js - 1, je + 1)
    pos = ut[R] > 0.0
    fx1 = np.zeros_like(delp)
    fx = np.zeros_like(delp)
    fx2 = np.zeros_like(delp)
    fx1[R] = ut[R] * np.where(pos, delp[Rm], delp[R])
    fx[R] = fx1[R] * np.where(pos, pt[Rm], pt[R])
    fx2[R] = fx1[R] * np.where(pos, w[Rm], w[R])
    for q in (delp, pt, w):
        fill_4corners(D, q, 2)
    R = S(is_ - 1, ie + 1, js - 1, je + 2)
    Rm = S(is_ - 1, ie + 1, js - 2, je + 1)
    pos = vt[R] > 0.0
    fy1 = np.zeros_like(delp)
    fy = np.zeros_like(delp)
    fy2 = np.zeros_like(delp)
    fy1[R] = vt[R] * np.where(pos, delp[Rm], delp[R])
    fy[R] = fy1[R] * np.where(pos, pt[Rm], pt[R])
    fy2[R] = fy1[R] * np.where(pos, w[Rm], w[R])
    R = S(is_ - 1, ie + 1, js - 1, je + 1)
    Rx = S(is_, ie + 2, js - 1, je + 1)
    Ry = S(is_ - 1, ie + 1, js, je + 2)
    delpc[R] = delp[R] + (fx1[R] - fx1[Rx] + fy1[R] - fy1[Ry]) * m.rarea[R]
    ptc[R] = (pt[R] * delp[R] + (fx[R] - fx[Rx] + fy[R] - fy[Ry]) * m.rarea[R]) / delpc[R]
    omga[R] = (w[R] * delp[R] + (fx2[R] - fx2[Rx] + fy2[R] - fy2[Ry]) * m.rarea[R]) / delpc[R]

    # kinetic energy and vorticity on the C grid
    ke = np.zeros_like(delp)
    vort = np.zeros_like(delp)
    Rx = S(is_, ie + 2, js - 1, je + 1)
    Ry = S(is_ - 1, ie + 1, js, je + 2)
    ke[R] = np.where(ua[R] > 0.0, uc[R], uc[Rx])
    vort[R] = np.where(va[R] > 0.0, vc[R], vc[Ry])
    # tile-edge cells: project with the edge metric (FV3 sw_corrected branch)
    for flag, iw, ie_ in ((D.west, 1, 0), (D.east, npx, npx - 1)):
        if flag:
            # ua > 0 at cell iw (upwind face i=iw) ; ua <= 0 at cell ie_ (face i=ie_+1)
            Rc = S(iw, iw, js - 1, je + 1)
            ke[Rc] = np.where(ua[Rc] > 0.0, uc[Rc] * m.sin_sg1[Rc] + v[Rc] * m.cos_sg1[Rc], ke[Rc])
            Rc = S(ie_, ie_, js - 1, je + 1)
            Rp = S(ie_ + 1, ie_ + 1, js - 1, je + 1)
            ke[Rc] = np.where(ua[Rc] > 0.0, ke[Rc], uc[Rp] * m.sin_sg3[Rc] + v[Rp] * m.cos_sg3[Rc])
    for flag, jw, je_ in ((D.south, 1, 0), (D.north, npy, npy - 1)):
        if flag:
            Rc = S(is_ - 1, ie + 1, jw, jw)
            vort[Rc] = np.where(va[Rc] > 0.0, vc[Rc] * m.sin_sg2[Rc] + u[Rc] * m.cos_sg2[Rc], vort[Rc])
            Rc = S(is_ - 1, ie + 1, je_, je_)
            Rp = S(is_ - 1, ie + 1, je_ + 1, je_ + 1)
            vort[Rc] = np.where(va[Rc] > 0.0, vort[Rc], vc[Rp] * m.sin_sg4[Rc] + u[Rp] * m.cos_sg4[Rc])
    ke[R] = 0.5 * dt2 * (ua[R] * ke[R] + va[R] * vort[R])

    # circulation -> absolute vorticity on corners
    fxc = np.zeros_like(delp)
    fyc = np.zeros_like(delp)
    Ra = S(is_, ie + 1, js - 1, je + 1)
    fxc[Ra] = uc[Ra] * m.dxc[Ra]
    Rb = S(is_ - 1, ie + 1, js, je + 1)
    fyc[Rb] = vc[Rb] * m.dyc[Rb]
    Rv = S(is_, ie + 1, js, je + 1)
    vort[Rv] = fxc[S(is_, ie + 1, js - 1, je)] - fxc[Rv] - fyc[S(is_ - 1, ie, js, je + 1)] + fyc[Rv]
    if D.sw:
        vort[1 + o, 1 + o] += fyc[0 + o, 1 + o]
    if D.se:
        vort[npx + o, 1 + o] -= fyc[npx + o, 1 + o]
    if D.ne:
        vort[npx + o, npy + o] -= fyc[npx + o, npy + o]
    if D.nw:
        vort[1 + o, npy + o] += fyc[0 + o, npy + o]
    vort[Rv] = m.fC[Rv] + m.rarea_c[Rv] * vort[Rv]

    # vorticity transport + KE gradient -> time-centred C-grid winds
    Ru = S(is_, ie + 1, js, je)
    fy1u = dt2 * (v[Ru] - uc[Ru] * m.cosa_u[Ru]) / m.sina_u[Ru]
    fy1f = np.zeros_like(delp)
    fy1f[Ru] = fy1u
    for flag, i in ((D.west, 1), (D.east, npx)):
        if flag:
            Rc = S(i, i, js, je)
            fy1f[Rc] = dt2 * v[Rc]
    fyv = np.where(fy1f[Ru] > 0.0, vort[Ru], vort[S(is_, ie + 1, js + 1, je + 1)])
    uc[Ru] = uc[Ru] + fy1f[Ru] * fyv + m.rdxc[Ru] * (ke[S(is_ - 1, ie, js, je)] - ke[Ru])
    Rvv = S(is_, ie, js, je + 1)
    fx1f = np.zeros_like(delp)
    fx1f[Rvv] = dt2 * (u[Rvv] - vc[Rvv] * m.cosa_v[Rvv]) / m.sina_v[Rvv]
    for flag, j in ((D.south, 1), (D.north, npy)):
        if flag:
            Rc = S(is_, ie, j, j)
            fx1f[Rc] = dt2 * u[Rc]
    fxv = np.where(fx1f[Rvv] > 0.0, vort[Rvv], vort[S(is_ + 1, ie + 1, js, je + 1)])
    vc[Rvv] = vc[Rvv] - fx1f[Rvv] * fxv + m.rdyc[Rvv] * (ke[S(is_, ie, js - 1, je)] - ke[Rvv])
    return delpc, ptc
