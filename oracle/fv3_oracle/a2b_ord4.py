"""ORACLE (test infrastructure) -- 4th-order cell-centre -> corner interpolation
``a2b_ord4``  [SURVEY A.13; FV3 a2b_edge.F90 a2b_ord4; pyFV3 ``a2b_ord4.AGrid2BGridFourthOrder``].
"""
from __future__ import annotations

import numpy as np

from .util import Dom

A1 = 0.5625
A2 = -0.0625
B1 = 7.0 / 12.0
B2 = -1.0 / 12.0
C1 = 2.0 / 3.0
C2 = -1.0 / 6.0
R3 = 1.0 / 3.0


def _extrap(fac, q1, q2):
    return q1 + fac * (q1 - q2)


def a2b_ord4(D: Dom, qin, replace=False):
    """Returns qout on corners i=is..ie+1, j=js..je+1 (3-D arrays, level independent)."""
    S = D.sl
    o = D.o
    m = D.m
    is_, ie, js, je, npx, npy = D.is_, D.ie, D.js, D.je, D.npx, D.npy
    W, E, Sd, N = D.west, D.east, D.south, D.north
    qout = np.zeros_like(qin)
    qx = np.zeros_like(qin)
    qy = np.zeros_like(qin)
    qxx = np.zeros_like(qin)
    qyy = np.zeros_like(qin)
    ce = D.grid.corner_extrap

    def Q(a, i, j):
        return a[i + o, j + o]

    # cube corners: mean of three one-sided extrapolations along the cell diagonals
    if D.sw:
        qout[1 + o, 1 + o] = (
            _extrap(ce[0, 0], Q(qin, 1, 1), Q(qin, 2, 2)) + _extrap(ce[0, 1], Q(qin, 0, 1), Q(qin, -1, 2)) + _extrap(ce[0, 2], Q(qin, 1, 0), Q(qin, 2, -1))
        ) * R3
    if D.se:
        qout[npx + o, 1 + o] = (
            _extrap(ce[1, 0], Q(qin, npx - 1, 1), Q(qin, npx - 2, 2))
            + _extrap(ce[1, 1], Q(qin, npx - 1, 0), Q(qin, npx - 2, -1))
            + _extrap(ce[1, 2], Q(qin, npx, 1), Q(qin, npx + 1, 2))
        ) * R3
    if D.ne:
        qout[npx + o, npy + o] = (
            _extrap(ce[2, 0], Q(qin, npx - 1, npy - 1), Q(qin, npx - 2, npy - 2))
            + _extrap(ce[2, 1], Q(qin, npx, npy - 1), Q(qin, npx + 1, npy - 2))
            + _extrap(ce[2, 2], Q(qin, npx - 1, npy), Q(qin, npx - 2, npy + 1))
        ) * R3
    if D.nw:
        qout[1 + o, npy + o] = (
            _extrap(ce[3, 0], Q(qin, 1, npy - 1), Q(qin, 2, npy - 2))
            + _extrap(ce[3, 1], Q(qin, 0, npy - 1), Q(qin, -1, npy - 2))
            + _extrap(ce[3, 2], Q(qin, 1, npy), Q(qin, 2, npy + 1))
        ) * R3

    is1 = 1 if W else is_ - 1
    ie1 = npx - 1 if E else ie + 1
    js1 = 1 if Sd else js - 1
    je1 = npy - 1 if N else je + 1
    is2 = 2 if W else is_
    ie2 = npx - 1 if E else ie + 1
    js2 = 2 if Sd else js
    je2 = npy - 1 if N else je + 1

    # ---- X interior
    jlo = 1 if Sd else js - 2
    jhi = npy - 1 if N else je + 2
    ilo = 3 if W else is_
    ihi = npx - 2 if E else ie + 1
    R = S(ilo, ihi, jlo, jhi)
    qx[R] = B2 * (qin[S(ilo - 2, ihi - 2, jlo, jhi)] + qin[S(ilo + 1, ihi + 1, jlo, jhi)]) + B1 * (qin[S(ilo - 1, ihi - 1, jlo, jhi)] + qin[R])

    def colx(a, i, j0, j1):
        return a[i + o : i + o + 1, j0 + o : j1 + o + 1]

    def rowy(a, i0, i1, j):
        return a[i0 + o : i1 + o + 1, j + o : j + o + 1]

    if W:
        q2 = (colx(qin, 0, js1, je1) * colx(m.dxa, 1, js1, je1) + colx(qin, 1, js1, je1) * colx(m.dxa, 0, js1, je1)) / (
            colx(m.dxa, 0, js1, je1) + colx(m.dxa, 1, js1, je1)
        )
        # q2 index: row j -> q2[:, j - js1]
        ew = D.edge_w[js2 + o : je2 + o + 1][None, :, None]
        qout[1 + o : 2 + o, js2 + o : je2 + o + 1] = ew * q2[:, js2 - 1 - js1 : je2 - 1 - js1 + 1] + (1.0 - ew) * q2[:, js2 - js1 : je2 - js1 + 1]
        g_in = colx(m.dxa, 2, jlo, jhi) / colx(m.dxa, 1, jlo, jhi)
        g_ou = colx(m.dxa, -1, jlo, jhi) / colx(m.dxa, 0, jlo, jhi)
        qx[1 + o : 2 + o, jlo + o : jhi + o + 1] = 0.5 * (
            ((2.0 + g_in) * colx(qin, 1, jlo, jhi) - colx(qin, 2, jlo, jhi)) / (1.0 + g_in)
            + ((2.0 + g_ou) * colx(qin, 0, jlo, jhi) - colx(qin, -1, jlo, jhi)) / (1.0 + g_ou)
        )
        qx[2 + o : 3 + o, jlo + o : jhi + o + 1] = (
            3.0 * (g_in * colx(qin, 1, jlo, jhi) + colx(qin, 2, jlo, jhi)) - (g_in * colx(qx, 1, jlo, jhi) + colx(qx, 3, jlo, jhi))
        ) / (2.0 + 2.0 * g_in)
    if E:
        q2 = (colx(qin, npx - 1, js1, je1) * colx(m.dxa, npx, js1, je1) + colx(qin, npx, js1, je1) * colx(m.dxa, npx - 1, js1, je1)) / (
            colx(m.dxa, npx - 1, js1, je1) + colx(m.dxa, npx, js1, je1)
        )
        ee = D.edge_e[js2 + o : je2 + o + 1][None, :, None]
        qout[npx + o : npx + o + 1, js2 + o : je2 + o + 1] = ee * q2[:, js2 - 1 - js1 : je2 - 1 - js1 + 1] + (1.0 - ee) * q2[:, js2 - js1 : je2 - js1 + 1]
        g_in = colx(m.dxa, npx - 2, jlo, jhi) / colx(m.dxa, npx - 1, jlo, jhi)
        g_ou = colx(m.dxa, npx + 1, jlo, jhi) / colx(m.dxa, npx, jlo, jhi)
        qx[npx + o : npx + o + 1, jlo + o : jhi + o + 1] = 0.5 * (
            ((2.0 + g_in) * colx(qin, npx - 1, jlo, jhi) - colx(qin, npx - 2, jlo, jhi)) / (1.0 + g_in)
            + ((2.0 + g_ou) * colx(qin, npx, jlo, jhi) - colx(qin, npx + 1, jlo, jhi)) / (1.0 + g_ou)
        )
        qx[npx - 1 + o : npx + o, jlo + o : jhi + o + 1] = (
            3.0 * (colx(qin, npx - 2, jlo, jhi) + g_in * colx(qin, npx - 1, jlo, jhi)) - (g_in * colx(qx, npx, jlo, jhi) + colx(qx, npx - 2, jlo, jhi))
        ) / (2.0 + 2.0 * g_in)

    # ---- Y interior
    ilo_y = 1 if W else is_ - 2
    ihi_y = npx - 1 if E else ie + 2
    jlo_y = 3 if Sd else js
    jhi_y = npy - 2 if N else je + 1
    R = S(ilo_y, ihi_y, jlo_y, jhi_y)
    qy[R] = B2 * (qin[S(ilo_y, ihi_y, jlo_y - 2, jhi_y - 2)] + qin[S(ilo_y, ihi_y, jlo_y + 1, jhi_y + 1)]) + B1 * (
        qin[S(ilo_y, ihi_y, jlo_y - 1, jhi_y - 1)] + qin[R]
    )
    if Sd:
        q1 = (rowy(qin, is1, ie1, 0) * rowy(m.dya, is1, ie1, 1) + rowy(qin, is1, ie1, 1) * rowy(m.dya, is1, ie1, 0)) / (
            rowy(m.dya, is1, ie1, 0) + rowy(m.dya, is1, ie1, 1)
        )
        es = D.edge_s[is2 + o : ie2 + o + 1][:, None, None]
        qout[is2 + o : ie2 + o + 1, 1 + o : 2 + o] = es * q1[is2 - 1 - is1 : ie2 - 1 - is1 + 1] + (1.0 - es) * q1[is2 - is1 : ie2 - is1 + 1]
        g_in = rowy(m.dya, ilo_y, ihi_y, 2) / rowy(m.dya, ilo_y, ihi_y, 1)
        g_ou = rowy(m.dya, ilo_y, ihi_y, -1) / rowy(m.dya, ilo_y, ihi_y, 0)
        qy[ilo_y + o : ihi_y + o + 1, 1 + o : 2 + o] = 0.5 * (
            ((2.0 + g_in) * rowy(qin, ilo_y, ihi_y, 1) - rowy(qin, ilo_y, ihi_y, 2)) / (1.0 + g_in)
            + ((2.0 + g_ou) * rowy(qin, ilo_y, ihi_y, 0) - rowy(qin, ilo_y, ihi_y, -1)) / (1.0 + g_ou)
        )
        qy[ilo_y + o : ihi_y + o + 1, 2 + o : 3 + o] = (
            3.0 * (g_in * rowy(qin, ilo_y, ihi_y, 1) + rowy(qin, ilo_y, ihi_y, 2)) - (g_in * rowy(qy, ilo_y, ihi_y, 1) + rowy(qy, ilo_y, ihi_y, 3))
        ) / (2.0 + 2.0 * g_in)
    if N:
        q1 = (rowy(qin, is1, ie1, npy - 1) * rowy(m.dya, is1, ie1, npy) + rowy(qin, is1, ie1, npy) * rowy(m.dya, is1, ie1, npy - 1)) / (
            rowy(m.dya, is1, ie1, npy - 1) + rowy(m.dya, is1, ie1, npy)
        )
        en = D.edge_n[is2 + o : ie2 + o + 1][:, None, None]
        qout[is2 + o : ie2 + o + 1, npy + o : npy + o + 1] = en * q1[is2 - 1 - is1 : ie2 - 1 - is1 + 1] + (1.0 - en) * q1[is2 - is1 : ie2 - is1 + 1]
        g_in = rowy(m.dya, ilo_y, ihi_y, npy - 2) / rowy(m.dya, ilo_y, ihi_y, npy - 1)
        g_ou = rowy(m.dya, ilo_y, ihi_y, npy + 1) / rowy(m.dya, ilo_y, ihi_y, npy)
        qy[ilo_y + o : ihi_y + o + 1, npy + o : npy + o + 1] = 0.5 * (
            ((2.0 + g_in) * rowy(qin, ilo_y, ihi_y, npy - 1) - rowy(qin, ilo_y, ihi_y, npy - 2)) / (1.0 + g_in)
            + ((2.0 + g_ou) * rowy(qin, ilo_y, ihi_y, npy) - rowy(qin, ilo_y, ihi_y, npy + 1)) / (1.0 + g_ou)
        )
        qy[ilo_y + o : ihi_y + o + 1, npy - 1 + o : npy + o] = (
            3.0 * (rowy(qin, ilo_y, ihi_y, npy - 2) + g_in * rowy(qin, ilo_y, ihi_y, npy - 1))
            - (g_in * rowy(qy, ilo_y, ihi_y, npy) + rowy(qy, ilo_y, ihi_y, npy - 2))
        ) / (2.0 + 2.0 * g_in)

    # ---- second stage
    ia, ib = (2 if W else is_), (npx - 1 if E else ie + 1)
    ja, jb = (3 if Sd else js), (npy - 2 if N else je + 1)
    R = S(ia, ib, ja, jb)
    qxx[R] = A2 * (qx[S(ia, ib, ja - 2, jb - 2)] + qx[S(ia, ib, ja + 1, jb + 1)]) + A1 * (qx[S(ia, ib, ja - 1, jb - 1)] + qx[R])
    if Sd:
        qxx[ia + o : ib + o + 1, 2 + o : 3 + o] = C1 * (rowy(qx, ia, ib, 1) + rowy(qx, ia, ib, 2)) + C2 * (rowy(qout, ia, ib, 1) + rowy(qxx, ia, ib, 3))
    if N:
        qxx[ia + o : ib + o + 1, npy - 1 + o : npy + o] = C1 * (rowy(qx, ia, ib, npy - 2) + rowy(qx, ia, ib, npy - 1)) + C2 * (
            rowy(qout, ia, ib, npy) + rowy(qxx, ia, ib, npy - 2)
        )
    ja2, jb2 = (2 if Sd else js), (npy - 1 if N else je + 1)
    ia2, ib2 = (3 if W else is_), (npx - 2 if E else ie + 1)
    R = S(ia2, ib2, ja2, jb2)
    qyy[R] = A2 * (qy[S(ia2 - 2, ib2 - 2, ja2, jb2)] + qy[S(ia2 + 1, ib2 + 1, ja2, jb2)]) + A1 * (qy[S(ia2 - 1, ib2 - 1, ja2, jb2)] + qy[R])
    if W:
        qyy[2 + o : 3 + o, ja2 + o : jb2 + o + 1] = C1 * (colx(qy, 1, ja2, jb2) + colx(qy, 2, ja2, jb2)) + C2 * (colx(qout, 1, ja2, jb2) + colx(qyy, 3, ja2, jb2))
    if E:
        qyy[npx - 1 + o : npx + o, ja2 + o : jb2 + o + 1] = C1 * (colx(qy, npx - 2, ja2, jb2) + colx(qy, npx - 1, ja2, jb2)) + C2 * (
            colx(qout, npx, ja2, jb2) + colx(qyy, npx - 2, ja2, jb2)
        )
    R = S(ia, ib, ja2, jb2)
    qout[R] = 0.5 * (qxx[R] + qyy[R])
    if replace:
        Rr = S(is_, ie + 1, js, je + 1)
        qin[Rr] = qout[Rr]
    return qout
