"""ORACLE (test infrastructure) -- 1-D PPM flux operators, hord/iord 5-6 (unlimited
PPM with the ``smt5`` linear-scheme fallback)  [SURVEY A.4.2; FV3 tp_core xppm/yppm,
sw_core xtp_u/ytp_v; pyFV3 ``xppm.py`` / ``yppm.py`` / ``xtp_u.py`` / ``ytp_v.py``
as used by every reference config: hord_* = 6, REF driver/examples/configs/baroclinic_c12.yaml:57-61].
The y variants are the x variants on the transposed rank.
"""
from __future__ import annotations

import numpy as np

from .util import Dom, tr, alt

P1 = 7.0 / 12.0
P2 = -1.0 / 12.0
C1 = -2.0 / 14.0
C2 = 11.0 / 14.0
C3 = 5.0 / 14.0


def _col(D, a, i, j0, j1):
    o = D.o
    return a[i + o : i + o + 1, j0 + o : j1 + o + 1]


def _edge_mean(q_m1, q_0, q_p1, q_p2, d_m1, d_0, d_p1, d_p2):
    """Two-sided metric-weighted extrapolation to the tile edge between cells 0 and +1."""
    return 0.5 * (
        ((2.0 * d_0 + d_m1) * q_0 - d_0 * q_m1) / (d_m1 + d_0)
        + ((2.0 * d_p1 + d_p2) * q_p1 - d_p1 * q_p2) / (d_p1 + d_p2)
    )


def compute_al(D: Dom, q, dm, j0, j1):
    """Edge values al(i) (west face of cell i) on rows j0..j1, incl. tile-edge formulas."""
    S = D.sl
    npx = D.npx
    al = np.zeros_like(q)
    i_lo = 3 if D.west else D.is_ - 1
    i_hi = npx - 2 if D.east else D.ie + 2
    if i_hi >= i_lo:
        al[S(i_lo, i_hi, j0, j1)] = P1 * (q[S(i_lo - 1, i_hi - 1, j0, j1)] + q[S(i_lo, i_hi, j0, j1)]) + P2 * (
            q[S(i_lo - 2, i_hi - 2, j0, j1)] + q[S(i_lo + 1, i_hi + 1, j0, j1)]
        )

    def Q(i):
        return _col(D, q, i, j0, j1)

    def M(i):
        return _col(D, dm, i, j0, j1)

    def setal(i, v):
        al[i + D.o : i + D.o + 1, j0 + D.o : j1 + D.o + 1] = v

    if D.west:
        setal(0, C1 * Q(-2) + C2 * Q(-1) + C3 * Q(0))
        setal(1, _edge_mean(Q(-1), Q(0), Q(1), Q(2), M(-1), M(0), M(1), M(2)))
        setal(2, C3 * Q(1) + C2 * Q(2) + C1 * Q(3))
    if D.east:
        setal(npx - 1, C1 * Q(npx - 3) + C2 * Q(npx - 2) + C3 * Q(npx - 1))
        setal(npx, _edge_mean(Q(npx - 2), Q(npx - 1), Q(npx), Q(npx + 1), M(npx - 2), M(npx - 1), M(npx), M(npx + 1)))
        setal(npx + 1, C3 * Q(npx) + C2 * Q(npx + 1) + C1 * Q(npx + 2))
    return al


def _flux_from_blbr(D, q, c, bl, br, j0, j1, mord, cfl_scale=None):
    """flux(i) for i = is..ie+1 given bl/br on is-1..ie+1."""
    S = D.sl
    b0 = bl + br
    if mord == 5:
        smt5 = (bl * br) < 0.0
    else:
        # (FV3_ALT=smt5_lim_fac: tp_core.F90 writes abs(lim_fac * b0) with the namelist default lim_fac = 1 -- DESIGN §2, uncertain restatement 3)
        smt5 = ((1.0 if alt("smt5_lim_fac") else 3.0) * np.abs(b0)) < np.abs(bl - br)
    R0 = S(D.is_, D.ie + 1, j0, j1)
    Rm = S(D.is_ - 1, D.ie, j0, j1)
    cc = c[R0]
    if cfl_scale is None:
        cfl = cc
    else:
        cfl = np.where(cc > 0.0, cc * cfl_scale[Rm], cc * cfl_scale[R0])
    fx1 = np.where(cc > 0.0, (1.0 - cfl) * (br[Rm] - cfl * b0[Rm]), (1.0 + cfl) * (bl[R0] + cfl * b0[R0]))
    flux = np.where(cc > 0.0, q[Rm], q[R0])
    flux = np.where(smt5[Rm] | smt5[R0], flux + fx1, flux)
    out = np.zeros_like(q)
    out[R0] = flux
    return out


S11, S14, S15, R3 = 11.0 / 14.0, 4.0 / 7.0, 3.0 / 14.0, 1.0 / 3.0


def _pert_ppm_full(bl, br):
    """pert_ppm(..., iv = 1): the full monotonicity constraint on the (bl, br) pairs of a few cells (tp_core.F90)."""
    da1 = bl - br
    da2 = da1 * da1
    a6da = 3.0 * (bl + br) * da1
    opposite = bl * br < 0.0
    nbr = np.where(opposite & (a6da < -da2), -2.0 * bl, br)
    nbl = np.where(opposite & ~(a6da < -da2) & (a6da > da2), -2.0 * br, bl)
    return np.where(opposite, nbl, 0.0), np.where(opposite, nbr, 0.0)


def xppm8(D: Dom, q, c, j0, j1):
    """iord = 8 (hord_tr of the reference configs [REF driver/examples/configs/baroclinic_c12.yaml:60]): PPM with Lin's fast
    monotone constraint -- monotonized slopes dm, edge values from them, bl / br limited to 2 |dm|; the flux always carries
    the sub-grid correction (no smt5 switch).  Tile edges: one-sided bl / br for the three cells either side + pert_ppm.
    Restated from GFDL_atmos_cubed_sphere tp_core.F90 (xppm, iord >= 8 branch); PARITY UNPINNED."""
    S, o = D.sl, D.o
    npx = D.npx
    dxa = D.m.dxa
    is_, ie = D.is_, D.ie
    dm = np.zeros_like(q)
    R, Rm, Rp = S(is_ - 2, ie + 2, j0, j1), S(is_ - 3, ie + 1, j0, j1), S(is_ - 1, ie + 3, j0, j1)
    xt = 0.25 * (q[Rp] - q[Rm])
    hi = np.maximum(np.maximum(q[Rm], q[R]), q[Rp]) - q[R]
    lo = q[R] - np.minimum(np.minimum(q[Rm], q[R]), q[Rp])
    dm[R] = np.copysign(np.minimum(np.minimum(np.abs(xt), hi), lo), xt)
    is1 = max(3, is_ - 1) if D.west else is_ - 1
    ie1 = min(npx - 3, ie + 1) if D.east else ie + 1
    al = np.zeros_like(q)
    R, Rm = S(is1, ie1 + 1, j0, j1), S(is1 - 1, ie1, j0, j1)
    al[R] = 0.5 * (q[Rm] + q[R]) + R3 * (dm[Rm] - dm[R])
    bl, br = np.zeros_like(q), np.zeros_like(q)
    R, Rp = S(is1, ie1, j0, j1), S(is1 + 1, ie1 + 1, j0, j1)
    x2 = 2.0 * dm[R]
    bl[R] = -np.copysign(np.minimum(np.abs(x2), np.abs(al[R] - q[R])), x2)
    br[R] = np.copysign(np.minimum(np.abs(x2), np.abs(al[Rp] - q[R])), x2)

    def Q(i):
        return _col(D, q, i, j0, j1)

    def M(i):
        return _col(D, dxa, i, j0, j1)

    def DM(i):
        return _col(D, dm, i, j0, j1)

    def AL(i):
        return _col(D, al, i, j0, j1)

    def put(a, i, v):
        a[i + o : i + o + 1, j0 + o : j1 + o + 1] = v

    def get(a, i0, i1):
        return a[i0 + o : i1 + o + 1, j0 + o : j1 + o + 1]

    if D.west:
        put(br, 2, AL(3) - Q(2))
        xt = _edge_mean(Q(-1), Q(0), Q(1), Q(2), M(-1), M(0), M(1), M(2))
        # iord >= 8: the two-sided tile-edge value stays inside the range of the four cells around the edge
        # (tp_core.F90 xppm, "xt = max(xt, min(q1(-1), q1(0), q1(1), q1(2))); xt = min(xt, max(...))"; pyFV3 xppm.xt_dxa_edge_0 with xt_minmax)
        xt = np.minimum(np.maximum(xt, np.minimum(np.minimum(Q(-1), Q(0)), np.minimum(Q(1), Q(2)))), np.maximum(np.maximum(Q(-1), Q(0)), np.maximum(Q(1), Q(2))))
        put(bl, 1, xt - Q(1))
        put(br, 0, xt - Q(0))
        put(bl, 0, S14 * DM(-1) + S11 * (Q(-1) - Q(0)))
        xt = S15 * Q(1) + S11 * Q(2) - S14 * DM(2)
        put(br, 1, xt - Q(1))
        put(bl, 2, xt - Q(2))
        nbl, nbr = _pert_ppm_full(get(bl, 0, 2), get(br, 0, 2))
        bl[0 + o : 3 + o, j0 + o : j1 + o + 1], br[0 + o : 3 + o, j0 + o : j1 + o + 1] = nbl, nbr
    if D.east:
        put(bl, npx - 2, AL(npx - 2) - Q(npx - 2))
        xt = _edge_mean(Q(npx - 2), Q(npx - 1), Q(npx), Q(npx + 1), M(npx - 2), M(npx - 1), M(npx), M(npx + 1))
        xt = np.minimum(np.maximum(xt, np.minimum(np.minimum(Q(npx - 2), Q(npx - 1)), np.minimum(Q(npx), Q(npx + 1)))),
                        np.maximum(np.maximum(Q(npx - 2), Q(npx - 1)), np.maximum(Q(npx), Q(npx + 1))))
        put(br, npx - 1, xt - Q(npx - 1))
        put(bl, npx, xt - Q(npx))
        put(br, npx, S11 * (Q(npx + 1) - Q(npx)) - S14 * DM(npx + 1))
        xt = S15 * Q(npx - 1) + S11 * Q(npx - 2) + S14 * DM(npx - 2)
        put(br, npx - 2, xt - Q(npx - 2))
        put(bl, npx - 1, xt - Q(npx - 1))
        nbl, nbr = _pert_ppm_full(get(bl, npx - 2, npx), get(br, npx - 2, npx))
        bl[npx - 2 + o : npx + 1 + o, j0 + o : j1 + o + 1], br[npx - 2 + o : npx + 1 + o, j0 + o : j1 + o + 1] = nbl, nbr
    R0, Rm = S(is_, ie + 1, j0, j1), S(is_ - 1, ie, j0, j1)
    cc = c[R0]
    flux = np.where(cc > 0.0, q[Rm] + (1.0 - cc) * (br[Rm] - cc * (bl[Rm] + br[Rm])), q[R0] + (1.0 + cc) * (bl[R0] + cc * (bl[R0] + br[R0])))
    out = np.zeros_like(q)
    out[R0] = flux
    return out


def xppm(D: Dom, q, c, j0, j1, iord=6):
    """Flux-form PPM value at x faces i = is..ie+1 on rows j0..j1 (Courant number c)."""
    if iord == 8:
        return xppm8(D, q, c, j0, j1)
    if iord not in (5, 6):
        raise NotImplementedError("oracle restates hord 5 / 6 / 8 (the reference configs use 6 and, for tracers, 8)")
    S = D.sl
    al = compute_al(D, q, D.m.dxa, j0, j1)
    bl = np.zeros_like(q)
    br = np.zeros_like(q)
    R = S(D.is_ - 1, D.ie + 1, j0, j1)
    Rp = S(D.is_, D.ie + 2, j0, j1)
    bl[R] = al[R] - q[R]
    br[R] = al[Rp] - q[R]
    return _flux_from_blbr(D, q, c, bl, br, j0, j1, iord)


def yppm(D: Dom, q, c, i0, i1, jord=6):
    return tr(xppm(D.T, tr(q), tr(c), i0, i1, jord))


def xtp_u(D: Dom, c, u, iord=6):
    """Advective-form PPM of the D-grid wind u along x (FV3 xtp_u): value of u carried
    across corner (i, j) by the corner wind c, i = is..ie+1, j = js..je+1."""
    if iord not in (5, 6):
        raise NotImplementedError
    S = D.sl
    o = D.o
    npx, npy = D.npx, D.npy
    j0, j1 = D.js, D.je + 1
    dx = D.m.dx
    al = compute_al(D, u, dx, j0, j1)
    bl = np.zeros_like(u)
    br = np.zeros_like(u)
    R = S(D.is_ - 1, D.ie + 1, j0, j1)
    Rp = S(D.is_, D.ie + 2, j0, j1)
    bl[R] = al[R] - u[R]
    br[R] = al[Rp] - u[R]
    # at the tile's own corners the edge-row values are not defined: zero the slopes
    rows = []
    if D.south:
        rows.append(1)
    if D.north:
        rows.append(npy)
    for j in rows:
        if D.west:
            bl[0 + o : 2 + o, j + o] = 0.0
            br[0 + o : 2 + o, j + o] = 0.0
        if D.east:
            bl[npx - 1 + o : npx + 1 + o, j + o] = 0.0
            br[npx - 1 + o : npx + 1 + o, j + o] = 0.0
    return _flux_from_blbr(D, u, c, bl, br, j0, j1, iord, cfl_scale=D.m.rdx)


def ytp_v(D: Dom, c, v, jord=6):
    return tr(xtp_u(D.T, tr(c), tr(v), jord))
