"""ORACLE (test infrastructure) -- D-grid full step ``d_sw``: ``fxadv`` flux preparation,
transport of delp / w / q_con / pt, kinetic energy, divergence damping, vorticity
flux and damping heat  [SURVEY A.3; FV3 sw_core.F90 d_sw; pyFV3 ``d_sw.DGridShallowWaterLagrangianDynamics``,
``fxadv.FiniteVolumeFluxPrep``, ``divergence_damping.DivergenceDamping``; checkpoint
variables REF tests/savepoint/thresholds/fv_dynamics.yaml:76-170; config fields
REF driver/examples/configs/baroclinic_c12.yaml:43-76].
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List

import numpy as np

from .a2b_ord4 import a2b_ord4
from .fvtp2d import del6_vt_flux, fv_tp_2d
from .ppm import xtp_u, ytp_v
from .util import Dom, fill_corners_bgrid, fill_corners_dgrid_vector


# ---------------------------------------------------------------------------
# per-level parameters  [SURVEY A.3.9; pyFV3 d_sw.get_column_namelist]
# ---------------------------------------------------------------------------
def get_column_namelist(cfg, nz: int) -> Dict[str, np.ndarray]:
    col = {}
    for name in ("ke_bg", "d_con", "nord"):
        col[name] = np.full(nz, float(getattr(cfg, name)))
    col["d2_divg"] = np.full(nz, min(0.2, cfg.d2_bg))
    col["nord_v"] = np.full(nz, float(min(2, cfg.nord)))
    col["nord_w"] = col["nord_v"].copy()
    col["nord_t"] = col["nord_v"].copy()
    col["damp_vt"] = np.full(nz, cfg.vtdm4 if cfg.do_vort_damp else 0.0)
    col["damp_w"] = col["damp_vt"].copy()
    col["damp_t"] = col["damp_vt"].copy()

    def set_low(k):
        for name in ("nord", "nord_w", "d_con"):
            col[name][k] = 0
        col["damp_w"][k] = col["d2_divg"][k]

    def lowest(k):
        set_low(k)
        if cfg.do_vort_damp:
            col["nord_v"][k] = 0
            col["damp_vt"][k] = 0.5 * col["d2_divg"][k]

    if nz == 1 or cfg.n_sponge < 0:
        col["d2_divg"][0] = cfg.d2_bg
    else:
        col["d2_divg"][0] = max(0.01, cfg.d2_bg, cfg.d2_bg_k1)
        lowest(0)
        if cfg.d2_bg_k2 > 0.01 and nz > 1:
            col["d2_divg"][1] = max(cfg.d2_bg, cfg.d2_bg_k2)
            lowest(1)
        if cfg.d2_bg_k2 > 0.05 and nz > 2:
            col["d2_divg"][2] = max(cfg.d2_bg, 0.2 * cfg.d2_bg_k2)
            set_low(2)
    return col


def k_groups(col: Dict[str, np.ndarray]) -> List[range]:
    """Maximal k ranges over which every column parameter is constant."""
    nz = len(col["nord"])
    keys = sorted(col)
    groups = []
    k0 = 0
    for k in range(1, nz + 1):
        if k == nz or any(col[n][k] != col[n][k0] for n in keys):
            groups.append(range(k0, k))
            k0 = k
    return groups


# ---------------------------------------------------------------------------
# fxadv  [SURVEY A.3.1]
# ---------------------------------------------------------------------------
def _corner_solve(D, uc, vc, ut, vt, corner):
    """Coupled 2x2 solves next to one cube corner (derived, not tabulated: each target
    uses the general averaging formula with its one not-yet-known neighbour -- the
    partner across the corner -- substituted by the partner's own formula)."""
    o = D.o
    m = D.m
    npx, npy = D.npx, D.npy
    west = corner in ("sw", "nw")
    south = corner in ("sw", "se")
    # ut targets: column next to the edge cell column, the two rows straddling the S/N edge
    it = 2 if west else npx - 1
    pi = 1 if west else npx - 1  # partner vt column (cell column adjacent to the W/E edge)
    jrows = (0, 1) if south else (npy - 1, npy)
    jedge = 1 if south else npy
    new_ut = {}
    for j in jrows:
        # vt neighbours of ut(it, j): (it-1, j), (it, j), (it-1, j+1), (it, j+1)
        nb = [(it - 1, j), (it, j), (it - 1, j + 1), (it, j + 1)]
        pj = [jj for (_, jj) in nb if jj != jedge][0]
        partner = (pi, pj)
        others = [p for p in nb if p != partner]
        # ut neighbours of the partner vt(pi, pj): (pi, pj-1), (pi+1, pj-1), (pi, pj), (pi+1, pj)
        pnb = [(pi, pj - 1), (pi + 1, pj - 1), (pi, pj), (pi + 1, pj)]
        pothers = [p for p in pnb if p != (it, j)]
        cu = m.cosa_u[it + o, j + o]
        cv = m.cosa_v[pi + o, pj + o]
        damp = 1.0 / (1.0 - 0.0625 * cu * cv)
        s_v = sum(vt[a + o, b + o] for a, b in others)
        s_u = sum(ut[a + o, b + o] for a, b in pothers)
        new_ut[(it, j)] = (uc[it + o, j + o] - 0.25 * cu * (s_v + vc[pi + o, pj + o] - 0.25 * cv * s_u)) * damp
    jt = 2 if south else npy - 1
    pj = 1 if south else npy - 1  # partner ut row (cell row adjacent to the S/N edge)
    icols = (0, 1) if west else (npx - 1, npx)
    iedge = 1 if west else npx
    new_vt = {}
    for i in icols:
        nb = [(i, jt - 1), (i + 1, jt - 1), (i, jt), (i + 1, jt)]
        pi_ = [ii for (ii, _) in nb if ii != iedge][0]
        partner = (pi_, pj)
        others = [p for p in nb if p != partner]
        pnb = [(pi_ - 1, pj), (pi_, pj), (pi_ - 1, pj + 1), (pi_, pj + 1)]
        pothers = [p for p in pnb if p != (i, jt)]
        cv = m.cosa_v[i + o, jt + o]
        cu = m.cosa_u[pi_ + o, pj + o]
        damp = 1.0 / (1.0 - 0.0625 * cu * cv)
        s_u = sum(ut[a + o, b + o] for a, b in others)
        s_v = sum(vt[a + o, b + o] for a, b in pothers)
        new_vt[(i, jt)] = (vc[i + o, jt + o] - 0.25 * cv * (s_u + uc[pi_ + o, pj + o] - 0.25 * cu * s_v)) * damp
    for (i, j), val in new_ut.items():
        ut[i + o, j + o] = val
    for (i, j), val in new_vt.items():
        vt[i + o, j + o] = val


def fxadv(D: Dom, uc, vc, crx, cry, xfx, yfx, ut, vt, dt):
    """Contravariant winds, Courant numbers and area fluxes (FiniteVolumeFluxPrep).
    Same argument order as the reference [REF examples/notebooks/functions.py:877-891].
    Returns (ra_x, ra_y)."""
    S = D.sl
    o = D.o
    m = D.m
    is_, ie, js, je, npx, npy = D.is_, D.ie, D.js, D.je, D.npx, D.npy
    isd, ied, jsd, jed = D.isd, D.ied, D.jsd, D.jed
    W, E, Sd, N = D.west, D.east, D.south, D.north

    R = S(is_ - 1, ie + 3, jsd, jed)
    val = (uc[R] - 0.25 * m.cosa_u[R] * (vc[S(is_ - 2, ie + 2, jsd, jed)] + vc[R] + vc[S(is_ - 2, ie + 2, jsd + 1, jed + 1)] + vc[S(is_ - 1, ie + 3, jsd + 1, jed + 1)])) * m.rsin_u[R]
    keep = ut[R].copy()
    ut[R] = val
    # rows on / next to the S and N tile edges are not touched by the general formula
    for flag, rows in ((Sd, (0, 1)), (N, (npy - 1, npy))):
        if flag:
            for j in rows:
                ut[is_ - 1 + o : ie + 3 + o + 1, j + o] = keep[:, j - jsd]
    R = S(isd, ied, js - 1, je + 3)
    val = (vc[R] - 0.25 * m.cosa_v[R] * (uc[S(isd, ied, js - 2, je + 2)] + uc[S(isd + 1, ied + 1, js - 2, je + 2)] + uc[R] + uc[S(isd + 1, ied + 1, js - 1, je + 3)])) * m.rsin_v[R]
    keep = vt[R].copy()
    vt[R] = val
    for flag, j in ((Sd, 1), (N, npy)):
        if flag:
            vt[isd + o : ied + o + 1, j + o] = keep[:, j - (js - 1)]

    def col(a, i, j0=jsd, j1=jed):
        return a[i + o : i + o + 1, j0 + o : j1 + o + 1]

    def row(a, j, i0=isd, i1=ied):
        return a[i0 + o : i1 + o + 1, j + o : j + o + 1]

    for flag, i in ((W, 1), (E, npx)):
        if flag:
            ucol = col(uc, i)
            ut[i + o : i + o + 1, jsd + o : jed + o + 1] = np.where(ucol * dt > 0.0, ucol / col(m.sin_sg3, i - 1), ucol / col(m.sin_sg1, i))
            j0, j1 = max(3, js), min(npy - 2, je + 1)
            if not Sd:
                j0 = js
            if not N:
                j1 = je + 1
            for iv in (i - 1, i):
                vt[iv + o : iv + o + 1, j0 + o : j1 + o + 1] = col(vc, iv, j0, j1) - 0.25 * col(m.cosa_v, iv, j0, j1) * (
                    col(ut, iv, j0 - 1, j1 - 1) + col(ut, iv + 1, j0 - 1, j1 - 1) + col(ut, iv, j0, j1) + col(ut, iv + 1, j0, j1)
                )
    for flag, j in ((Sd, 1), (N, npy)):
        if flag:
            vrow = row(vc, j)
            vt[isd + o : ied + o + 1, j + o : j + o + 1] = np.where(vrow * dt > 0.0, vrow / row(m.sin_sg4, j - 1), vrow / row(m.sin_sg2, j))
            i0, i1 = max(3, is_), min(npx - 2, ie + 1)
            if not W:
                i0 = is_
            if not E:
                i1 = ie + 1
            for ju in (j - 1, j):
                ut[i0 + o : i1 + o + 1, ju + o : ju + o + 1] = row(uc, ju, i0, i1) - 0.25 * row(m.cosa_u, ju, i0, i1) * (
                    row(vt, ju, i0 - 1, i1 - 1) + row(vt, ju, i0, i1) + row(vt, ju + 1, i0 - 1, i1 - 1) + row(vt, ju + 1, i0, i1)
                )
    for corner, has in (("sw", D.sw), ("se", D.se), ("ne", D.ne), ("nw", D.nw)):
        if has:
            _corner_solve(D, uc, vc, ut, vt, corner)

    R = S(is_, ie + 1, jsd, jed)
    Rm = S(is_ - 1, ie, jsd, jed)
    x = dt * ut[R]
    crx[R] = np.where(x > 0.0, x * m.rdxa[Rm], x * m.rdxa[R])
    xfx[R] = np.where(x > 0.0, m.dy[R] * x * m.sin_sg3[Rm], m.dy[R] * x * m.sin_sg1[R])
    R = S(isd, ied, js, je + 1)
    Rm = S(isd, ied, js - 1, je)
    y = dt * vt[R]
    cry[R] = np.where(y > 0.0, y * m.rdya[Rm], y * m.rdya[R])
    yfx[R] = np.where(y > 0.0, m.dx[R] * y * m.sin_sg4[Rm], m.dx[R] * y * m.sin_sg2[R])
    ra_x = np.zeros_like(uc)
    ra_y = np.zeros_like(uc)
    R = S(is_, ie, jsd, jed)
    ra_x[R] = m.area[R] + xfx[R] - xfx[S(is_ + 1, ie + 1, jsd, jed)]
    R = S(isd, ied, js, je)
    ra_y[R] = m.area[R] + yfx[R] - yfx[S(isd, ied, js + 1, je + 1)]
    return ra_x, ra_y


# ---------------------------------------------------------------------------
# divergence damping  [SURVEY A.3.6]
# ---------------------------------------------------------------------------
def divergence_damping(D: Dom, u, v, va, ptc, vort, ua, divg_d, vc, uc, delpc, ke, wk, dt, nord, d2_bg, d4_bg, dddmp):
    """Adds the damping term to ke (corners is..ie+1, js..je+1); leaves the damping
    field in vort and the un-iterated divergence in delpc (FV3 d_sw divergence block)."""
    S = D.sl
    o = D.o
    m = D.m
    is_, ie, js, je, npx, npy = D.is_, D.ie, D.js, D.je, D.npx, D.npy
    W, E, Sd, N = D.west, D.east, D.south, D.north
    da_min_c = D.grid.da_min_c
    Rc = S(is_, ie + 1, js, je + 1)
    if nord == 0:
        R = S(is_ - 1, ie + 1, js, je + 1)
        Rm = S(is_ - 1, ie + 1, js - 1, je)
        ptc[R] = (u[R] - 0.5 * (va[Rm] + va[R]) * m.cosa_v[R]) * m.dyc[R] * m.sina_v[R]
        for flag, j in ((Sd, 1), (N, npy)):
            if flag:
                R1 = S(is_ - 1, ie + 1, j, j)
                R1m = S(is_ - 1, ie + 1, j - 1, j - 1)
                ptc[R1] = np.where(vc[R1] * dt > 0.0, u[R1] * m.dyc[R1] * m.sin_sg4[R1m], u[R1] * m.dyc[R1] * m.sin_sg2[R1])
        R = S(is_, ie + 1, js - 1, je + 1)
        Rm = S(is_ - 1, ie, js - 1, je + 1)
        vort[R] = (v[R] - 0.5 * (ua[Rm] + ua[R]) * m.cosa_u[R]) * m.dxc[R] * m.sina_u[R]
        for flag, i in ((W, 1), (E, npx)):
            if flag:
                R1 = S(i, i, js - 1, je + 1)
                R1m = S(i - 1, i - 1, js - 1, je + 1)
                vort[R1] = np.where(uc[R1] * dt > 0.0, v[R1] * m.dxc[R1] * m.sin_sg3[R1m], v[R1] * m.dxc[R1] * m.sin_sg1[R1])
        delpc[Rc] = vort[S(is_, ie + 1, js - 1, je)] - vort[Rc] + ptc[S(is_ - 1, ie, js, je + 1)] - ptc[Rc]
        if D.sw:
            delpc[1 + o, 1 + o] -= vort[1 + o, 0 + o]
        if D.se:
            delpc[npx + o, 1 + o] -= vort[npx + o, 0 + o]
        if D.ne:
            delpc[npx + o, npy + o] += vort[npx + o, npy + o]
        if D.nw:
            delpc[1 + o, npy + o] += vort[1 + o, npy + o]
        delpc[Rc] = m.rarea_c[Rc] * delpc[Rc]
        damp = da_min_c * np.maximum(d2_bg, np.minimum(0.20, dddmp * np.abs(delpc[Rc] * dt)))
        vort[Rc] = damp * delpc[Rc]
        ke[Rc] += vort[Rc]
        return
    # ---- higher order
    delpc[Rc] = divg_d[Rc]
    any_corner = D.sw or D.se or D.ne or D.nw
    for n in range(1, nord + 1):
        nt = nord - n
        fill_c = (nt != 0) and any_corner
        if fill_c:
            fill_corners_bgrid(D, divg_d, 1)
        R = S(is_ - 1 - nt, ie + 1 + nt, js - nt, je + 1 + nt)
        vc[R] = (divg_d[S(is_ - nt, ie + 2 + nt, js - nt, je + 1 + nt)] - divg_d[R]) * m.divg_u[R]
        if fill_c:
            fill_corners_bgrid(D, divg_d, 2)
        R = S(is_ - nt, ie + 1 + nt, js - 1 - nt, je + 1 + nt)
        uc[R] = (divg_d[S(is_ - nt, ie + 1 + nt, js - nt, je + 2 + nt)] - divg_d[R]) * m.divg_v[R]
        if fill_c:
            fill_corners_dgrid_vector(D, vc, uc, -1.0)
        R = S(is_ - nt, ie + 1 + nt, js - nt, je + 1 + nt)
        divg_d[R] = uc[S(is_ - nt, ie + 1 + nt, js - nt - 1, je + nt)] - uc[R] + vc[S(is_ - nt - 1, ie + nt, js - nt, je + 1 + nt)] - vc[R]
        if D.sw:
            divg_d[1 + o, 1 + o] -= uc[1 + o, 0 + o]
        if D.se:
            divg_d[npx + o, 1 + o] -= uc[npx + o, 0 + o]
        if D.ne:
            divg_d[npx + o, npy + o] += uc[npx + o, npy + o]
        if D.nw:
            divg_d[1 + o, npy + o] += uc[1 + o, npy + o]
        divg_d[R] = divg_d[R] * m.rarea_c[R]
    if dddmp < 1.0e-5:
        vort[...] = 0.0
    else:
        wkb = a2b_ord4(D, wk, replace=False)
        vort[Rc] = np.abs(dt) * np.sqrt(delpc[Rc] ** 2 + wkb[Rc] ** 2)
    dd8 = (da_min_c * d4_bg) ** (nord + 1)
    damp2 = da_min_c * np.maximum(d2_bg, np.minimum(0.20, dddmp * vort[Rc]))
    vort[Rc] = damp2 * delpc[Rc] + dd8 * divg_d[Rc]
    ke[Rc] += vort[Rc]


# ---------------------------------------------------------------------------
# d_sw proper
# ---------------------------------------------------------------------------
@dataclass
class DSWParams:
    """Per-level scalar parameters of one k group."""

    nord: int
    nord_v: int
    nord_w: int
    nord_t: int
    damp_vt: float
    damp_w: float
    damp_t: float
    d2_divg: float
    d_con: float
    ke_bg: float


def d_sw_levels(D: Dom, cfg, p: DSWParams, delpc, delp, pt, u, v, w, uc, vc, ua, va, divgd, mfx, mfy, cx, cy, crx, cry, xfx, yfx, q_con, heat_source, diss_est, dt):
    """d_sw on a block of levels sharing one parameter set (arrays are k-slices, updated in place)."""
    S = D.sl
    o = D.o
    m = D.m
    is_, ie, js, je, npx, npy = D.is_, D.ie, D.js, D.je, D.npx, D.npy
    isd, ied, jsd, jed = D.isd, D.ied, D.jsd, D.jed
    W, E, Sd, N = D.west, D.east, D.south, D.north
    da_min_c = D.grid.da_min_c
    Rc = S(is_, ie, js, je)
    Rcx = S(is_ + 1, ie + 1, js, je)
    Rcy = S(is_, ie, js + 1, je + 1)
    ut = np.zeros_like(delp)
    vt = np.zeros_like(delp)

    ra_x, ra_y = fxadv(D, uc, vc, crx, cry, xfx, yfx, ut, vt, dt)

    # ---- air mass
    fx, fy = fv_tp_2d(D, delp, crx, cry, xfx, yfx, ra_x, ra_y, cfg.hord_dp, nord=p.nord_v, damp_c=p.damp_vt)
    R = S(is_, ie + 1, jsd, jed)
    cx[R] += crx[R]
    R = S(is_, ie + 1, js, je)
    mfx[R] += fx[R]
    R = S(isd, ied, js, je + 1)
    cy[R] += cry[R]
    R = S(is_, ie, js, je + 1)
    mfy[R] += fy[R]

    # ---- vertical velocity
    heat_s = np.zeros_like(delp)
    diss_e = np.zeros_like(delp)
    dw = np.zeros_like(delp)
    if p.damp_w > 1.0e-5:
        dd8 = p.ke_bg * abs(dt)
        damp4 = (p.damp_w * da_min_c) ** (p.nord_w + 1)
        fx2, fy2, _ = del6_vt_flux(D, p.nord_w, damp4, w)
        dw[Rc] = (fx2[Rc] - fx2[Rcx] + fy2[Rc] - fy2[Rcy]) * m.rarea[Rc]
        heat_s[Rc] = dd8 - dw[Rc] * (w[Rc] + 0.5 * dw[Rc])
        diss_e[Rc] = heat_s[Rc]
    gx, gy = fv_tp_2d(D, w, crx, cry, xfx, yfx, ra_x, ra_y, cfg.hord_vt, mfx=fx, mfy=fy)
    w[Rc] = delp[Rc] * w[Rc] + (gx[Rc] - gx[Rcx] + gy[Rc] - gy[Rcy]) * m.rarea[Rc]

    # ---- condensate and potential temperature
    gx, gy = fv_tp_2d(D, q_con, crx, cry, xfx, yfx, ra_x, ra_y, cfg.hord_dp, mfx=fx, mfy=fy, mass=delp, nord=p.nord_t, damp_c=p.damp_t)
    q_con[Rc] = delp[Rc] * q_con[Rc] + (gx[Rc] - gx[Rcx] + gy[Rc] - gy[Rcy]) * m.rarea[Rc]
    gx, gy = fv_tp_2d(D, pt, crx, cry, xfx, yfx, ra_x, ra_y, cfg.hord_tm, mfx=fx, mfy=fy, mass=delp, nord=p.nord_v, damp_c=p.damp_vt)
    pt[Rc] = pt[Rc] * delp[Rc] + (gx[Rc] - gx[Rcx] + gy[Rc] - gy[Rcy]) * m.rarea[Rc]
    delp[Rc] = delp[Rc] + (fx[Rc] - fx[Rcx] + fy[Rc] - fy[Rcy]) * m.rarea[Rc]
    pt[Rc] = pt[Rc] / delp[Rc]
    w[Rc] = w[Rc] / delp[Rc]
    if p.damp_w > 1.0e-5:
        w[Rc] = w[Rc] + dw[Rc]
    q_con[Rc] = q_con[Rc] / delp[Rc]

    # ---- kinetic energy on corners
    dt5 = 0.5 * dt
    dt4 = 0.25 * dt
    is2 = 2 if W else is_
    ie1 = npx - 1 if E else ie + 1
    js2 = 2 if Sd else js
    je1 = npy - 1 if N else je + 1
    vb = np.zeros_like(delp)
    ub = np.zeros_like(delp)
    ke = np.zeros_like(delp)
    for flag, j in ((Sd, 1), (N, npy)):
        if flag:
            R = S(is_, ie + 1, j, j)
            vb[R] = dt5 * (vt[S(is_ - 1, ie, j, j)] + vt[R])
    R = S(is2, ie1, js2, je1)
    vb[R] = dt5 * (vc[S(is2 - 1, ie1 - 1, js2, je1)] + vc[R] - (uc[S(is2, ie1, js2 - 1, je1 - 1)] + uc[R]) * m.cosa[R]) * m.rsina[R]
    for flag, i in ((W, 1), (E, npx)):
        if flag:
            R = S(i, i, js2, je1)
            vb[R] = dt4 * (-vt[S(i - 2, i - 2, js2, je1)] + 3.0 * (vt[S(i - 1, i - 1, js2, je1)] + vt[R]) - vt[S(i + 1, i + 1, js2, je1)])
    ubk = ytp_v(D, vb, v, cfg.hord_mt)
    Rk = S(is_, ie + 1, js, je + 1)
    ke[Rk] = vb[Rk] * ubk[Rk]

    for flag, i in ((W, 1), (E, npx)):
        if flag:
            R = S(i, i, js, je + 1)
            ub[R] = dt5 * (ut[S(i, i, js - 1, je)] + ut[R])
    R = S(is2, ie1, js, je + 1)
    ub[R] = dt5 * (uc[S(is2, ie1, js - 1, je)] + uc[R] - (vc[S(is2 - 1, ie1 - 1, js, je + 1)] + vc[R]) * m.cosa[R]) * m.rsina[R]
    for flag, j in ((Sd, 1), (N, npy)):
        if flag:
            R = S(is2, ie1, j, j)
            ub[R] = dt4 * (-ut[S(is2, ie1, j - 2, j - 2)] + 3.0 * (ut[S(is2, ie1, j - 1, j - 1)] + ut[R]) - ut[S(is2, ie1, j + 1, j + 1)])
    vbk = xtp_u(D, ub, u, cfg.hord_mt)
    ke[Rk] = 0.5 * (ke[Rk] + ub[Rk] * vbk[Rk])

    dt6 = dt / 6.0

    def A(a, i, j):
        return a[i + o, j + o]

    if D.sw:
        ke[1 + o, 1 + o] = dt6 * ((A(ut, 1, 1) + A(ut, 1, 0)) * A(u, 1, 1) + (A(vt, 1, 1) + A(vt, 0, 1)) * A(v, 1, 1) + (A(ut, 1, 1) + A(vt, 1, 1)) * A(u, 0, 1))
    if D.se:
        i = npx
        ke[i + o, 1 + o] = dt6 * ((A(ut, i, 1) + A(ut, i, 0)) * A(u, i - 1, 1) + (A(vt, i, 1) + A(vt, i - 1, 1)) * A(v, i, 1) + (A(ut, i, 1) - A(vt, i - 1, 1)) * A(u, i, 1))
    if D.ne:
        i, j = npx, npy
        ke[i + o, j + o] = dt6 * (
            (A(ut, i, j) + A(ut, i, j - 1)) * A(u, i - 1, j) + (A(vt, i, j) + A(vt, i - 1, j)) * A(v, i, j - 1) + (A(ut, i, j - 1) + A(vt, i - 1, j)) * A(u, i, j)
        )
    if D.nw:
        j = npy
        ke[1 + o, j + o] = dt6 * ((A(ut, 1, j) + A(ut, 1, j - 1)) * A(u, 1, j) + (A(vt, 1, j) + A(vt, 0, j)) * A(v, 1, j - 1) + (A(ut, 1, j - 1) - A(vt, 1, j)) * A(u, 0, j))

    # ---- relative vorticity (cell mean)
    R = S(isd, ied, jsd, jed + 1)
    vt[R] = u[R] * m.dx[R]
    R = S(isd, ied + 1, jsd, jed)
    ut[R] = v[R] * m.dy[R]
    wk = np.zeros_like(delp)
    R = S(isd, ied, jsd, jed)
    wk[R] = m.rarea[R] * (vt[R] - vt[S(isd, ied, jsd + 1, jed + 1)] - ut[R] + ut[S(isd + 1, ied + 1, jsd, jed)])

    # ---- divergence damping (ptc / delpc / vort are work arrays)
    ptc = np.zeros_like(delp)
    vort = np.zeros_like(delp)
    divergence_damping(D, u, v, va, ptc, vort, ua, divgd, vc, uc, delpc, ke, wk, dt, p.nord, p.d2_divg, cfg.d4_bg, cfg.dddmp)

    if p.d_con > 1.0e-5:
        R = S(is_, ie, js, je + 1)
        ub[R] = vort[R] - vort[S(is_ + 1, ie + 1, js, je + 1)]
        R = S(is_, ie + 1, js, je)
        vb[R] = vort[R] - vort[S(is_, ie + 1, js + 1, je + 1)]

    # ---- vorticity transport and wind update
    R = S(isd, ied, jsd, jed)
    vort[R] = wk[R] + m.f0[R]
    fx, fy = fv_tp_2d(D, vort, crx, cry, xfx, yfx, ra_x, ra_y, cfg.hord_vt)
    R = S(is_, ie, js, je + 1)
    u[R] = vt[R] + ke[R] - ke[S(is_ + 1, ie + 1, js, je + 1)] + fy[R]
    R = S(is_, ie + 1, js, je)
    v[R] = ut[R] + ke[R] - ke[S(is_, ie + 1, js + 1, je + 1)] - fx[R]

    if p.damp_vt > 1.0e-5:
        damp4 = (p.damp_vt * da_min_c) ** (p.nord_v + 1)
        ut, vt, _ = del6_vt_flux(D, p.nord_v, damp4, wk)

    if p.d_con > 1.0e-5:
        Ru = S(is_, ie, js, je + 1)
        Rv = S(is_, ie + 1, js, je)
        fyh = np.zeros_like(delp)
        gyh = np.zeros_like(delp)
        fxh = np.zeros_like(delp)
        gxh = np.zeros_like(delp)
        ub[Ru] = (ub[Ru] + vt[Ru]) * m.rdx[Ru]
        fyh[Ru] = u[Ru] * m.rdx[Ru]
        gyh[Ru] = fyh[Ru] * ub[Ru]
        vb[Rv] = (vb[Rv] - ut[Rv]) * m.rdy[Rv]
        fxh[Rv] = v[Rv] * m.rdy[Rv]
        gxh[Rv] = fxh[Rv] * vb[Rv]
        u2 = fyh[Rc] + fyh[Rcy]
        du2 = ub[Rc] + ub[Rcy]
        v2 = fxh[Rc] + fxh[Rcx]
        dv2 = vb[Rc] + vb[Rcx]
        heat_s[Rc] = delp[Rc] * (
            heat_s[Rc]
            - 0.25
            * p.d_con
            * m.rsin2[Rc]
            * (
                (ub[Rc] ** 2 + ub[Rcy] ** 2 + vb[Rc] ** 2 + vb[Rcx] ** 2)
                + 2.0 * (gyh[Rc] + gyh[Rcy] + gxh[Rc] + gxh[Rcx])
                - m.cosa_s[Rc] * (u2 * dv2 + v2 * du2 + du2 * dv2)
            )
        )
    if cfg.d_con > 1.0e-5:
        heat_source[Rc] += heat_s[Rc]
    if p.damp_vt > 1.0e-5:
        R = S(is_, ie, js, je + 1)
        u[R] += vt[R]
        R = S(is_, ie + 1, js, je)
        v[R] -= ut[R]


def d_sw(D: Dom, cfg, col, delpc, delp, pt, u, v, w, uc, vc, ua, va, divgd, mfx, mfy, cx, cy, crx, cry, xfx, yfx, q_con, zh, heat_source, diss_est, dt):
    """Full-step D-grid update; argument order of the reference operator
    (``DGridShallowWaterLagrangianDynamics.__call__``; zh is carried for signature parity only)."""
    nz = D.nz
    for g in k_groups(col):
        k0 = g.start
        p = DSWParams(
            nord=int(col["nord"][k0]),
            nord_v=int(col["nord_v"][k0]),
            nord_w=int(col["nord_w"][k0]),
            nord_t=int(col["nord_t"][k0]),
            damp_vt=float(col["damp_vt"][k0]),
            damp_w=float(col["damp_w"][k0]),
            damp_t=float(col["damp_t"][k0]),
            d2_divg=float(col["d2_divg"][k0]),
            d_con=float(col["d_con"][k0]),
            ke_bg=float(col["ke_bg"][k0]),
        )
        ks = slice(g.start, g.stop)
        args = [a[:, :, ks] for a in (delpc, delp, pt, u, v, w, uc, vc, ua, va, divgd, mfx, mfy, cx, cy, crx, cry, xfx, yfx, q_con, heat_source, diss_est)]
        d_sw_levels(D, cfg, p, *args, dt)
