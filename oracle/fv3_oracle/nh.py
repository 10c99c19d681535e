"""ORACLE (test infrastructure) -- non-hydrostatic column and pressure-gradient operators:
``update_dz_c``, ``update_dz_d``, ``Riem_Solver_c``, ``Riem_Solver3`` (+ ``SIM1_solver``),
``p_grad_c``, ``nh_p_grad``, ``pk3_halo``, ``pe_halo`` (edge_pe), ``Ray_fast``,
``del2_cubed`` and the diffusive-heating application  [SURVEY A.5-A.12; FV3
nh_utils.F90, nh_core.F90, dyn_core.F90; pyFV3 ``updatedzc / updatedzd / riem_solver_c /
riem_solver3 / sim1_solver / nh_p_grad / pk3_halo / ray_fast / del2cubed /
temperature_adjust``].  Config anchors: a_imp=1 (SIM1), p_fac, rf_cutoff, tau, delt_max
[REF driver/examples/configs/baroclinic_c12.yaml:43,53,72-75].
"""
from __future__ import annotations

import numpy as np

from .a2b_ord4 import a2b_ord4
from .fvtp2d import del6_vt_flux, fv_tp_2d
from .util import Dom, alt, copy_corners, fill_4corners


# ---------------------------------------------------------------------------
# update_dz_c  [SURVEY A.5]
# ---------------------------------------------------------------------------
def update_dz_c(D: Dom, dp_ref, zs, ut, vt, gz, ws, dt):
    """(dp_ref, zs, ut, vt, gz, ws, dt) as the reference operator; gz (nz+1 levels, metres) and ws in place."""
    S = D.sl
    m = D.m
    nz = D.nz
    is_, ie, js, je = D.is_, D.ie, D.js, D.je
    dz_min = D.c.DZ_MIN
    top_ratio = dp_ref[0] / (dp_ref[0] + dp_ref[1])
    bot_ratio = dp_ref[nz - 1] / (dp_ref[nz - 2] + dp_ref[nz - 1])
    Rx = S(is_ - 1, ie + 2, js - 1, je + 1)
    Ry = S(is_ - 1, ie + 1, js - 1, je + 2)
    xfx = np.zeros((gz.shape[0], gz.shape[1], nz + 1))
    yfx = np.zeros_like(xfx)
    for a, src, R in ((xfx, ut, Rx), (yfx, vt, Ry)):
        s = src[R]
        a[R + (slice(0, 1),)] = s[:, :, 0:1] + (s[:, :, 0:1] - s[:, :, 1:2]) * top_ratio
        a[R + (slice(nz, nz + 1),)] = s[:, :, nz - 1 : nz] + (s[:, :, nz - 1 : nz] - s[:, :, nz - 2 : nz - 1]) * bot_ratio
        int_ratio = 1.0 / (dp_ref[:-1] + dp_ref[1:])
        a[R + (slice(1, nz),)] = (dp_ref[1:] * s[:, :, : nz - 1] + dp_ref[:-1] * s[:, :, 1:nz]) * int_ratio
    gz2 = gz.copy()
    fill_4corners(D, gz2, 1)
    fx = np.zeros_like(gz)
    fy = np.zeros_like(gz)
    Rm = S(is_ - 2, ie + 1, js - 1, je + 1)
    fx[Rx] = xfx[Rx] * np.where(xfx[Rx] > 0.0, gz2[Rm], gz2[Rx])
    fill_4corners(D, gz2, 2)
    Rm = S(is_ - 1, ie + 1, js - 2, je + 1)
    fy[Ry] = yfx[Ry] * np.where(yfx[Ry] > 0.0, gz2[Rm], gz2[Ry])
    R = S(is_ - 1, ie + 1, js - 1, je + 1)
    Rxp = S(is_, ie + 2, js - 1, je + 1)
    Ryp = S(is_ - 1, ie + 1, js, je + 2)
    gz[R] = (gz2[R] * m.area[R] + fx[R] - fx[Rxp] + fy[R] - fy[Ryp]) / (m.area[R] + xfx[R] - xfx[Rxp] + yfx[R] - yfx[Ryp])
    ws[R] = (zs[R] - gz[R + (slice(nz, nz + 1),)]) / dt
    g = gz[R]
    for k in range(nz - 1, -1, -1):
        g[:, :, k] = np.maximum(g[:, :, k], g[:, :, k + 1] + dz_min)
    gz[R] = g


# ---------------------------------------------------------------------------
# update_dz_d  [SURVEY A.8]
# ---------------------------------------------------------------------------
def edge_profile(q, dp0):
    """Layer means -> interface values by the cubic-spline-like profile of FV3 ``edge_profile``
    (limiter 0).  q: [..., nz] -> [..., nz+1]."""
    nz = q.shape[-1]
    qe = np.zeros(q.shape[:-1] + (nz + 1,))
    gam = np.zeros(nz + 1)
    g0 = dp0[1] / dp0[0]
    xt1 = 2.0 * g0 * (g0 + 1.0)
    bet = g0 * (g0 + 0.5)
    qe[..., 0] = (xt1 * q[..., 0] + q[..., 1]) / bet
    gam[0] = (1.0 + g0 * (g0 + 1.5)) / bet
    gk = g0
    for k in range(1, nz):
        gk = dp0[k - 1] / dp0[k]
        bet = 2.0 + 2.0 * gk - gam[k - 1]
        qe[..., k] = (3.0 * (q[..., k - 1] + gk * q[..., k]) - qe[..., k - 1]) / bet
        gam[k] = gk / bet
    a_bot = 1.0 + gk * (gk + 1.5)
    xt1 = 2.0 * gk * (gk + 1.0)
    xt2 = gk * (gk + 0.5) - a_bot * gam[nz - 1]
    qe[..., nz] = (xt1 * q[..., nz - 1] + q[..., nz - 2] - a_bot * qe[..., nz - 1]) / xt2
    for k in range(nz - 1, -1, -1):
        qe[..., k] = qe[..., k] - gam[k] * qe[..., k + 1]
    return qe


def update_dz_d(D: Dom, cfg, col, dp_ref, zs, zh, crx, cry, xfx, yfx, ws, dt):
    """zh (nz+1 interfaces) advected by the interface-interpolated D-grid fluxes; zh, ws in place.

    The damping coefficient handed to del6_vt_flux is the raw ``damp_vt`` column value,
    as in the Fortran/pyFV3 call (see DESIGN.md "uncertain restatements")."""
    S = D.sl
    m = D.m
    nz = D.nz
    is_, ie, js, je, isd, ied, jsd, jed = D.is_, D.ie, D.js, D.je, D.isd, D.ied, D.jsd, D.jed
    dz_min = D.c.DZ_MIN
    damp = np.append(col["damp_vt"], col["damp_vt"][-1])
    ndif = np.append(col["nord_v"], col["nord_v"][-1]).astype(int)


    damp_on = damp.copy()  # (the switch "is this interface damped" is the raw coefficient in both forms)
    if alt("dz_damp_scaled"):  # FV3_ALT=dz_damp_scaled: the coefficient d_sw's vorticity damping uses
        damp = (damp * D.grid.da_min_c) ** (ndif + 1)
    shp = zh.shape[:2] + (nz + 1,)
    crx_adv = np.zeros(shp)
    xfx_adv = np.zeros(shp)
    cry_adv = np.zeros(shp)
    yfx_adv = np.zeros(shp)
    Rx = S(is_, ie + 1, jsd, jed)
    Ry = S(isd, ied, js, je + 1)
    crx_adv[Rx] = edge_profile(crx[Rx][:, :, :nz], dp_ref)
    xfx_adv[Rx] = edge_profile(xfx[Rx][:, :, :nz], dp_ref)
    cry_adv[Ry] = edge_profile(cry[Ry][:, :, :nz], dp_ref)
    yfx_adv[Ry] = edge_profile(yfx[Ry][:, :, :nz], dp_ref)
    ra_x = np.zeros(shp)
    ra_y = np.zeros(shp)
    R = S(is_, ie, jsd, jed)
    ra_x[R] = m.area[R] + xfx_adv[R] - xfx_adv[S(is_ + 1, ie + 1, jsd, jed)]
    R = S(isd, ied, js, je)
    ra_y[R] = m.area[R] + yfx_adv[R] - yfx_adv[S(isd, ied, js + 1, je + 1)]
    Rc = S(is_, ie, js, je)
    Rcx = S(is_ + 1, ie + 1, js, je)
    Rcy = S(is_, ie, js + 1, je + 1)
    # group interfaces by (ndif, damp) so each group is one vectorised call
    k0 = 0
    for k in range(1, nz + 2):
        if k == nz + 1 or damp[k] != damp[k0] or ndif[k] != ndif[k0] or damp_on[k] != damp_on[k0]:
            ks = slice(k0, k)
            z2 = zh[:, :, ks].copy()
            fx, fy = fv_tp_2d(D, z2, crx_adv[:, :, ks], cry_adv[:, :, ks], xfx_adv[:, :, ks], yfx_adv[:, :, ks], ra_x[:, :, ks], ra_y[:, :, ks], cfg.hord_tm)
            new = (z2[Rc] * m.area[Rc] + fx[Rc] - fx[Rcx] + fy[Rc] - fy[Rcy]) / (ra_x[:, :, ks][Rc] + ra_y[:, :, ks][Rc] - m.area[Rc])
            if damp_on[k0] > 1.0e-5:
                fx2, fy2, _ = del6_vt_flux(D, int(ndif[k0]), float(damp[k0]), z2)
                new = new + (fx2[Rc] - fx2[Rcx] + fy2[Rc] - fy2[Rcy]) * m.rarea[Rc]
            zh[:, :, ks][Rc] = new
            k0 = k
    ws[Rc] = (zs[Rc] - zh[Rc + (slice(nz, nz + 1),)]) / dt
    g = zh[Rc]
    for k in range(nz - 1, -1, -1):
        g[:, :, k] = np.maximum(g[:, :, k], g[:, :, k + 1] + dz_min)
    zh[Rc] = g


# ---------------------------------------------------------------------------
# semi-implicit solver  [SURVEY A.6]
# ---------------------------------------------------------------------------
def sim1_solver(c, dt, gm, cp2, pe, dm, pm, pem, w, dz, pt, ws, p_fac):
    """FV3 SIM1_solver (MOIST_CAPPA form) on columns [..., k].  w, dz, pe updated in place.
    pe: [..., nz+1] work/output (non-hydrostatic pressure perturbation)."""
    nz = w.shape[-1]
    rgas = c.RDGAS
    t1g = 2.0 * dt * dt
    rdt = 1.0 / dt
    r3 = 1.0 / 3.0
    pe1 = np.exp(gm * np.log(-dm / dz * rgas * pt)) - pm  # [..., nz]
    w1 = w.copy()
    g_rat = dm[..., :-1] / dm[..., 1:]
    bb = np.zeros_like(dm)
    dd = np.zeros_like(dm)
    bb[..., :-1] = 2.0 * (1.0 + g_rat)
    dd[..., :-1] = 3.0 * (pe1[..., :-1] + g_rat * pe1[..., 1:])
    bb[..., nz - 1] = 2.0
    dd[..., nz - 1] = 3.0 * pe1[..., nz - 1]
    pp = np.zeros(dm.shape[:-1] + (nz + 1,))
    gam = np.zeros_like(dm)
    bet = bb[..., 0].copy()
    pp[..., 1] = dd[..., 0] / bet
    for k in range(1, nz):
        gam[..., k] = g_rat[..., k - 1] / bet
        bet = bb[..., k] - gam[..., k]
        pp[..., k + 1] = (dd[..., k] - pp[..., k]) / bet
    for k in range(nz - 1, 0, -1):
        pp[..., k] = pp[..., k] - gam[..., k] * pp[..., k + 1]
    aa = np.zeros_like(dm)
    aa[..., 1:] = t1g * 0.5 * (gm[..., :-1] + gm[..., 1:]) / (dz[..., :-1] + dz[..., 1:]) * (pem[..., 1:nz] + pp[..., 1:nz])
    bet = dm[..., 0] - aa[..., 1]
    w[..., 0] = (dm[..., 0] * w1[..., 0] + dt * pp[..., 1]) / bet
    for k in range(1, nz - 1):
        gam[..., k] = aa[..., k] / bet
        bet = dm[..., k] - (aa[..., k] + aa[..., k + 1] + aa[..., k] * gam[..., k])
        w[..., k] = (dm[..., k] * w1[..., k] + dt * (pp[..., k + 1] - pp[..., k]) - aa[..., k] * w[..., k - 1]) / bet
    p1 = t1g * gm[..., nz - 1] / dz[..., nz - 1] * (pem[..., nz] + pp[..., nz])
    gam[..., nz - 1] = aa[..., nz - 1] / bet
    bet = dm[..., nz - 1] - (aa[..., nz - 1] + p1 + aa[..., nz - 1] * gam[..., nz - 1])
    w[..., nz - 1] = (dm[..., nz - 1] * w1[..., nz - 1] + dt * (pp[..., nz] - pp[..., nz - 1]) - p1 * ws - aa[..., nz - 1] * w[..., nz - 2]) / bet
    for k in range(nz - 2, -1, -1):
        w[..., k] = w[..., k] - gam[..., k + 1] * w[..., k + 1]
    pe[..., 0] = 0.0
    for k in range(nz):
        pe[..., k + 1] = pe[..., k] + dm[..., k] * (w[..., k] - w1[..., k]) * rdt
    p1 = (pe[..., nz - 1] + 2.0 * pe[..., nz]) * r3
    dz[..., nz - 1] = -dm[..., nz - 1] * rgas * pt[..., nz - 1] * np.exp((cp2[..., nz - 1] - 1.0) * np.log(np.maximum(p_fac * pm[..., nz - 1], p1 + pm[..., nz - 1])))
    for k in range(nz - 2, -1, -1):
        p1 = (pe[..., k] + bb[..., k] * pe[..., k + 1] + g_rat[..., k] * pe[..., k + 2]) * r3 - g_rat[..., k] * p1
        dz[..., k] = -dm[..., k] * rgas * pt[..., k] * np.exp((cp2[..., k] - 1.0) * np.log(np.maximum(p_fac * pm[..., k], p1 + pm[..., k])))


def riem_solver_c(D: Dom, dt2, cappa, ptop, phis, ws, ptc, q_con, delpc, gz, pef, w3, p_fac):
    """(dt2, cappa, ptop, phis, ws, ptc, q_con, delpc, gz, pef, w3) as the reference operator.
    In: gz = interface height (m); out: gz = geopotential, pef = full interface pressure (pkc)."""
    S = D.sl
    nz = D.nz
    c = D.c
    R = S(D.is_ - 1, D.ie + 1, D.js - 1, D.je + 1)
    dm = delpc[R][:, :, :nz].copy()
    cp2 = cappa[R][:, :, :nz]
    qc = q_con[R][:, :, :nz]
    shp = dm.shape[:2] + (nz + 1,)
    pem = np.zeros(shp)
    peg = np.zeros(shp)
    pem[..., 0] = ptop
    peg[..., 0] = ptop
    for k in range(nz):
        pem[..., k + 1] = pem[..., k] + dm[..., k]
        peg[..., k + 1] = peg[..., k] + dm[..., k] * (1.0 - qc[..., k])
    g = gz[R]
    dz2 = g[..., 1:] - g[..., :-1]
    pm2 = (peg[..., 1:] - peg[..., :-1]) / np.log(peg[..., 1:] / peg[..., :-1])
    gm2 = 1.0 / (1.0 - cp2)
    dm = dm * c.RGRAV
    w2 = w3[R][:, :, :nz].copy()
    pe2 = np.zeros(shp)
    sim1_solver(c, dt2, gm2, cp2, pe2, dm, pm2, pem, w2, dz2, ptc[R][:, :, :nz], ws[R][:, :, 0], p_fac)
    out = pe2 + pem
    out[..., 0] = ptop
    pef[R] = out
    g = np.zeros(shp)
    g[..., nz] = phis[R][:, :, 0]
    for k in range(nz - 1, -1, -1):
        g[..., k] = g[..., k + 1] - dz2[..., k] * c.GRAV
    gz[R] = g


def riem_solver3(D: Dom, last_call, dt, cappa, ptop, zs, ws, delz, q_con, delp, pt, zh, pe, ppe, pk3, pk, peln, w, p_fac):
    """(last_call, dt, cappa, ptop, zs, wsd, delz, q_con, delp, pt, zh, pe, ppe, pk3, pk, peln, w) as the
    reference operator.  Updates w, delz, zh, ppe (= pkc), pk3 (+ pe, pk, peln on the last call)."""
    S = D.sl
    nz = D.nz
    c = D.c
    akap = c.KAPPA
    R = S(D.is_, D.ie, D.js, D.je)
    dm = delp[R][:, :, :nz].copy()
    cp2 = cappa[R][:, :, :nz]
    qc = q_con[R][:, :, :nz]
    shp = dm.shape[:2] + (nz + 1,)
    pem = np.zeros(shp)
    peg = np.zeros(shp)
    pem[..., 0] = ptop
    peg[..., 0] = ptop
    for k in range(nz):
        pem[..., k + 1] = pem[..., k] + dm[..., k]
        peg[..., k + 1] = peg[..., k] + dm[..., k] * (1.0 - qc[..., k])
    peln2 = np.log(pem)
    pelng = np.log(peg)
    pk3v = np.exp(akap * peln2)
    pm2 = (peg[..., 1:] - peg[..., :-1]) / (pelng[..., 1:] - pelng[..., :-1])
    gm2 = 1.0 / (1.0 - cp2)
    dm = dm * c.RGRAV
    z = zh[R]
    dz2 = z[..., 1:] - z[..., :-1]
    w2 = w[R][:, :, :nz].copy()
    pe2 = np.zeros(shp)
    sim1_solver(c, dt, gm2, cp2, pe2, dm, pm2, pem, w2, dz2, pt[R][:, :, :nz], ws[R][:, :, 0], p_fac)
    w[R + (slice(0, nz),)] = w2
    delz[R + (slice(0, nz),)] = dz2
    pk3[R] = pk3v
    if last_call:
        peln[R] = peln2
        pk[R] = pk3v
        pe[R] = pem
    ppe[R] = pe2
    znew = np.zeros(shp)
    znew[..., nz] = zs[R][:, :, 0]
    for k in range(nz - 1, -1, -1):
        znew[..., k] = znew[..., k + 1] - dz2[..., k]
    zh[R] = znew


# ---------------------------------------------------------------------------
# pressure-gradient forces  [SURVEY A.7 / A.10]
# ---------------------------------------------------------------------------
def p_grad_c(D: Dom, rdxc, rdyc, uc, vc, delpc, pkc, gz, dt2):
    """C-grid pressure gradient update of uc, vc (non-hydrostatic: weight = delpc)."""
    S = D.sl
    nz = D.nz
    is_, ie, js, je = D.is_, D.ie, D.js, D.je
    K0 = (slice(0, nz),)
    K1 = (slice(1, nz + 1),)
    R = S(is_, ie + 1, js, je)
    Rm = S(is_ - 1, ie, js, je)
    uc[R + K0] = uc[R + K0] + dt2 * rdxc[R] / (delpc[Rm + K0] + delpc[R + K0]) * (
        (gz[Rm + K1] - gz[R + K0]) * (pkc[R + K1] - pkc[Rm + K0]) + (gz[Rm + K0] - gz[R + K1]) * (pkc[Rm + K1] - pkc[R + K0])
    )
    R = S(is_, ie, js, je + 1)
    Rm = S(is_, ie, js - 1, je)
    vc[R + K0] = vc[R + K0] + dt2 * rdyc[R] / (delpc[Rm + K0] + delpc[R + K0]) * (
        (gz[Rm + K1] - gz[R + K0]) * (pkc[R + K1] - pkc[Rm + K0]) + (gz[Rm + K0] - gz[R + K1]) * (pkc[Rm + K1] - pkc[R + K0])
    )


def nh_p_grad(D: Dom, u, v, pp, gz, pk3, delp, dt, ptop, akap):
    """D-grid non-hydrostatic pressure gradient (u, v, pp, gz, pk3, delp, dt, ptop, akap).
    pp, pk3, gz are interpolated to corners in place (a2b_ord4 replace) like the reference."""
    S = D.sl
    m = D.m
    nz = D.nz
    is_, ie, js, je = D.is_, D.ie, D.js, D.je
    Rc = S(is_, ie + 1, js, je + 1)
    pp[Rc + (slice(0, 1),)] = 0.0
    pk3[Rc + (slice(0, 1),)] = ptop**akap
    a = pp[:, :, 1 : nz + 1]
    a2b_ord4(D, a, replace=True)
    a = pk3[:, :, 1 : nz + 1]
    a2b_ord4(D, a, replace=True)
    a = gz[:, :, 0 : nz + 1]
    a2b_ord4(D, a, replace=True)
    wk1 = a2b_ord4(D, delp[:, :, :nz].copy(), replace=False)
    wk = np.zeros_like(wk1)
    wk[Rc] = pk3[Rc][:, :, 1 : nz + 1] - pk3[Rc][:, :, 0:nz]
    K0 = (slice(0, nz),)
    K1 = (slice(1, nz + 1),)
    R = S(is_, ie, js, je + 1)
    Rp = S(is_ + 1, ie + 1, js, je + 1)
    du = dt / (wk[R] + wk[Rp]) * ((gz[R + K1] - gz[Rp + K0]) * (pk3[Rp + K1] - pk3[R + K0]) + (gz[R + K0] - gz[Rp + K1]) * (pk3[R + K1] - pk3[Rp + K0]))
    u[R + K0] = (
        u[R + K0]
        + du
        + dt / (wk1[R] + wk1[Rp]) * ((gz[R + K1] - gz[Rp + K0]) * (pp[Rp + K1] - pp[R + K0]) + (gz[R + K0] - gz[Rp + K1]) * (pp[R + K1] - pp[Rp + K0]))
    ) * m.rdx[R]
    R = S(is_, ie + 1, js, je)
    Rp = S(is_, ie + 1, js + 1, je + 1)
    dv = dt / (wk[R] + wk[Rp]) * ((gz[R + K1] - gz[Rp + K0]) * (pk3[Rp + K1] - pk3[R + K0]) + (gz[R + K0] - gz[Rp + K1]) * (pk3[R + K1] - pk3[Rp + K0]))
    v[R + K0] = (
        v[R + K0]
        + dv
        + dt / (wk1[R] + wk1[Rp]) * ((gz[R + K1] - gz[Rp + K0]) * (pp[Rp + K1] - pp[R + K0]) + (gz[R + K0] - gz[Rp + K1]) * (pp[R + K1] - pp[Rp + K0]))
    ) * m.rdy[R]


def pk3_halo(D: Dom, pk3, delp, ptop, akap):
    """pk3 = pe**akap in the 2-wide halo ring (FV3 pk3_halo)."""
    nz = D.nz
    o = D.o
    is_, ie, js, je = D.is_, D.ie, D.js, D.je

    def column(si, sj):
        pei = np.full(delp[si, sj, 0].shape, ptop)
        for k in range(nz):
            pei = pei + delp[si, sj, k]
            pk3[si, sj, k + 1] = np.exp(akap * np.log(pei))

    jr = slice(js + o, je + o + 1)
    column(slice(is_ - 2 + o, is_ + o), jr)
    column(slice(ie + 1 + o, ie + 3 + o), jr)
    ir = slice(is_ - 2 + o, ie + 3 + o)
    column(ir, slice(js - 2 + o, js + o))
    column(ir, slice(je + 1 + o, je + 3 + o))


def pe_halo(D: Dom, pe, delp, ptop):
    """Hydrostatic interface pressure in the 1-wide halo ring (FV3 pe_halo / pyFV3 edge_pe)."""
    nz = D.nz
    o = D.o
    is_, ie, js, je = D.is_, D.ie, D.js, D.je

    def column(si, sj):
        pei = np.full(delp[si, sj, 0].shape, ptop)
        pe[si, sj, 0] = pei
        for k in range(nz):
            pei = pei + delp[si, sj, k]
            pe[si, sj, k + 1] = pei

    jr = slice(js + o, je + o + 1)
    column(slice(is_ - 1 + o, is_ + o), jr)
    column(slice(ie + 1 + o, ie + 2 + o), jr)
    ir = slice(is_ - 1 + o, ie + 2 + o)
    column(ir, slice(js - 1 + o, js + o))
    column(ir, slice(je + 1 + o, je + 2 + o))


# ---------------------------------------------------------------------------
# Rayleigh damping  [SURVEY A.11; pyFV3 ray_fast.py]
# ---------------------------------------------------------------------------
def ray_fast(D: Dom, cfg, u, v, w, dp, pfull, dt, ptop):
    S = D.sl
    nz = D.nz
    c = D.c
    rf_cutoff = cfg.rf_cutoff
    rf_cutoff_nudge = rf_cutoff + min(100.0, 10.0 * ptop)
    tau0 = cfg.tau * c.SECONDS_PER_DAY
    damped = pfull < rf_cutoff
    nudged = pfull < rf_cutoff_nudge
    rf = np.ones(nz)
    rfv = dt / tau0 * np.sin(0.5 * c.PI * np.log(rf_cutoff / pfull[damped]) / np.log(rf_cutoff / ptop)) ** 2
    rf[damped] = 1.0 / (1.0 + rfv)
    if not nudged.any():
        return
    dm = np.sum(dp[nudged])
    for a, R in ((u, S(D.is_, D.ie, D.js, D.je + 1)), (v, S(D.is_, D.ie + 1, D.js, D.je))):
        x = a[R][:, :, :nz]
        dmdir = np.sum(((1.0 - rf) * dp)[damped] * x[:, :, damped], axis=-1, keepdims=True)
        x[:, :, damped] = x[:, :, damped] * rf[damped]
        if not alt("ray_fast_plain"):  # (FV3_ALT=ray_fast_plain: the older form without the momentum fix -- DESIGN §2, uncertain restatement 4)
            x[:, :, nudged] = x[:, :, nudged] + dmdir / dm
        a[R + (slice(0, nz),)] = x
    R = S(D.is_, D.ie, D.js, D.je)
    x = w[R][:, :, :nz]
    x[:, :, damped] = x[:, :, damped] * rf[damped]
    w[R + (slice(0, nz),)] = x


# ---------------------------------------------------------------------------
# end-of-call heat diffusion  [SURVEY A.12]
# ---------------------------------------------------------------------------
def del2_cubed(D: Dom, q, cd, nmax=3):
    S = D.sl
    o = D.o
    m = D.m
    is_, ie, js, je, npx, npy = D.is_, D.ie, D.js, D.je, D.npx, D.npy
    ntimes = min(3, nmax)
    r3 = 1.0 / 3.0
    for n in range(1, ntimes + 1):
        nt = ntimes - n
        if D.sw:
            q[1 + o, 1 + o] = (q[1 + o, 1 + o] + q[0 + o, 1 + o] + q[1 + o, 0 + o]) * r3
            q[0 + o, 1 + o] = q[1 + o, 1 + o]
            q[1 + o, 0 + o] = q[1 + o, 1 + o]
        if D.se:
            q[ie + o, 1 + o] = (q[ie + o, 1 + o] + q[npx + o, 1 + o] + q[ie + o, 0 + o]) * r3
            q[npx + o, 1 + o] = q[ie + o, 1 + o]
            q[ie + o, 0 + o] = q[ie + o, 1 + o]
        if D.ne:
            q[ie + o, je + o] = (q[ie + o, je + o] + q[npx + o, je + o] + q[ie + o, npy + o]) * r3
            q[npx + o, je + o] = q[ie + o, je + o]
            q[ie + o, npy + o] = q[ie + o, je + o]
        if D.nw:
            q[1 + o, je + o] = (q[1 + o, je + o] + q[0 + o, je + o] + q[1 + o, npy + o]) * r3
            q[0 + o, je + o] = q[1 + o, je + o]
            q[1 + o, npy + o] = q[1 + o, je + o]
        fx = np.zeros_like(q)
        fy = np.zeros_like(q)
        if nt > 0 and (D.sw or D.se or D.ne or D.nw):
            copy_corners(D, q, 1)
        R = S(is_ - nt, ie + 1 + nt, js - nt, je + nt)
        fx[R] = m.del6_v[R] * (q[S(is_ - nt - 1, ie + nt, js - nt, je + nt)] - q[R])
        if nt > 0 and (D.sw or D.se or D.ne or D.nw):
            copy_corners(D, q, 2)
        R = S(is_ - nt, ie + nt, js - nt, je + 1 + nt)
        fy[R] = m.del6_u[R] * (q[S(is_ - nt, ie + nt, js - nt - 1, je + nt)] - q[R])
        R = S(is_ - nt, ie + nt, js - nt, je + nt)
        q[R] = q[R] + cd * m.rarea[R] * (fx[R] - fx[S(is_ - nt + 1, ie + nt + 1, js - nt, je + nt)] + fy[R] - fy[S(is_ - nt, ie + nt, js - nt + 1, je + nt + 1)])


def apply_diffusive_heating(D: Dom, delp, delz, cappa, heat_source, pt, delt_time_factor):
    S = D.sl
    nz = D.nz
    c = D.c
    R = S(D.is_, D.ie, D.js, D.je) + (slice(0, nz),)
    pkz = np.exp(cappa[R] / (1.0 - cappa[R]) * np.log(c.RDG * delp[R] / delz[R] * pt[R]))
    dtmp = heat_source[R] / (c.CV_AIR * delp[R])
    lim = np.full(nz, delt_time_factor)
    lim[0] *= 0.1
    if nz > 1:
        lim[1] *= 0.5
    pt[R] = pt[R] + np.sign(dtmp) * np.minimum(lim, np.abs(dtmp)) / pkz
