"""Seeded synthetic initial states for the acoustic path (host side, numpy).

The reference initialises from the JW2006 baroclinic wave through
``pyFV3.initialization.analytic_init`` [REF driver/pace/driver/initialization.py:116-124],
which is un-vendored; for throughput and parity runs SURVEY §8d prescribes
seeded smooth synthetic fields instead.  The recipe here follows it, with two
physical refinements so that the acoustic loop is *stable* on the input:

* the wind is one smooth 3-D vector field projected on the local D-grid edge
  directions (globally continuous across tile edges), and
* ``delz`` is in discrete hydrostatic balance with ``delp``/``pt`` in the sense of
  the semi-implicit solver (its pressure perturbation starts at ~0).

State arrays are ``[i, j, k]`` with the padded storage shape
``(nx+7, ny+7, nz+1)``; value ranges satisfy the reference ``SafetyChecker``
bounds (ua,va in [-200,200], delp in [-1,4000], pt in [100,380] K when
converted) [REF driver/pace/driver/driver.py:557-560].
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .constants import ConstantSet, get_constants
from .grid import GridData

STATE_3D = "u v w ua va uc vc delp delz pt pe pk peln pkz q_con omga cappa mfxd mfyd cxd cyd diss_estd".split()


def _sph(lon, lat):
    return np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)], axis=-1)


class _Modes:
    """Sum of a few low-wavenumber plane waves evaluated on the unit sphere."""

    def __init__(self, rng, n=4, kmax=3):
        self.k = rng.integers(-kmax, kmax + 1, size=(n, 3)).astype(np.float64)
        self.k[np.all(self.k == 0, axis=1)] = (1.0, 0.0, 0.0)
        self.phase = rng.uniform(0, 2 * np.pi, size=n)
        self.amp = rng.uniform(0.5, 1.0, size=n) / n

    def __call__(self, p):
        out = np.zeros(p.shape[:-1])
        for k, ph, a in zip(self.k, self.phase, self.amp):
            out += a * np.sin(p @ k + ph)
        return out


def synthetic_state(
    grid: GridData,
    seed: int = 20261002,
    constants: Optional[ConstantSet] = None,
    noise: float = 0.01,
    wind: float = 20.0,
    mountain: float = 0.0,
    rank: int = 0,
) -> Dict[str, np.ndarray]:
    """Synthetic fields for one rank.  The *smooth* part depends only on ``seed`` (so it is
    one global field sampled by every rank); the white-noise part is seeded per rank."""
    c = constants or get_constants()
    nh, nx, ny, nz = grid.n_halo, grid.nx, grid.ny, grid.nz
    shp2 = (nx + 2 * nh + 1, ny + 2 * nh + 1)
    shp = shp2 + (nz + 1,)
    rng = np.random.default_rng(seed)
    modes = [_Modes(rng) for _ in range(8)]
    axis = _sph(np.array(0.3), np.array(1.1))  # rotation axis of the background flow
    wvec_modes = [_Modes(rng) for _ in range(3)]
    rrng = np.random.default_rng(seed + 1000 + rank)

    Pc = _sph(grid.lon, grid.lat)  # corners
    Pa = _sph(grid.lon_agrid, grid.lat_agrid)  # centres (last row/col padded)

    def wind3(p):
        """smooth tangent wind: solid-body rotation + rotational modes"""
        w = np.cross(axis, p) * 1.0
        psi = np.stack([m(p) for m in wvec_modes], axis=-1)
        w = w + 0.5 * np.cross(psi, p)
        return wind * w

    st = {n: np.zeros(shp) for n in STATE_3D}
    ak, bk = grid.ak, grid.bk
    ps = 1.0e5 * (1.0 + 0.01 * modes[0](Pa))
    # vertical structure factors so fields decorrelate in k
    kk = np.arange(nz)
    vs = [np.cos(2 * np.pi * (kk + 3.0 * i) / nz) for i in range(6)]

    def field3(i, p):
        base = modes[i](p)
        other = modes[(i + 3) % len(modes)](p)
        f = base[:, :, None] * (0.7 + 0.3 * vs[i % 6])[None, None, :] + 0.3 * other[:, :, None] * vs[(i + 1) % 6][None, None, :]
        if noise > 0:
            f = f + noise * rrng.standard_normal(f.shape)
        return f

    pe = ak[None, None, :] + bk[None, None, :] * ps[:, :, None]
    delp = (pe[:, :, 1:] - pe[:, :, :-1]) * (1.0 + 0.002 * field3(1, Pa))
    q_con = 1.0e-4 * np.abs(field3(6, Pa))
    cappa = c.KAPPA * (1.0 - 0.2 * q_con)
    pem = np.concatenate([np.full(shp2 + (1,), ak[0]), ak[0] + np.cumsum(delp, axis=-1)], axis=-1)
    peg = np.concatenate([np.full(shp2 + (1,), ak[0]), ak[0] + np.cumsum(delp * (1.0 - q_con), axis=-1)], axis=-1)
    pm = (peg[:, :, 1:] - peg[:, :, :-1]) / np.log(peg[:, :, 1:] / peg[:, :, :-1])
    temp = np.maximum(200.0, 288.0 * (pm / 1.0e5) ** 0.19) * (1.0 + 0.005 * field3(2, Pa))
    pkz = np.exp(cappa / (1.0 - cappa) * np.log(pm ** (1.0 - cappa)))  # = pm**cappa
    pt = temp / pkz
    delz = -c.RDGAS * temp * delp / (c.GRAV * pm)

    # winds: project the smooth 3-D field on the D-grid edge directions
    ex = Pc[1:, :, :] - Pc[:-1, :, :]
    mx = Pc[1:, :, :] + Pc[:-1, :, :]
    mx /= np.linalg.norm(mx, axis=-1, keepdims=True)
    ex -= np.sum(ex * mx, -1, keepdims=True) * mx
    ex /= np.linalg.norm(ex, axis=-1, keepdims=True)
    ey = Pc[:, 1:, :] - Pc[:, :-1, :]
    my = Pc[:, 1:, :] + Pc[:, :-1, :]
    my /= np.linalg.norm(my, axis=-1, keepdims=True)
    ey -= np.sum(ey * my, -1, keepdims=True) * my
    ey /= np.linalg.norm(ey, axis=-1, keepdims=True)
    u2 = np.sum(wind3(mx) * ex, -1)  # (nI-1, nJ)
    v2 = np.sum(wind3(my) * ey, -1)  # (nI, nJ-1)
    prof = (0.6 + 0.4 * np.cos(np.pi * (kk + 0.5) / nz))[None, None, :]
    st["u"][:-1, :, :nz] = u2[:, :, None] * prof
    st["v"][:, :-1, :nz] = v2[:, :, None] * prof
    if noise > 0:
        st["u"][:, :, :nz] += wind * 0.2 * noise * rrng.standard_normal(st["u"][:, :, :nz].shape)
        st["v"][:, :, :nz] += wind * 0.2 * noise * rrng.standard_normal(st["v"][:, :, :nz].shape)
    st["w"][:, :, :nz] = 0.1 * field3(5, Pa)
    st["delp"][:, :, :nz] = delp
    st["pt"][:, :, :nz] = pt
    st["delz"][:, :, :nz] = delz
    st["q_con"][:, :, :nz] = q_con
    st["cappa"][:, :, :nz] = cappa
    st["pkz"][:, :, :nz] = pkz
    st["pe"][:, :, :] = pem
    st["peln"][:, :, :] = np.log(pem)
    st["pk"][:, :, :] = np.exp(c.KAPPA * np.log(pem))
    # keep the padded level finite
    for n in ("delp", "pt", "delz", "cappa"):
        st[n][:, :, nz] = st[n][:, :, nz - 1]
    phis = np.zeros(shp2 + (1,))
    if mountain > 0:
        phis[:, :, 0] = c.GRAV * mountain * np.maximum(0.0, modes[7](Pa)) ** 2
    st["phis"] = phis
    return st


def synthetic_state_device(grid: GridData, device, seed: int = 20261002, constants: Optional[ConstantSet] = None, noise: float = 0.01, wind: float = 20.0, rank: int = 0):
    """The same recipe as :func:`synthetic_state`, evaluated with torch on ``device`` (host numpy
    takes ~1 min per 384^2 x 79 rank; the benchmark fills 24 of them).  Smooth parts agree with
    the numpy version to round-off; the white noise comes from torch's generator instead of
    numpy's.  Returns {name: tensor[i, j, k]} plus "phis" [i, j]."""
    import torch

    c = constants or get_constants()
    nh, nx, ny, nz = grid.n_halo, grid.nx, grid.ny, grid.nz
    shp2 = (nx + 2 * nh + 1, ny + 2 * nh + 1)
    f64 = torch.float64
    dev = torch.device(device)
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=f64, device=dev)  # noqa: E731
    rng = np.random.default_rng(seed)
    modes = [_Modes(rng) for _ in range(8)]
    axis = T(_sph(np.array(0.3), np.array(1.1)))
    wvec_modes = [_Modes(rng) for _ in range(3)]
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + 1000 + rank)

    def sph(lon, lat):
        lon, lat = T(lon), T(lat)
        return torch.stack([torch.cos(lat) * torch.cos(lon), torch.cos(lat) * torch.sin(lon), torch.sin(lat)], dim=-1)

    def ev(m, p):
        out = torch.zeros(p.shape[:-1], dtype=f64, device=dev)
        for k, ph, a in zip(m.k, m.phase, m.amp):
            out = out + float(a) * torch.sin(p @ T(k) + float(ph))
        return out

    Pc = sph(grid.lon, grid.lat)
    Pa = sph(grid.lon_agrid, grid.lat_agrid)

    def wind3(p):
        w = torch.cross(axis.expand_as(p), p, dim=-1)
        psi = torch.stack([ev(m, p) for m in wvec_modes], dim=-1)
        return wind * (w + 0.5 * torch.cross(psi, p, dim=-1))

    kk = torch.arange(nz, dtype=f64, device=dev)
    vs = [torch.cos(2 * np.pi * (kk + 3.0 * i) / nz) for i in range(6)]

    def field3(i, p):
        base = ev(modes[i], p)
        other = ev(modes[(i + 3) % len(modes)], p)
        f = base[:, :, None] * (0.7 + 0.3 * vs[i % 6])[None, None, :] + 0.3 * other[:, :, None] * vs[(i + 1) % 6][None, None, :]
        if noise > 0:
            f = f + noise * torch.randn(f.shape, dtype=f64, device=dev, generator=gen)
        return f

    ak, bk = T(grid.ak), T(grid.bk)
    ps = 1.0e5 * (1.0 + 0.01 * ev(modes[0], Pa))
    pe = ak[None, None, :] + bk[None, None, :] * ps[:, :, None]
    delp = (pe[:, :, 1:] - pe[:, :, :-1]) * (1.0 + 0.002 * field3(1, Pa))
    q_con = 1.0e-4 * torch.abs(field3(6, Pa))
    cappa = c.KAPPA * (1.0 - 0.2 * q_con)
    top = torch.full(shp2 + (1,), float(grid.ak[0]), dtype=f64, device=dev)
    pem = torch.cat([top, float(grid.ak[0]) + torch.cumsum(delp, dim=-1)], dim=-1)
    peg = torch.cat([top, float(grid.ak[0]) + torch.cumsum(delp * (1.0 - q_con), dim=-1)], dim=-1)
    pm = (peg[:, :, 1:] - peg[:, :, :-1]) / torch.log(peg[:, :, 1:] / peg[:, :, :-1])
    temp = torch.clamp(288.0 * (pm / 1.0e5) ** 0.19, min=200.0) * (1.0 + 0.005 * field3(2, Pa))
    pkz = torch.exp(cappa * torch.log(pm))
    st = {n: torch.zeros(shp2 + (nz + 1,), dtype=f64, device=dev) for n in STATE_3D}

    def edge_dir(a, b):
        e = b - a
        m = a + b
        m = m / torch.linalg.norm(m, dim=-1, keepdim=True)
        e = e - torch.sum(e * m, -1, keepdim=True) * m
        return e / torch.linalg.norm(e, dim=-1, keepdim=True), m

    ex, mx = edge_dir(Pc[:-1, :, :], Pc[1:, :, :])
    ey, my = edge_dir(Pc[:, :-1, :], Pc[:, 1:, :])
    prof = (0.6 + 0.4 * torch.cos(np.pi * (kk + 0.5) / nz))[None, None, :]
    st["u"][:-1, :, :nz] = torch.sum(wind3(mx) * ex, -1)[:, :, None] * prof
    st["v"][:, :-1, :nz] = torch.sum(wind3(my) * ey, -1)[:, :, None] * prof
    if noise > 0:
        st["u"][:, :, :nz] += wind * 0.2 * noise * torch.randn(st["u"][:, :, :nz].shape, dtype=f64, device=dev, generator=gen)
        st["v"][:, :, :nz] += wind * 0.2 * noise * torch.randn(st["v"][:, :, :nz].shape, dtype=f64, device=dev, generator=gen)
    st["w"][:, :, :nz] = 0.1 * field3(5, Pa)
    st["delp"][:, :, :nz] = delp
    st["pt"][:, :, :nz] = temp / pkz
    st["delz"][:, :, :nz] = -c.RDGAS * temp * delp / (c.GRAV * pm)
    st["q_con"][:, :, :nz] = q_con
    st["cappa"][:, :, :nz] = cappa
    st["pkz"][:, :, :nz] = pkz
    st["pe"][:] = pem
    st["peln"][:] = torch.log(pem)
    st["pk"][:] = torch.exp(c.KAPPA * torch.log(pem))
    for n in ("delp", "pt", "delz", "cappa"):
        st[n][:, :, nz] = st[n][:, :, nz - 1]
    st["phis"] = torch.zeros(shp2, dtype=f64, device=dev)
    return st


# ---------------------------------------------------------------------------------------------
# Jablonowski & Williamson (2006) baroclinic-wave test case -- the reference's default analytic
# initial state (`initialization: {type: analytic, config: {case: baroclinic}}`
# [REF driver/examples/configs/baroclinic_c12.yaml:8-11; tests/main/fv3core/test_dycore_call.py:108-118]).
# pyFV3's `init_analytic_state` is un-vendored, so this is a restatement of the published
# formulas (JW2006, QJRMS 132, eqs. 2-10), sampled the way the acoustic path expects its inputs:
# point values at cell centres and at the layer-mean eta, D-grid winds as projections of the
# zonal wind on the local edge directions, pt = T / pkz, delz in discrete hydrostatic balance,
# w = 0, dry (q_con = 0, cappa = kappa).  pyFV3 integrates some fields over the cell / layer instead
# of sampling them, so values will differ from it at truncation-error level: parity against the
# reference for this state is UNPINNED like the rest.
# ---------------------------------------------------------------------------------------------
JW_U0 = 35.0
JW_ETA0 = 0.252
JW_ETA_T = 0.2
JW_T0 = 288.0
JW_GAMMA = 0.005
JW_DELTA_T = 4.8e5
JW_UP = 1.0
JW_LON_C = np.pi / 9.0
JW_LAT_C = 2.0 * np.pi / 9.0


def jw_temperature(lat, eta, c: ConstantSet):
    """T(lat, eta): horizontal-mean profile + the thermal-wind balanced deviation (JW2006 eqs. 4-6)."""
    eta_v = (eta - JW_ETA0) * 0.5 * np.pi
    tbar = JW_T0 * eta ** (c.RDGAS * JW_GAMMA / c.GRAV)
    tbar = np.where(eta < JW_ETA_T, tbar + JW_DELTA_T * np.maximum(JW_ETA_T - eta, 0.0) ** 5, tbar)
    s, co = np.sin(lat), np.cos(lat)
    a_term = (-2.0 * s**6 * (co**2 + 1.0 / 3.0) + 10.0 / 63.0) * 2.0 * JW_U0 * np.cos(eta_v) ** 1.5
    b_term = (1.6 * co**3 * (s**2 + 2.0 / 3.0) - 0.25 * np.pi) * c.RADIUS * c.OMEGA
    return tbar + 0.75 * eta * np.pi * JW_U0 / c.RDGAS * np.sin(eta_v) * np.sqrt(np.cos(eta_v)) * (a_term + b_term)


def jw_surface_geopotential(lat, c: ConstantSet):
    """phis(lat) balancing the zonal flow at eta = 1 (JW2006 eq. 7)."""
    cz = np.cos((1.0 - JW_ETA0) * 0.5 * np.pi) ** 1.5
    s, co = np.sin(lat), np.cos(lat)
    return JW_U0 * cz * ((-2.0 * s**6 * (co**2 + 1.0 / 3.0) + 10.0 / 63.0) * JW_U0 * cz + (1.6 * co**3 * (s**2 + 2.0 / 3.0) - 0.25 * np.pi) * c.RADIUS * c.OMEGA)


def jw_zonal_wind(lon, lat, eta, c: ConstantSet, perturbation=True):
    """u(lon, lat, eta) (JW2006 eq. 2) + the Gaussian perturbation centred at (20E, 40N) (eq. 10)."""
    eta_v = (eta - JW_ETA0) * 0.5 * np.pi
    u = JW_U0 * np.cos(eta_v) ** 1.5 * np.sin(2.0 * lat) ** 2
    if perturbation:
        r = np.arccos(np.clip(np.sin(JW_LAT_C) * np.sin(lat) + np.cos(JW_LAT_C) * np.cos(lat) * np.cos(lon - JW_LON_C), -1.0, 1.0))
        u = u + JW_UP * np.exp(-((r * 10.0) ** 2))  # R = a / 10, r in units of a
    return u


def baroclinic_state(grid: GridData, constants: Optional[ConstantSet] = None, perturbation: bool = True) -> Dict[str, np.ndarray]:
    """JW2006 baroclinic wave for one rank; same array conventions as :func:`synthetic_state`."""
    c = constants or get_constants()
    nh, nx, ny, nz = grid.n_halo, grid.nx, grid.ny, grid.nz
    shp2 = (nx + 2 * nh + 1, ny + 2 * nh + 1)
    shp = shp2 + (nz + 1,)
    st = {n: np.zeros(shp) for n in STATE_3D}
    ak, bk = grid.ak, grid.bk
    p0 = 1.0e5
    pe1 = ak + bk * p0  # interface pressure (ps = p0 everywhere)
    eta = 0.5 * (pe1[1:] + pe1[:-1]) / p0  # layer-mean eta
    delp1 = pe1[1:] - pe1[:-1]
    pm1 = delp1 / np.log(pe1[1:] / pe1[:-1])
    lat_a, lon_a = grid.lat_agrid, grid.lon_agrid
    temp = jw_temperature(lat_a[:, :, None], eta[None, None, :], c)
    delp = np.broadcast_to(delp1[None, None, :], shp2 + (nz,)).copy()
    pm = np.broadcast_to(pm1[None, None, :], shp2 + (nz,))
    pkz = pm**c.KAPPA
    st["delp"][:, :, :nz] = delp
    st["pt"][:, :, :nz] = temp / pkz
    st["delz"][:, :, :nz] = -c.RDGAS * temp * delp / (c.GRAV * pm)
    st["cappa"][:, :, :nz] = c.KAPPA
    st["pkz"][:, :, :nz] = pkz
    pem = np.broadcast_to(pe1[None, None, :], shp)
    st["pe"][:, :, :] = pem
    st["peln"][:, :, :] = np.log(pem)
    st["pk"][:, :, :] = np.exp(c.KAPPA * np.log(pem))
    for n in ("delp", "pt", "delz", "cappa"):
        st[n][:, :, nz] = st[n][:, :, nz - 1]

    # D-grid winds: zonal wind at the edge mid-points projected on the edge direction
    Pc = _sph(grid.lon, grid.lat)

    def edges(a, b):
        e = b - a
        m = a + b
        m /= np.linalg.norm(m, axis=-1, keepdims=True)
        e -= np.sum(e * m, -1, keepdims=True) * m
        e /= np.linalg.norm(e, axis=-1, keepdims=True)
        return e, m

    def project(e, m):
        lon = np.arctan2(m[..., 1], m[..., 0])
        lat = np.arcsin(np.clip(m[..., 2], -1.0, 1.0))
        east = np.stack([-np.sin(lon), np.cos(lon), np.zeros_like(lon)], axis=-1)
        along = np.sum(east * e, -1)  # cos of the angle between the edge and the local east
        return jw_zonal_wind(lon[:, :, None], lat[:, :, None], eta[None, None, :], c, perturbation) * along[:, :, None]

    ex, mx = edges(Pc[:-1, :, :], Pc[1:, :, :])
    ey, my = edges(Pc[:, :-1, :], Pc[:, 1:, :])
    st["u"][:-1, :, :nz] = project(ex, mx)
    st["v"][:, :-1, :nz] = project(ey, my)
    phis = np.zeros(shp2 + (1,))
    phis[:, :, 0] = jw_surface_geopotential(lat_a, c)
    st["phis"] = phis
    return st


# ---------------------------------------------------------------------------------------------
# A real FV3 model state: the Fortran restart the reference tree holds (C12, L63, six tiles)
# [REF tests/main/data/c12_restart/fv_core.res.tile[1-6].nc, fv_tracer.res.tile[1-6].nc; read by the reference through
# FortranRestartInit, driver/pace/driver/initialization.py:174-229, tests/main/driver/test_restart_fortran.py:21-67].
# ---------------------------------------------------------------------------------------------
def restart_state(grid: GridData, data, tile: int, origin=(0, 0), constants: Optional[ConstantSet] = None) -> Dict[str, np.ndarray]:
    """One rank's state from the six-tile restart arrays ``data`` (``tests/golden/c12_restart_6tiles.npz``: per tile
    ``u[k, y_if, x]``, ``v[k, y, x_if]``, ``W, DZ, T, delp, sphum, liq_wat [k, y, x]``, ``phis[y, x]``), sub-domain at ``origin``
    (cells) of tile ``tile``.  As the reference does for a Fortran restart: the prognostic fields are taken as they are
    (u, v, w, delz, delp), pt = T / pkz with the state's own full pressure (non-hydrostatic), pe / peln / pk from ptop + the
    running sum of delp [REF driver/pace/driver/initialization.py:375-395].  q_con = the liquid-water mixing ratio and
    cappa = kappa (1 - 0.2 q_con) stand in for moist_cv, which is outside this build.  Halos are NOT model data: they
    hold edge-replicated values until the first halo update (every operator of the path exchanges before it reads them)."""
    c = constants or get_constants()
    nh, nx, ny, nz = grid.n_halo, grid.nx, grid.ny, grid.nz
    ox, oy = origin
    shp2 = (nx + 2 * nh + 1, ny + 2 * nh + 1)
    shp = shp2 + (nz + 1,)
    st = {n: np.zeros(shp) for n in STATE_3D}

    def cells(name):
        return np.transpose(np.asarray(data[name][tile], dtype=np.float64), (2, 1, 0))[ox : ox + nx, oy : oy + ny, :nz]

    def place(name, a, ex=0, ey=0):
        """compute domain (+ the staggered interface) from a; everything else edge-replicated"""
        full = np.pad(a, ((nh, nh + 1 - ex), (nh, nh + 1 - ey), (0, 1)), mode="edge")
        st[name][...] = full

    delp, delz, q = cells("delp"), cells("DZ"), cells("liq_wat")
    # the dycore's temperature is the density temperature T (1 + zvir q_v) (1 - q_con): the equation of state of the moist model
    # state p = rho R T_v is what balances its pressure gradients (with the dry T the real state sheds its mountains in a few steps)
    zvir = c.RVGAS / c.RDGAS - 1.0
    temp = cells("T") * (1.0 + zvir * cells("sphum")) * (1.0 - q)
    cappa = c.KAPPA * (1.0 - 0.2 * q)
    pkz = np.exp(cappa * np.log(c.RDG * delp / delz * temp))  # p^cappa of the full (non-hydrostatic) pressure rho R T_v
    place("delp", delp)
    place("delz", delz)
    place("w", cells("W"))
    place("q_con", q)
    place("cappa", cappa)
    place("pt", temp / pkz)
    place("pkz", pkz)
    u = np.transpose(np.asarray(data["u"][tile], dtype=np.float64), (2, 1, 0))[ox : ox + nx, oy : oy + ny + 1, :nz]
    v = np.transpose(np.asarray(data["v"][tile], dtype=np.float64), (2, 1, 0))[ox : ox + nx + 1, oy : oy + ny, :nz]
    place("u", u, 0, 1)
    place("v", v, 1, 0)
    ptop = float(grid.ak[0])
    pe = np.concatenate([np.full(delp.shape[:2] + (1,), ptop), ptop + np.cumsum(delp, axis=2)], axis=2)
    for name, a in (("pe", pe), ("peln", np.log(pe)), ("pk", np.exp(c.KAPPA * np.log(pe)))):
        st[name][...] = np.pad(a, ((nh, nh + 1), (nh, nh + 1), (0, 0)), mode="edge")
    phis = np.zeros(shp2 + (1,))
    phis[:, :, 0] = np.pad(np.asarray(data["phis"][tile], dtype=np.float64).T[ox : ox + nx, oy : oy + ny], ((nh, nh + 1), (nh, nh + 1)), mode="edge")
    st["phis"] = phis
    return st
