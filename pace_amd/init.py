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
