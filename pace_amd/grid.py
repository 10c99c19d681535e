"""Equal-edge gnomonic cubed-sphere grid and the metric terms the acoustic path reads.

This is the *input contract* of the hot path: the attribute list of NDSL's
``GridData`` / ``DampingCoefficients`` as the reference tests name it
[REF tests/mpi_54rank/test_grid_init.py:33-120, tests/main/fv3core/test_cartesian_grid.py:44-81].
The reference obtains these from ``ndsl.grid.MetricTerms`` (un-vendored
submodule) [REF driver/pace/driver/grid.py:104-142]; here they are computed
directly from the geometry of the FV3 ``gnomonic_ed`` construction (equal
angular spacing along the cube edges, tensor-product on the face) with the
same 3-cell halo, so every operator of the path has realistic, self-consistent
metric input at any resolution.  Values inside the 3x3 cube-corner halo blocks
follow the x-direction corner fill of the grid points (a convention, as it is
in FV3); no operator on the path depends on more than their symmetry.

Arrays are numpy float64, indexed ``[i, j]`` with storage shape
``(nx + 2*n_halo + 1, ny + 2*n_halo + 1)`` for every staggering (unused last
row/column padded), matching NDSL's padded storages [SURVEY §8].
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np

from .constants import ConstantSet, get_constants
from .corners import copy_corners_index, corner_flags
from .topology import EAST, FACES, NORTH, SOUTH, WEST, CubedSpherePartitioner, edge_transform

TINY = 1.0e-8
LON_SHIFT = np.pi / 18.0  # FV3 shifts the cube by 10 degrees


# ----------------------------------------------------------------------------
# geometry helpers (unit sphere, cartesian)
# ----------------------------------------------------------------------------
def _normalize(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _xi(idx, n):
    alpha = np.arcsin(1.0 / np.sqrt(3.0))
    return np.sqrt(2.0) * np.tan(alpha * (2.0 * np.asarray(idx, dtype=np.float64) / n - 1.0))


def _rotz(v, ang):
    c, s = np.cos(ang), np.sin(ang)
    out = np.empty_like(v)
    out[..., 0] = c * v[..., 0] - s * v[..., 1]
    out[..., 1] = s * v[..., 0] + c * v[..., 1]
    out[..., 2] = v[..., 2]
    return out


def tile_point(tile, xi_idx, yj_idx, n):
    """Unit vector of the grid corner with (integer) tile coordinates (x, y)."""
    c, ex, ey = FACES[tile]
    a = _xi(xi_idx, n)[..., None]
    b = _xi(yj_idx, n)[..., None]
    p = c.astype(np.float64) + a * ex + b * ey
    return _rotz(_normalize(p), -LON_SHIFT)


def great_circle_dist(p, q):
    """Angle between unit vectors (stable for small angles)."""
    cr = np.linalg.norm(np.cross(p, q), axis=-1)
    dt = np.sum(p * q, axis=-1)
    return np.arctan2(cr, dt)


def _mid(p, q):
    return _normalize(p + q)


def cos_angle(p1, p2, p3):
    """cos of the angle at p1 between the arcs p1->p2 and p1->p3."""
    a = np.cross(p1, p2)
    b = np.cross(p1, p3)
    d = np.sum(a * a, -1) * np.sum(b * b, -1)
    out = np.sum(a * b, -1) / np.sqrt(np.where(d > 0, d, 1.0))
    return np.clip(np.where(d > 0, out, 1.0), -1.0, 1.0)


def _spherical_angle(p1, p2, p3):
    return np.arccos(cos_angle(p1, p2, p3))


def quad_area(p1, p2, p3, p4):
    """Area on the unit sphere of the quadrilateral p1-p2-p3-p4 (in order)."""
    ang = (
        _spherical_angle(p1, p2, p4)
        + _spherical_angle(p2, p3, p1)
        + _spherical_angle(p3, p4, p2)
        + _spherical_angle(p4, p1, p3)
    )
    return ang - 2.0 * np.pi


def lonlat(p):
    lon = np.arctan2(p[..., 1], p[..., 0])
    lon = np.where(lon < 0, lon + 2 * np.pi, lon)
    lat = np.arcsin(np.clip(p[..., 2], -1.0, 1.0))
    return lon, lat


# ----------------------------------------------------------------------------
# eta levels
# ----------------------------------------------------------------------------
def load_eta79() -> Tuple[np.ndarray, np.ndarray]:
    """L79 hybrid coefficients, from the committed fixture (data of
    [REF examples/notebooks/generate_eta_file_netcdf.ipynb:82-135])."""
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "eta79.npz")
    with np.load(path) as f:
        return f["ak"].astype(np.float64), f["bk"].astype(np.float64)


def make_eta(nz: int) -> Tuple[np.ndarray, np.ndarray]:
    """ak/bk for ``nz`` layers: the L79 table, or a monotone interpolation of it in
    normalised index for other level counts (no other table exists in the
    reference tree -- stated in BASELINE.md cfg-4)."""
    ak79, bk79 = load_eta79()
    if nz == 79:
        return ak79, bk79
    s79 = np.linspace(0.0, 1.0, 80)
    s = np.linspace(0.0, 1.0, nz + 1)
    p79 = ak79 + bk79 * 1.0e5
    # interpolate pressure and b separately, keep a = p - b*p0 consistent
    p = np.interp(s, s79, p79)
    bk = np.interp(s, s79, bk79)
    bk[0] = 0.0
    bk[-1] = 1.0
    ak = p - bk * 1.0e5
    ak[-1] = 0.0
    return ak, bk


# ----------------------------------------------------------------------------
# containers
# ----------------------------------------------------------------------------
_GRID_2D = (
    "dx dy dxa dya dxc dyc rdx rdy rdxa rdya rdxc rdyc area rarea area_c rarea_c "
    "cosa sina rsina cosa_u cosa_v cosa_s sina_u sina_v rsin_u rsin_v rsin2 "
    "sin_sg1 sin_sg2 sin_sg3 sin_sg4 sin_sg5 cos_sg1 cos_sg2 cos_sg3 cos_sg4 "
    "fC f0 del6_u del6_v divg_u divg_v lon lat lon_agrid lat_agrid"
).split()


@dataclass
class GridData:
    """Read-only metric terms of one rank (numpy, [i, j])."""

    nx: int
    ny: int
    nz: int
    n_halo: int
    west_edge: bool
    east_edge: bool
    south_edge: bool
    north_edge: bool
    ak: np.ndarray
    bk: np.ndarray
    ptop: float
    ks: int
    da_min: float
    da_min_c: float
    edge_w: np.ndarray
    edge_e: np.ndarray
    edge_s: np.ndarray
    edge_n: np.ndarray
    # a2b_ord4 corner extrapolation factors x1/(x2-x1), 3 per cube corner (sw, se, ne, nw)
    corner_extrap: np.ndarray
    fields: Dict[str, np.ndarray]

    def __getattr__(self, name):
        f = self.__dict__.get("fields")
        if f is not None and name in f:
            return f[name]
        raise AttributeError(name)

    @property
    def p_ref(self) -> float:
        return 1.0e5

    @property
    def dp_ref(self) -> np.ndarray:
        """dp_ref[k] = ak[k+1]-ak[k] + (bk[k+1]-bk[k])*1e5  [SURVEY A.1]."""
        return (self.ak[1:] - self.ak[:-1]) + (self.bk[1:] - self.bk[:-1]) * 1.0e5

    @property
    def pfull(self) -> np.ndarray:
        ph = self.ak + self.bk * self.p_ref
        return (ph[1:] - ph[:-1]) / np.log(ph[1:] / ph[:-1])


@dataclass
class DampingCoefficients:
    """[REF driver/pace/driver/grid.py:104-142 returns this next to GridData]."""

    del6_u: np.ndarray
    del6_v: np.ndarray
    divg_u: np.ndarray
    divg_v: np.ndarray
    da_min: float
    da_min_c: float


# ----------------------------------------------------------------------------
# generator
# ----------------------------------------------------------------------------
def _corner_positions(part: CubedSpherePartitioner, rank: int, nh: int) -> np.ndarray:
    """Unit vectors of the grid corners of a rank incl. halo, shape (nx+2nh+1, ny+2nh+1, 3)."""
    n = part.nx_tile
    tile = part.tile_index(rank)
    x0, y0 = part.origin(rank)
    gx = x0 + np.arange(-nh, part.nx + nh + 1)
    gy = y0 + np.arange(-nh, part.ny + nh + 1)
    X, Y = np.meshgrid(gx, gy, indexing="ij")
    X = X.astype(np.int64)
    Y = Y.astype(np.int64)
    # cube-corner halo blocks: x-direction fill of the grid points (FV3 fill_corners XDir, BGRID)
    sw = (X < 0) & (Y < 0)
    se = (X > n) & (Y < 0)
    ne = (X > n) & (Y > n)
    nw = (X < 0) & (Y > n)
    Xs, Ys = X.copy(), Y.copy()
    Xs[sw], Ys[sw] = Y[sw], -X[sw]
    Xs[se], Ys[se] = n - Y[se], X[se] - n
    Xs[ne], Ys[ne] = Y[ne], 2 * n - X[ne]
    Xs[nw], Ys[nw] = n - Y[nw], n + X[nw]
    P = np.zeros(X.shape + (3,))
    ox = np.where(Xs < 0, -1, np.where(Xs > n, 1, 0))
    oy = np.where(Ys < 0, -1, np.where(Ys > n, 1, 0))
    assert not ((ox != 0) & (oy != 0)).any()
    inside = (ox == 0) & (oy == 0)
    P[inside] = tile_point(tile, Xs[inside], Ys[inside], n)
    for d, m in ((WEST, ox == -1), (EAST, ox == 1), (SOUTH, oy == -1), (NORTH, oy == 1)):
        if m.any():
            tr = edge_transform(tile, d)
            x2, y2 = tr.apply(Xs[m], Ys[m], n)
            P[m] = tile_point(tr.tile, x2, y2, n)
    return P


def _pad(a, shape):
    out = np.zeros(shape, dtype=np.float64)
    out[: a.shape[0], : a.shape[1]] = a
    # replicate into the padding so reciprocals stay finite
    if a.shape[0] < shape[0]:
        out[a.shape[0] :, : a.shape[1]] = a[-1:, :]
    if a.shape[1] < shape[1]:
        out[:, a.shape[1] :] = out[:, a.shape[1] - 1 : a.shape[1]]
    return out


def make_grid(
    part: CubedSpherePartitioner,
    rank: int,
    nz: int = 79,
    n_halo: int = 3,
    constants: Optional[ConstantSet] = None,
    ak: Optional[np.ndarray] = None,
    bk: Optional[np.ndarray] = None,
    da_min: Optional[float] = None,
    da_min_c: Optional[float] = None,
) -> GridData:
    """Metric terms of one rank.  ``da_min``/``da_min_c`` are global minima
    (the one reduction of grid setup [SURVEY A.15]); when None they are taken
    from the analytic symmetry of the grid (all six tiles are congruent, so the
    minimum over tile 0 is the global minimum)."""
    c = constants or get_constants()
    R = c.RADIUS
    nh = n_halo
    nx, ny = part.nx, part.ny
    n = part.nx_tile
    x0, y0 = part.origin(rank)
    flags = part.on_tile_edges(rank)
    P = _corner_positions(part, rank, nh)
    shape = (nx + 2 * nh + 1, ny + 2 * nh + 1)
    f: Dict[str, np.ndarray] = {}

    A = _normalize(P[:-1, :-1] + P[1:, :-1] + P[:-1, 1:] + P[1:, 1:])  # cell centres (nI-1, nJ-1)
    dx = R * great_circle_dist(P[:-1, :], P[1:, :])  # (nI-1, nJ)
    dy = R * great_circle_dist(P[:, :-1], P[:, 1:])  # (nI, nJ-1)
    mW = _mid(P[:-1, :-1], P[:-1, 1:])
    mE = _mid(P[1:, :-1], P[1:, 1:])
    mS = _mid(P[:-1, :-1], P[1:, :-1])
    mN = _mid(P[:-1, 1:], P[1:, 1:])
    dxa = R * great_circle_dist(mW, mE)
    dya = R * great_circle_dist(mS, mN)
    area = R * R * quad_area(P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:])

    # sines / cosines of the grid angle at the 4 mid-edges, centre and SW corner of each cell
    cos1 = cos_angle(mW, A, P[:-1, 1:])
    cos2 = cos_angle(mS, P[1:, :-1], A)
    cos3 = cos_angle(mE, A, P[1:, :-1])
    cos4 = cos_angle(mN, P[:-1, 1:], A)
    ec1 = mE - mW
    ec1 = _normalize(ec1 - np.sum(ec1 * A, -1, keepdims=True) * A)
    ec2 = mN - mS
    ec2 = _normalize(ec2 - np.sum(ec2 * A, -1, keepdims=True) * A)
    cos5 = np.sum(ec1 * ec2, -1)
    cos6 = cos_angle(P[:-1, :-1], P[1:, :-1], P[:-1, 1:])
    cos8 = cos_angle(P[1:, 1:], P[:-1, 1:], P[1:, :-1])

    def _sin(cv):
        return np.minimum(1.0, np.sqrt(np.maximum(0.0, 1.0 - cv * cv)))

    sin = {1: _sin(cos1), 2: _sin(cos2), 3: _sin(cos3), 4: _sin(cos4), 5: _sin(cos5)}
    cos = {1: cos1, 2: cos2, 3: cos3, 4: cos4, 5: cos5}

    # Cube-corner halo blocks have no owner: fill the cell-centred terms from the
    # rotated edge-halo cells the matching sweep's copy_corners reads (x-type
    # terms from the x-sweep source, y-type from the y-sweep source; dxa<->dya
    # swap like FV3's A-grid vector fill).
    A_x, A_y = A.copy(), A.copy()
    dxa0, dya0, area0 = dxa.copy(), dya.copy(), area.copy()
    sin0 = {k: v.copy() for k, v in sin.items()}
    cos0 = {k: v.copy() for k, v in cos.items()}
    for corner, has in corner_flags(flags["west"], flags["east"], flags["south"], flags["north"]).items():
        if not has:
            continue
        di, dj, si, sj = copy_corners_index(nx, ny, nh, 1, corner)
        A_x[di, dj] = A[si, sj]
        dxa[di, dj] = dya0[si, sj]
        area[di, dj] = area0[si, sj]
        for k in sin:
            sin[k][di, dj] = sin0[k][si, sj]
            cos[k][di, dj] = cos0[k][si, sj]
        di, dj, si, sj = copy_corners_index(nx, ny, nh, 2, corner)
        A_y[di, dj] = A[si, sj]
        dya[di, dj] = dxa0[si, sj]

    dxc = np.zeros((shape[0], shape[1] - 1))
    dxc[1:-1, :] = R * great_circle_dist(A_x[:-1, :], A_x[1:, :])
    dxc[0, :] = dxc[1, :]
    dxc[-1, :] = dxc[-2, :]
    dyc = np.zeros((shape[0] - 1, shape[1]))
    dyc[:, 1:-1] = R * great_circle_dist(A_y[:, :-1], A_y[:, 1:])
    dyc[:, 0] = dyc[:, 1]
    dyc[:, -1] = dyc[:, -2]

    # dual-cell areas at the grid corners: quad of the 4 surrounding centres;
    # on a tile edge line twice the half cell on the tile side; at a cube
    # corner three times the kite (FV3 grid_area)
    area_c = np.zeros(shape)
    area_c[1:-1, 1:-1] = R * R * np.abs(quad_area(A_x[:-1, :-1], A_x[1:, :-1], A_x[1:, 1:], A_x[:-1, 1:]))
    area_c[0, :] = area_c[1, :]
    area_c[-1, :] = area_c[-2, :]
    area_c[:, 0] = area_c[:, 1]
    area_c[:, -1] = area_c[:, -2]
    gx = x0 + np.arange(-nh, nx + nh + 1)
    gy = y0 + np.arange(-nh, ny + nh + 1)
    jr = np.arange(1, shape[1] - 1)
    ir = np.arange(1, shape[0] - 1)
    for ii in np.nonzero((gx == 0) | (gx == n))[0]:
        ci = ii if gx[ii] == 0 else ii - 1  # cell column on the tile side
        p1 = _mid(P[ii, jr - 1], P[ii, jr])
        p4 = _mid(P[ii, jr], P[ii, jr + 1])
        val = 2.0 * R * R * np.abs(quad_area(p1, A[ci, jr - 1], A[ci, jr], p4))
        keep = ~((gy[jr] == 0) | (gy[jr] == n))
        area_c[ii, jr[keep]] = val[keep]
    for jj in np.nonzero((gy == 0) | (gy == n))[0]:
        cj = jj if gy[jj] == 0 else jj - 1
        p1 = _mid(P[ir - 1, jj], P[ir, jj])
        p4 = _mid(P[ir, jj], P[ir + 1, jj])
        val = 2.0 * R * R * np.abs(quad_area(p1, A[ir - 1, cj], A[ir, cj], p4))
        keep = ~((gx[ir] == 0) | (gx[ir] == n))
        area_c[ir[keep], jj] = val[keep]
    for ii in np.nonzero((gx == 0) | (gx == n))[0]:
        for jj in np.nonzero((gy == 0) | (gy == n))[0]:
            xg, yg = gx[ii], gy[jj]
            ci = ii if xg == 0 else ii - 1
            cj = jj if yg == 0 else jj - 1
            px = _mid(P[ii, jj], P[ii + (1 if xg == 0 else -1), jj])
            py = _mid(P[ii, jj], P[ii, jj + (1 if yg == 0 else -1)])
            area_c[ii, jj] = 3.0 * R * R * abs(quad_area(P[ii, jj], px, A[ci, cj], py))

    # corner-halo overrides "for transport operation" (FV3 grid_utils_init): the
    # corner-halo cells touching the tile take the rotated edge-halo values.
    # Local Fortran index f -> storage index f + nh - 1, with npx = nx + 1.
    o = nh - 1
    npx, npy = nx + 1, ny + 1
    if flags["west"] and flags["south"]:
        for i in range(-2, 1):
            sin[3][0 + o, i + o] = sin[2][i + o, 1 + o]
            sin[4][i + o, 0 + o] = sin[1][1 + o, i + o]
    if flags["west"] and flags["north"]:
        for i in range(npy, npy + 3):
            sin[3][0 + o, i + o] = sin[4][npy - i + o, npy - 1 + o]
        for i in range(-2, 1):
            sin[2][i + o, npy + o] = sin[1][1 + o, npy - i + o]
    if flags["east"] and flags["south"]:
        for j in range(-2, 1):
            sin[1][npx + o, j + o] = sin[2][npx - j + o, 1 + o]
        for i in range(npx, npx + 3):
            sin[4][i + o, 0 + o] = sin[3][npx - 1 + o, npx - i + o]
    if flags["east"] and flags["north"]:
        for d in range(3):  # distance from the corner (local indices: nx != ny on non-square sub-domains)
            sin[1][npx + o, npy + d + o] = sin[4][npx + d + o, npy - 1 + o]
            sin[2][npx + d + o, npy + o] = sin[3][npx - 1 + o, npy + d + o]

    cosa_u = np.zeros((shape[0], shape[1] - 1))
    sina_u = np.ones((shape[0], shape[1] - 1))
    cosa_u[1:-1] = 0.5 * (cos[3][:-1] + cos[1][1:])
    sina_u[1:-1] = 0.5 * (sin[3][:-1] + sin[1][1:])
    cosa_u[0], sina_u[0] = cos[1][0], sin[1][0]
    cosa_u[-1], sina_u[-1] = cos[3][-1], sin[3][-1]
    cosa_v = np.zeros((shape[0] - 1, shape[1]))
    sina_v = np.ones((shape[0] - 1, shape[1]))
    cosa_v[:, 1:-1] = 0.5 * (cos[4][:, :-1] + cos[2][:, 1:])
    sina_v[:, 1:-1] = 0.5 * (sin[4][:, :-1] + sin[2][:, 1:])
    cosa_v[:, 0], sina_v[:, 0] = cos[2][:, 0], sin[2][:, 0]
    cosa_v[:, -1], sina_v[:, -1] = cos[4][:, -1], sin[4][:, -1]
    rsin_u = 1.0 / np.maximum(TINY, sina_u**2)
    rsin_v = 1.0 / np.maximum(TINY, sina_v**2)
    # tile edges: 1/sin instead of 1/sin^2 (FV3 "set special sin values at edges")
    if flags["west"]:
        rsin_u[nh, :] = 1.0 / np.maximum(TINY, sina_u[nh, :])
    if flags["east"]:
        rsin_u[nh + nx, :] = 1.0 / np.maximum(TINY, sina_u[nh + nx, :])
    if flags["south"]:
        rsin_v[:, nh] = 1.0 / np.maximum(TINY, sina_v[:, nh])
    if flags["north"]:
        rsin_v[:, nh + ny] = 1.0 / np.maximum(TINY, sina_v[:, nh + ny])

    cosa = np.zeros(shape)
    cosa[:-1, :-1] = cos6
    cosa[-1, 1:] = np.concatenate([cos8[-1, :]])
    cosa[1:, -1] = cos8[:, -1]
    cosa[-1, 0] = cosa[-2, 0]
    cosa[0, -1] = cosa[0, -2]
    sina = _sin(cosa)
    rsina = 1.0 / np.maximum(TINY, sina**2)

    lon_c, lat_c = lonlat(P)
    lon_a, lat_a = lonlat(A)
    fC = 2.0 * c.OMEGA * np.sin(lat_c)
    f0 = 2.0 * c.OMEGA * np.sin(lat_a)

    divg_u = sina_v * dyc / dx
    del6_u = sina_v * dx / dyc
    divg_v = sina_u * dxc / dy
    del6_v = sina_u * dy / dxc

    def put(name, a):
        f[name] = _pad(np.asarray(a, dtype=np.float64), shape)

    put("dx", dx), put("dy", dy), put("dxa", dxa), put("dya", dya), put("dxc", dxc), put("dyc", dyc)
    put("area", area), put("area_c", area_c)
    for k in ("dx", "dy", "dxa", "dya", "dxc", "dyc", "area", "area_c"):
        f["r" + k] = 1.0 / f[k]
    put("cosa", cosa), put("sina", sina), put("rsina", rsina)
    put("cosa_u", cosa_u), put("cosa_v", cosa_v), put("cosa_s", cos[5])
    put("sina_u", sina_u), put("sina_v", sina_v), put("rsin_u", rsin_u), put("rsin_v", rsin_v)
    put("rsin2", 1.0 / np.maximum(TINY, sin[5] ** 2))
    for k in range(1, 5):
        put(f"sin_sg{k}", sin[k])
        put(f"cos_sg{k}", cos[k])
    put("sin_sg5", sin[5])  # cell centre (tracer_2d_1l's Courant-number bound)
    put("fC", fC), put("f0", f0)
    put("del6_u", del6_u), put("del6_v", del6_v), put("divg_u", divg_u), put("divg_v", divg_v)
    put("lon", lon_c), put("lat", lat_c), put("lon_agrid", lon_a), put("lat_agrid", lat_a)

    # a2b_ord4 edge factors (FV3 edge_factors): linear weights along the tile edge
    def edge_factor_x(ii):  # edge line at storage corner column ii; varies along j
        w = np.zeros(shape[1])
        py = _mid(A[ii - 1, :], A[ii, :])  # (nJ-1,) points on the edge, one per cell row
        d1 = great_circle_dist(py[:-1], P[ii, 1:-1])
        d2 = great_circle_dist(py[1:], P[ii, 1:-1])
        w[1:-1] = d2 / (d1 + d2)
        return w

    def edge_factor_y(jj):
        w = np.zeros(shape[0])
        px = _mid(A[:, jj - 1], A[:, jj])
        d1 = great_circle_dist(px[:-1], P[1:-1, jj])
        d2 = great_circle_dist(px[1:], P[1:-1, jj])
        w[1:-1] = d2 / (d1 + d2)
        return w

    edge_w = edge_factor_x(nh) if flags["west"] else np.zeros(shape[1])
    edge_e = edge_factor_x(nh + nx) if flags["east"] else np.zeros(shape[1])
    edge_s = edge_factor_y(nh) if flags["south"] else np.zeros(shape[0])
    edge_n = edge_factor_y(nh + ny) if flags["north"] else np.zeros(shape[0])

    # a2b_ord4 cube-corner extrapolation (FV3 extrap_corner): q1 + x1/(x2-x1)*(q1-q2)
    corner_extrap = np.zeros((4, 3))

    def _fac(p0, p1, p2):
        x1 = great_circle_dist(p1, p0)
        x2 = great_circle_dist(p2, p0)
        return x1 / (x2 - x1)

    is_, ie_, js_, je_ = nh, nh + nx - 1, nh, nh + ny - 1
    if flags["west"] and flags["south"]:
        p0 = P[is_, js_]
        corner_extrap[0] = [
            _fac(p0, A[is_, js_], A[is_ + 1, js_ + 1]),
            _fac(p0, A[is_ - 1, js_], A[is_ - 2, js_ + 1]),
            _fac(p0, A[is_, js_ - 1], A[is_ + 1, js_ - 2]),
        ]
    if flags["east"] and flags["south"]:
        p0 = P[ie_ + 1, js_]
        corner_extrap[1] = [
            _fac(p0, A[ie_, js_], A[ie_ - 1, js_ + 1]),
            _fac(p0, A[ie_, js_ - 1], A[ie_ - 1, js_ - 2]),
            _fac(p0, A[ie_ + 1, js_], A[ie_ + 2, js_ + 1]),
        ]
    if flags["east"] and flags["north"]:
        p0 = P[ie_ + 1, je_ + 1]
        corner_extrap[2] = [
            _fac(p0, A[ie_, je_], A[ie_ - 1, je_ - 1]),
            _fac(p0, A[ie_ + 1, je_], A[ie_ + 2, je_ - 1]),
            _fac(p0, A[ie_, je_ + 1], A[ie_ - 1, je_ + 2]),
        ]
    if flags["west"] and flags["north"]:
        p0 = P[is_, je_ + 1]
        corner_extrap[3] = [
            _fac(p0, A[is_, je_], A[is_ + 1, je_ - 1]),
            _fac(p0, A[is_ - 1, je_], A[is_ - 2, je_ - 1]),
            _fac(p0, A[is_, je_ + 1], A[is_ + 1, je_ + 2]),
        ]

    if ak is None or bk is None:
        ak, bk = make_eta(nz)
    ks = int(np.sum(bk[1:] == 0.0))  # number of pure-pressure layers
    if da_min is None or da_min_c is None:
        gm = global_area_minima(part.nx_tile, c, nh)
        da_min = gm[0] if da_min is None else da_min
        da_min_c = gm[1] if da_min_c is None else da_min_c
    return GridData(
        nx=nx,
        ny=ny,
        nz=nz,
        n_halo=nh,
        west_edge=flags["west"],
        east_edge=flags["east"],
        south_edge=flags["south"],
        north_edge=flags["north"],
        ak=np.asarray(ak, dtype=np.float64),
        bk=np.asarray(bk, dtype=np.float64),
        ptop=float(ak[0]),
        ks=ks,
        da_min=float(da_min),
        da_min_c=float(da_min_c),
        edge_w=edge_w,
        edge_e=edge_e,
        edge_s=edge_s,
        edge_n=edge_n,
        corner_extrap=corner_extrap,
        fields=f,
    )


_AREA_MIN_CACHE: Dict[Tuple[int, str, int], Tuple[float, float]] = {}


def global_area_minima(nx_tile: int, c: ConstantSet, nh: int = 3) -> Tuple[float, float]:
    """(da_min, da_min_c): global minima of ``area`` and ``area_c`` over compute
    domains.  All six tiles are congruent, so tile 0 alone decides; computed
    once per resolution from a 1x1-layout tile with explicit values (not by
    recursion into make_grid's default)."""
    key = (nx_tile, c.name, nh)
    if key not in _AREA_MIN_CACHE:
        # cell areas of the equal-edge gnomonic grid shrink monotonically towards the cube
        # corners, so a small corner patch holds both minima (checked against the full tile
        # in tests/test_grid.py); avoids a 775^2 setup at C768.
        div = 1
        for d in range(nx_tile // 8, 0, -1):
            if nx_tile % d == 0 and nx_tile // d >= 8:
                div = d
                break
        part = CubedSpherePartitioner(nx_tile, (div, div))
        n = part.nx
        g = make_grid(part, 0, nz=1, n_halo=nh, constants=c, ak=np.array([1.0, 0.0]), bk=np.array([0.0, 1.0]), da_min=1.0, da_min_c=1.0)
        a = g.area[nh : nh + n, nh : nh + n]
        ac = g.area_c[nh : nh + n + 1, nh : nh + n + 1]
        _AREA_MIN_CACHE[key] = (float(a.min()), float(ac.min()))
    return _AREA_MIN_CACHE[key]


def damping_coefficients(g: GridData) -> DampingCoefficients:
    return DampingCoefficients(g.del6_u, g.del6_v, g.divg_u, g.divg_v, g.da_min, g.da_min_c)
