"""Cubed-sphere topology, rank partitioning and halo gather maps.

Host-side mirror of the slice of NDSL the acoustic path needs
(``TilePartitioner`` / ``CubedSpherePartitioner`` / boundary slices and
rotations) [REF driver/pace/driver/driver.py:419-430,
docs/util/communication.rst:20-109].  NDSL itself is an un-vendored submodule,
so nothing here is transcribed: tile adjacency, rotations and the vector sign
rules are *derived from geometry* (six cube faces with explicit local frames)
instead of being tabulated, which removes a whole class of convention errors.

Conventions
-----------
* tile ``t`` (0-based; FV3 tile ``t+1``) has a centre normal ``c`` and local
  axes ``ex``/``ey`` on the cube; FV3's odd/even adjacency rule (odd tiles:
  E->t+1, N->t+2 rotated, W->t-2 rotated, S->t-1; even tiles: E->t+2 rotated,
  N->t+1, W->t-1, S->t-2 rotated; 1-based) falls out of these frames and is
  asserted in ``tests/test_topology.py``.
* continuous tile coordinates ``(x, y) in [0, N]^2``: a cell-centred storage
  index ``i`` sits at ``x = x0 + (i - n_halo) + 0.5``, an interface index at
  ``x = x0 + (i - n_halo)``.  Crossing a tile edge is an affine map
  ``(x', y') = R (x, y) + N s`` with ``R`` a signed permutation.
* rank = tile * (lx*ly) + sy * lx + sx  [REF driver/pace/driver/grid.py:241-244].
"""
from __future__ import annotations

from dataclasses import dataclass
from functools import lru_cache
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_X = np.array([1, 0, 0])
_Y = np.array([0, 1, 0])
_Z = np.array([0, 0, 1])

# (centre normal, local x axis, local y axis) per tile
FACES: Tuple[Tuple[np.ndarray, np.ndarray, np.ndarray], ...] = (
    (_X, _Y, _Z),
    (_Y, -_X, _Z),
    (_Z, -_X, -_Y),
    (-_X, -_Z, -_Y),
    (-_Y, -_Z, _X),
    (-_Z, _Y, _X),
)

WEST, EAST, SOUTH, NORTH = 0, 1, 2, 3
_OUTWARD = {WEST: (-1, 0), EAST: (1, 0), SOUTH: (0, -1), NORTH: (0, 1)}
_EDGE_POINT = {WEST: (0.0, 0.0), EAST: (1.0, 0.0), SOUTH: (0.0, 0.0), NORTH: (0.0, 1.0)}


@dataclass(frozen=True)
class EdgeTransform:
    """Affine map from tile coordinates to the coordinates of the tile across one edge."""

    tile: int
    R: Tuple[Tuple[int, int], Tuple[int, int]]
    s: Tuple[int, int]  # shift in units of N

    def apply(self, x, y, n):
        R = self.R
        return (
            R[0][0] * x + R[0][1] * y + n * self.s[0],
            R[1][0] * x + R[1][1] * y + n * self.s[1],
        )

    @property
    def n_clockwise_rotations(self) -> int:
        """How many quarter turns clockwise the *data* must be rotated to land in the neighbour frame."""
        R = self.R
        if R == ((1, 0), (0, 1)):
            return 0
        if R == ((0, 1), (-1, 0)):
            return 1
        if R == ((-1, 0), (0, -1)):
            return 2
        if R == ((0, -1), (1, 0)):
            return 3
        raise ValueError("not a rotation")


@lru_cache(maxsize=None)
def edge_transform(tile: int, direction: int) -> EdgeTransform:
    c, ex, ey = FACES[tile]
    o = np.array(_OUTWARD[direction])
    a = np.array([-o[1], o[0]])  # along-edge direction
    out3 = o[0] * ex + o[1] * ey
    other = None
    for t2, (c2, _, _) in enumerate(FACES):
        if np.array_equal(c2, out3):
            other = t2
    assert other is not None
    c2, ex2, ey2 = FACES[other]
    a3 = a[0] * ex + a[1] * ey
    o2 = np.array([-(c @ ex2), -(c @ ey2)])  # inward direction of the neighbour
    a2 = np.array([a3 @ ex2, a3 @ ey2])
    # R [o a] = [o2 a2]
    M = np.stack([o, a], axis=1)
    M2 = np.stack([o2, a2], axis=1)
    R = M2 @ np.linalg.inv(M)
    R = np.rint(R).astype(int)
    # shift from one shared point (unit tile coordinates)
    p = np.array(_EDGE_POINT[direction])
    P3 = c + (2 * p[0] - 1) * ex + (2 * p[1] - 1) * ey
    p2 = np.array([((P3 - c2) @ ex2 + 1) / 2.0, ((P3 - c2) @ ey2 + 1) / 2.0])
    s = np.rint(p2 - R @ p).astype(int)
    return EdgeTransform(
        tile=other,
        R=((int(R[0, 0]), int(R[0, 1])), (int(R[1, 0]), int(R[1, 1]))),
        s=(int(s[0]), int(s[1])),
    )


@dataclass(frozen=True)
class TilePartitioner:
    """Sub-tile decomposition of one tile [REF docs/util/communication.rst:20-41]."""

    layout: Tuple[int, int]

    @property
    def ranks_per_tile(self) -> int:
        return self.layout[0] * self.layout[1]

    def subtile_index(self, rank: int) -> Tuple[int, int]:
        """(sx, sy) of a rank inside its tile."""
        r = rank % self.ranks_per_tile
        return r % self.layout[0], r // self.layout[0]


@dataclass(frozen=True)
class CubedSpherePartitioner:
    """6 tiles x layout sub-tiles [REF driver/pace/driver/driver.py:419-424]."""

    nx_tile: int
    layout: Tuple[int, int] = (1, 1)

    def __post_init__(self):
        lx, ly = self.layout
        if self.nx_tile % lx or self.nx_tile % ly:
            raise ValueError(f"nx_tile={self.nx_tile} not divisible by layout {self.layout}")
        if min(self.nx, self.ny) < 3:
            raise ValueError("each rank needs at least n_halo=3 cells per side")

    @property
    def tile(self) -> TilePartitioner:
        return TilePartitioner(tuple(self.layout))

    @property
    def total_ranks(self) -> int:
        return 6 * self.layout[0] * self.layout[1]

    @property
    def nx(self) -> int:
        return self.nx_tile // self.layout[0]

    @property
    def ny(self) -> int:
        return self.nx_tile // self.layout[1]

    def tile_index(self, rank: int) -> int:
        return rank // self.tile.ranks_per_tile

    def subtile_index(self, rank: int) -> Tuple[int, int]:
        return self.tile.subtile_index(rank)

    def rank_of(self, tile: int, sx: int, sy: int) -> int:
        return tile * self.tile.ranks_per_tile + sy * self.layout[0] + sx

    def origin(self, rank: int) -> Tuple[int, int]:
        sx, sy = self.subtile_index(rank)
        return sx * self.nx, sy * self.ny

    def on_tile_edges(self, rank: int) -> Dict[str, bool]:
        """south/north/west/east_edge flags of GridIndexing [REF tests/main/fv3core/test_grid.py:56-101]."""
        sx, sy = self.subtile_index(rank)
        return {
            "west": sx == 0,
            "east": sx == self.layout[0] - 1,
            "south": sy == 0,
            "north": sy == self.layout[1] - 1,
        }


# ----------------------------------------------------------------------------
# gather maps
# ----------------------------------------------------------------------------

CELL = 0.5
IFACE = 0.0

STAGGER = {
    "cell": (CELL, CELL),
    "corner": (IFACE, IFACE),  # B-grid scalar (divgd)
    "dgrid_u": (CELL, IFACE),  # u: (x, y_interface)
    "dgrid_v": (IFACE, CELL),  # v: (x_interface, y)
    "cgrid_u": (IFACE, CELL),  # uc: (x_interface, y)
    "cgrid_v": (CELL, IFACE),  # vc: (x, y_interface)
}


@dataclass
class GatherMap:
    """Halo points of one rank and where each one comes from.

    All arrays have one entry per halo point that has a source (cube-corner
    points have none and are left untouched, like the reference).
    ``dst_flat``/``src_flat`` index the horizontal plane ``j * ni_alloc + i``.
    """

    dst_comp: np.ndarray
    dst_flat: np.ndarray
    src_rank: np.ndarray
    src_comp: np.ndarray
    src_flat: np.ndarray
    sign: np.ndarray

    def __len__(self):
        return len(self.dst_flat)


def _extent(n: int, off: float) -> int:
    return n + (1 if off == IFACE else 0)


def _locate(part: CubedSpherePartitioner, tile: int, x, y, xn, yn, evec):
    """Map points (x, y) of ``tile`` (possibly outside [0,N]^2) to their owner.

    ``xn, yn`` are nudged copies used only to choose the tile/rank.  ``evec``
    is the component direction (unit 2-vector) or None for scalars.  Returns
    valid mask, owner tile, mapped (x', y'), mapped nudged coords, mapped
    direction vectors.
    """
    n = part.nx_tile
    ox = np.where(xn < 0, -1, np.where(xn > n, 1, 0))
    oy = np.where(yn < 0, -1, np.where(yn > n, 1, 0))
    valid = ~((ox != 0) & (oy != 0))
    t_out = np.full(x.shape, tile, dtype=np.int64)
    x2, y2, xn2, yn2 = x.copy(), y.copy(), xn.copy(), yn.copy()
    e0 = np.full(x.shape, 0 if evec is None else evec[0], dtype=np.int64)
    e1 = np.full(x.shape, 0 if evec is None else evec[1], dtype=np.int64)
    for direction, mask in (
        (WEST, (ox == -1) & (oy == 0)),
        (EAST, (ox == 1) & (oy == 0)),
        (SOUTH, (oy == -1) & (ox == 0)),
        (NORTH, (oy == 1) & (ox == 0)),
    ):
        if not mask.any():
            continue
        tr = edge_transform(tile, direction)
        t_out[mask] = tr.tile
        x2[mask], y2[mask] = tr.apply(x[mask], y[mask], n)
        xn2[mask], yn2[mask] = tr.apply(xn[mask], yn[mask], n)
        if evec is not None:
            R = tr.R
            e0[mask] = R[0][0] * evec[0] + R[0][1] * evec[1]
            e1[mask] = R[1][0] * evec[0] + R[1][1] * evec[1]
    return valid, t_out, x2, y2, xn2, yn2, e0, e1


def _finish_map(part, n_halo_alloc, staggers, comp, di, dj, valid, t2, x2, y2, xn2, yn2, e0, e1, ni_alloc):
    lx, ly = part.layout
    nx, ny = part.nx, part.ny
    sx2 = np.clip(np.floor(xn2 / nx).astype(np.int64), 0, lx - 1)
    sy2 = np.clip(np.floor(yn2 / ny).astype(np.int64), 0, ly - 1)
    src_rank = t2 * (lx * ly) + sy2 * lx + sx2
    if len(staggers) == 1:
        src_comp = np.zeros(x2.shape, dtype=np.int64)
        sign = np.ones(x2.shape, dtype=np.int64)
    else:
        src_comp = np.where(e0 != 0, 0, 1)
        sign = np.where(e0 != 0, e0, e1)
    offx = np.array([st[0] for st in staggers])[src_comp]
    offy = np.array([st[1] for st in staggers])[src_comp]
    fi = x2 - sx2 * nx - offx + n_halo_alloc
    fj = y2 - sy2 * ny - offy + n_halo_alloc
    si = np.rint(fi).astype(np.int64)
    sj = np.rint(fj).astype(np.int64)
    if valid.any():
        assert np.allclose(fi[valid], si[valid]) and np.allclose(fj[valid], sj[valid]), "staggering mismatch"
        # sources must be compute-domain points of their owner
        ex_src = np.array([_extent(nx, st[0]) for st in staggers])[src_comp]
        ey_src = np.array([_extent(ny, st[1]) for st in staggers])[src_comp]
        ok = (
            (si[valid] >= n_halo_alloc)
            & (si[valid] < n_halo_alloc + ex_src[valid])
            & (sj[valid] >= n_halo_alloc)
            & (sj[valid] < n_halo_alloc + ey_src[valid])
        )
        assert ok.all(), "halo source outside the owner's compute domain"
    v = valid
    return GatherMap(
        dst_comp=np.full(int(v.sum()), comp, dtype=np.int32),
        dst_flat=(dj[v] * ni_alloc + di[v]).astype(np.int64),
        src_rank=src_rank[v].astype(np.int32),
        src_comp=src_comp[v].astype(np.int32),
        src_flat=(sj[v] * ni_alloc + si[v]).astype(np.int64),
        sign=sign[v].astype(np.int8),
    )


def _concat(maps: Sequence[GatherMap]) -> GatherMap:
    return GatherMap(
        *(np.concatenate([getattr(m, f) for m in maps]) for f in ("dst_comp", "dst_flat", "src_rank", "src_comp", "src_flat", "sign"))
    )


def build_halo_map(
    part: CubedSpherePartitioner,
    rank: int,
    staggers: Sequence[Tuple[float, float]],
    n_halo: int = 3,
    n_halo_alloc: int = 3,
    ni_alloc: Optional[int] = None,
) -> GatherMap:
    """Gather map that fills the ``n_halo``-wide halo of one rank.

    ``staggers`` has one entry for a scalar, two (x-component, y-component)
    for a vector pair; a vector's components swap and change sign across
    rotated tile edges exactly as the directed edge / face normal they live on
    does (the rule NDSL tabulates by ``n_clockwise_rotations``
    [REF docs/util/communication.rst:52]).
    """
    if len(staggers) == 1 and staggers[0][0] != staggers[0][1]:
        raise ValueError("a scalar with asymmetric staggering cannot cross rotated tile edges")
    if len(staggers) == 2 and (staggers[0][0], staggers[0][1]) != (staggers[1][1], staggers[1][0]):
        raise ValueError("vector components must have mirrored staggering")
    nx, ny = part.nx, part.ny
    if ni_alloc is None:
        ni_alloc = nx + 2 * n_halo_alloc + 1
    tile = part.tile_index(rank)
    x0, y0 = part.origin(rank)
    eps = 1e-3
    maps = []
    for comp, (offx, offy) in enumerate(staggers):
        ex_, ey_ = _extent(nx, offx), _extent(ny, offy)
        ii = np.arange(n_halo_alloc - n_halo, n_halo_alloc + ex_ + n_halo)
        jj = np.arange(n_halo_alloc - n_halo, n_halo_alloc + ey_ + n_halo)
        di, dj = np.meshgrid(ii, jj, indexing="ij")
        di, dj = di.ravel(), dj.ravel()
        side_x = np.where(di < n_halo_alloc, -1, np.where(di >= n_halo_alloc + ex_, 1, 0))
        side_y = np.where(dj < n_halo_alloc, -1, np.where(dj >= n_halo_alloc + ey_, 1, 0))
        halo = (side_x != 0) | (side_y != 0)
        di, dj, side_x, side_y = di[halo], dj[halo], side_x[halo], side_y[halo]
        x = x0 + (di - n_halo_alloc) + offx
        y = y0 + (dj - n_halo_alloc) + offy
        # nudge: inside our closed extent -> toward our interior; outside -> toward us
        xn = np.where(side_x == 0, np.clip(x, x0 + eps, x0 + nx - eps), x - side_x * eps)
        yn = np.where(side_y == 0, np.clip(y, y0 + eps, y0 + ny - eps), y - side_y * eps)
        evec = None if len(staggers) == 1 else ((1, 0) if comp == 0 else (0, 1))
        res = _locate(part, tile, x.astype(float), y.astype(float), xn.astype(float), yn.astype(float), evec)
        maps.append(_finish_map(part, n_halo_alloc, staggers, comp, di, dj, *res, ni_alloc))
    return _concat(maps)


def build_interface_sync_map(
    part: CubedSpherePartitioner,
    rank: int,
    staggers: Sequence[Tuple[float, float]],
    n_halo_alloc: int = 3,
    ni_alloc: Optional[int] = None,
) -> GatherMap:
    """Map for ``synchronize_vector_interfaces``: the shared interface points on
    this rank's north / east compute boundary are overwritten by the other
    owner's south / west values (rotated where a tile edge is crossed)
    [REF docs/util/communication.rst:169-176; SURVEY A.14].
    """
    assert len(staggers) == 2
    nx, ny = part.nx, part.ny
    if ni_alloc is None:
        ni_alloc = nx + 2 * n_halo_alloc + 1
    tile = part.tile_index(rank)
    x0, y0 = part.origin(rank)
    eps = 1e-3
    maps = []
    for comp, (offx, offy) in enumerate(staggers):
        ex_, ey_ = _extent(nx, offx), _extent(ny, offy)
        pts = []
        if offx == IFACE:  # east boundary column
            jj = np.arange(n_halo_alloc, n_halo_alloc + ey_)
            pts.append((np.full(jj.shape, n_halo_alloc + nx), jj, 1, 0))
        if offy == IFACE:  # north boundary row
            ii = np.arange(n_halo_alloc, n_halo_alloc + ex_)
            pts.append((ii, np.full(ii.shape, n_halo_alloc + ny), 0, 1))
        for di, dj, sxn, syn in pts:
            x = (x0 + (di - n_halo_alloc) + offx).astype(float)
            y = (y0 + (dj - n_halo_alloc) + offy).astype(float)
            xn = np.where(sxn == 0, np.clip(x, x0 + eps, x0 + nx - eps), x + eps)
            yn = np.where(syn == 0, np.clip(y, y0 + eps, y0 + ny - eps), y + eps)
            evec = (1, 0) if comp == 0 else (0, 1)
            res = _locate(part, tile, x, y, xn, yn, evec)
            maps.append(_finish_map(part, n_halo_alloc, staggers, comp, di, dj, *res, ni_alloc))
    return _concat(maps)
