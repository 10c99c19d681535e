"""ORACLE (test infrastructure, never the shipped path) -- shared helpers.

CPU restatement in whole-array numpy/fp64 of the FV3 acoustic dynamics that the
reference reaches through ``pyFV3.stencils.dyn_core.AcousticDynamics``
[REF driver/pace/driver/driver.py:494-504,641; tests/savepoint/thresholds/fv_dynamics.yaml:2-170].

PARITY UNPINNED: pyFV3 / NDSL are un-vendored git submodules (``.gitmodules``:
branch ``develop``, no SHA; directories empty) and cannot be imported here, and
the reference's golden savepoints live in an external bucket
[REF Makefile.data_download:2-16].  The algorithm below is therefore restated
from the published FV3 formulation (Lin 2004; Putman & Lin 2007; Harris et al.
2021 GFDL tech memo; GFDL_atmos_cubed_sphere ``sw_core/tp_core/nh_core/nh_utils/
a2b_edge/dyn_core``) anchored on the reference's call sites, variable names and
configs.  It is pinned only by internal properties (conservation, rotation /
transposition symmetry, decomposition identity) -- see DESIGN.md.

Index convention: arrays are ``[i, j(, k)]`` with ``n_halo`` ghost cells; code is
written in *local Fortran numbering* (first compute cell 1, last ``nx``;
``npx = nx + 1``) and ``Dom.sl`` turns inclusive Fortran ranges into slices, so
every loop bound can be read against the Fortran / pyFV3 originals.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

# transposition partner of every metric term (i <-> j)
_SWAP = {
    "dx": "dy", "dxa": "dya", "dxc": "dyc", "rdx": "rdy", "rdxa": "rdya", "rdxc": "rdyc",
    "cosa_u": "cosa_v", "sina_u": "sina_v", "rsin_u": "rsin_v",
    "sin_sg1": "sin_sg2", "sin_sg3": "sin_sg4", "cos_sg1": "cos_sg2", "cos_sg3": "cos_sg4",
    "del6_u": "del6_v", "divg_u": "divg_v", "lon": "lon", "lat": "lat",
}  # fmt: skip
_SWAP.update({v: k for k, v in list(_SWAP.items())})


class Dom:
    """Index space + metric terms of one rank."""

    def __init__(self, grid, consts, _transposed_from=None):
        self.grid = grid
        self.c = consts
        self.nh = grid.n_halo
        self.o = self.nh - 1
        if _transposed_from is None:
            self.nx, self.ny = grid.nx, grid.ny
            self.west, self.east = grid.west_edge, grid.east_edge
            self.south, self.north = grid.south_edge, grid.north_edge
            self.m = SimpleNamespace(**{k: v[:, :, None] for k, v in grid.fields.items()})
            self.edge_w, self.edge_e, self.edge_s, self.edge_n = grid.edge_w, grid.edge_e, grid.edge_s, grid.edge_n
            self._T = None
        else:
            d = _transposed_from
            self.nx, self.ny = d.ny, d.nx
            self.west, self.east, self.south, self.north = d.south, d.north, d.west, d.east
            mm = {}
            for k, v in vars(d.m).items():
                mm[_SWAP.get(k, k)] = v.transpose(1, 0, 2)
            self.m = SimpleNamespace(**mm)
            self.edge_w, self.edge_e, self.edge_s, self.edge_n = d.edge_s, d.edge_n, d.edge_w, d.edge_e
            self._T = d
        self.nz = grid.nz
        self.npx, self.npy = self.nx + 1, self.ny + 1
        self.is_, self.ie, self.js, self.je = 1, self.nx, 1, self.ny
        self.isd, self.ied = 1 - self.nh, self.nx + self.nh
        self.jsd, self.jed = 1 - self.nh, self.ny + self.nh
        self.sw = self.west and self.south
        self.se = self.east and self.south
        self.ne = self.east and self.north
        self.nw = self.west and self.north

    @property
    def T(self) -> "Dom":
        """The same rank seen with i and j exchanged."""
        if self._T is None:
            self._T = Dom(self.grid, self.c, _transposed_from=self)
        return self._T

    def sl(self, i0, i1, j0, j1):
        o = self.o
        return (slice(i0 + o, i1 + o + 1), slice(j0 + o, j1 + o + 1))

    def shape2(self):
        return (self.nx + 2 * self.nh + 1, self.ny + 2 * self.nh + 1)


def tr(a):
    """Transpose the horizontal axes of a 2-D or 3-D field (view)."""
    return a.transpose(1, 0, 2) if a.ndim == 3 else a.T


# ---------------------------------------------------------------------------
# cube-corner fills  [SURVEY A.13; FV3 copy_corners / fill_4corners / fill_corners]
# ---------------------------------------------------------------------------
def copy_corners(D: Dom, q, direction: int):
    """Cell-centred corner fill for an x (1) or y (2) sweep; in place."""
    o, nh, npx, npy = D.o, D.nh, D.npx, D.npy
    lo = range(1 - nh, 1)
    if direction == 1:
        if D.sw:
            for j in lo:
                for i in lo:
                    q[i + o, j + o] = q[j + o, 1 - i + o]
        if D.se:
            for j in lo:
                for i in range(npx, npx + nh):
                    q[i + o, j + o] = q[npx - j + o, i - npx + 1 + o]
        if D.ne:
            for j in range(npy, npy + nh):
                for i in range(npx, npx + nh):
                    q[i + o, j + o] = q[npx + (j - npy) + o, npy - 1 - (i - npx) + o]
        if D.nw:
            for j in range(npy, npy + nh):
                for i in lo:
                    q[i + o, j + o] = q[npy - j + o, npy - 1 + i + o]
    else:
        if D.sw:
            for j in lo:
                for i in lo:
                    q[i + o, j + o] = q[1 - j + o, i + o]
        if D.se:
            for j in lo:
                for i in range(npx, npx + nh):
                    q[i + o, j + o] = q[npx - 1 + j + o, npx - i + o]
        if D.ne:
            for j in range(npy, npy + nh):
                for i in range(npx, npx + nh):
                    q[i + o, j + o] = q[npx - 1 - (j - npy) + o, npy + (i - npx) + o]
        if D.nw:
            for j in range(npy, npy + nh):
                for i in lo:
                    q[i + o, j + o] = q[j - npy + 1 + o, npy - i + o]


def fill_4corners(D: Dom, q, direction: int):
    """Two-cell corner fill used by c_sw / update_dz_c (FV3 fill_4corners); in place."""
    o, npx, npy = D.o, D.npx, D.npy

    def Q(i, j):
        return (i + o, j + o)

    if direction == 1:
        if D.sw:
            q[Q(-1, 0)] = q[Q(0, 2)]
            q[Q(0, 0)] = q[Q(0, 1)]
        if D.se:
            q[Q(npx + 1, 0)] = q[Q(npx, 2)]
            q[Q(npx, 0)] = q[Q(npx, 1)]
        if D.ne:
            q[Q(npx, npy)] = q[Q(npx, npy - 1)]
            q[Q(npx + 1, npy)] = q[Q(npx, npy - 2)]
        if D.nw:
            q[Q(0, npy)] = q[Q(0, npy - 1)]
            q[Q(-1, npy)] = q[Q(0, npy - 2)]
    else:
        if D.sw:
            q[Q(0, 0)] = q[Q(1, 0)]
            q[Q(0, -1)] = q[Q(2, 0)]
        if D.se:
            q[Q(npx, 0)] = q[Q(npx - 1, 0)]
            q[Q(npx, -1)] = q[Q(npx - 2, 0)]
        if D.ne:
            q[Q(npx, npy)] = q[Q(npx - 1, npy)]
            q[Q(npx, npy + 1)] = q[Q(npx - 2, npy)]
        if D.nw:
            q[Q(0, npy)] = q[Q(1, npy)]
            q[Q(0, npy + 1)] = q[Q(2, npy)]


def fill_corners_bgrid(D: Dom, q, direction: int):
    """Corner fill of a corner-staggered (B-grid) scalar (FV3 fill_corners_2d BGRID); in place."""
    o, nh, npx, npy = D.o, D.nh, D.npx, D.npy
    for j in range(1, nh + 1):
        for i in range(1, nh + 1):
            if direction == 1:
                if D.sw:
                    q[1 - i + o, 1 - j + o] = q[1 - j + o, i + 1 + o]
                if D.nw:
                    q[1 - i + o, npy + j + o] = q[1 - j + o, npy - i + o]
                if D.se:
                    q[npx + i + o, 1 - j + o] = q[npx + j + o, i + 1 + o]
                if D.ne:
                    q[npx + i + o, npy + j + o] = q[npx + j + o, npy - i + o]
            else:
                if D.sw:
                    q[1 - j + o, 1 - i + o] = q[i + 1 + o, 1 - j + o]
                if D.nw:
                    q[1 - j + o, npy + i + o] = q[i + 1 + o, npy + j + o]
                if D.se:
                    q[npx + j + o, 1 - i + o] = q[npx - i + o, 1 - j + o]
                if D.ne:
                    q[npx + j + o, npy + i + o] = q[npx - i + o, npy + j + o]


def fill_corners_dgrid_vector(D: Dom, x, y, sign=-1.0):
    """Corner fill of a D-grid staggered vector pair (FV3 fill_corners_dgrid); in place.

    ``x`` lives at (x cell, y interface), ``y`` at (x interface, y cell).
    """
    o, nh, npx, npy = D.o, D.nh, D.npx, D.npy
    x0 = x.copy()
    y0 = y.copy()
    for j in range(1, nh + 1):
        for i in range(1, nh + 1):
            if D.sw:
                x[1 - i + o, 1 - j + o] = sign * y0[1 - j + o, i + o]
            if D.nw:
                x[1 - i + o, npy + j + o] = y0[1 - j + o, npy - i + o]
            if D.se:
                x[npx - 1 + i + o, 1 - j + o] = y0[npx + j + o, i + o]
            if D.ne:
                x[npx - 1 + i + o, npy + j + o] = sign * y0[npx + j + o, npy - i + o]
    for j in range(1, nh + 1):
        for i in range(1, nh + 1):
            if D.sw:
                y[1 - j + o, 1 - i + o] = sign * x0[i + o, 1 - j + o]
            if D.nw:
                y[1 - j + o, npy - 1 + i + o] = x0[i + o, npy + j + o]
            if D.se:
                y[npx + j + o, 1 - i + o] = x0[npx - i + o, 1 - j + o]
            if D.ne:
                y[npx + j + o, npy - 1 + i + o] = sign * x0[npx - i + o, npy + j + o]


# ---------------------------------------------------------------------------------------------
# Named alternatives for the restatements DESIGN.md §2 lists as uncertain.  One environment variable, read by the oracle
# (here) AND by the library (fv3_alt in csrc/fv3_common.h): FV3_ALT="name[,name...]".  Default (unset) = the choice both were
# written with.  The point: the first run against real reference savepoints (tools/gen_golden.py + tests/test_reference_golden*.py)
# can try the alternatives in one pass -- `FV3_ALT=dz_damp_scaled pytest tests/test_reference_golden_dynamics.py` -- instead of one
# debugging session per suspect.  Known names:
#   dz_damp_scaled   update_dz_d hands del6_vt_flux (damp_vt * da_min_c)^(nord_v + 1) instead of the raw damp_vt column value
#   heat_dt_full     apply_diffusive_heating limits with |timestep * delt_max| (the whole acoustic call) instead of the sub-step dt
#   smt5_lim_fac     hord 6: the linear-scheme switch is |lim_fac * b0| < |bl - br| with the Fortran default lim_fac = 1 instead of pyFV3's 3 |b0|
#   ray_fast_plain   ray_fast without the redistribution of the damped column momentum (the older plain damping u, v, w /= 1 + rf)
#   heat_zero_first_call   heat_source is emptied on the first acoustic call of a step only (n_map == 1) instead of at the top of every call
# ---------------------------------------------------------------------------------------------
ALT_NAMES = ("dz_damp_scaled", "heat_dt_full", "smt5_lim_fac", "ray_fast_plain", "heat_zero_first_call")


def alt(name: str) -> bool:
    import os

    assert name in ALT_NAMES, name
    chosen = [x.strip() for x in os.environ.get("FV3_ALT", "").split(",") if x.strip()]
    unknown = [x for x in chosen if x not in ALT_NAMES]
    if unknown:
        raise ValueError(f"FV3_ALT: unknown alternative(s) {unknown}; known: {ALT_NAMES}")
    return name in chosen
