"""Index algebra of the cube-corner halo fills (``copy_corners``, ``fill_corners``).

At a cube corner only three tiles meet, so the 3x3 corner block of a rank's
halo has no owner; the reference's stencils fill it, per sweep direction, with
the rotated edge-halo block so that 1-D operators can run straight through
[SURVEY A.13].  The source indices below are written in *local* Fortran
numbering (first compute cell = 1, ``npx = nx + 1``) and converted to storage
indices; they are shared by the grid generator, the numpy oracle's tests and
the device index tables so that one definition is exercised everywhere.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

CORNERS = ("sw", "se", "ne", "nw")


def corner_flags(west: bool, east: bool, south: bool, north: bool) -> Dict[str, bool]:
    return {"sw": west and south, "se": east and south, "ne": east and north, "nw": west and north}


def copy_corners_index(nx: int, ny: int, nh: int, direction: int, corner: str) -> Tuple[np.ndarray, ...]:
    """(dst_i, dst_j, src_i, src_j) storage indices for cell-centred ``copy_corners``.

    direction 1 = x sweep, 2 = y sweep.
    """
    npx, npy = nx + 1, ny + 1
    o = nh - 1
    lo = np.arange(1 - nh, 1)
    hix = np.arange(npx, npx + nh)
    hiy = np.arange(npy, npy + nh)
    ii = lo if corner in ("sw", "nw") else hix
    jj = lo if corner in ("sw", "se") else hiy
    I, J = np.meshgrid(ii, jj, indexing="ij")
    if direction == 1:
        if corner == "sw":
            SI, SJ = J, 1 - I
        elif corner == "se":
            SI, SJ = npx - J, I - npx + 1
        elif corner == "ne":
            SI, SJ = npx + (J - npy), npy - 1 - (I - npx)
        else:
            SI, SJ = npy - J, npy - 1 + I
    else:
        if corner == "sw":
            SI, SJ = 1 - J, I
        elif corner == "se":
            SI, SJ = npx - 1 + J, npx - I
        elif corner == "ne":
            SI, SJ = npx - 1 - (J - npy), npy + (I - npx)
        else:
            SI, SJ = J - npy + 1, npy - I
    return (I + o).ravel(), (J + o).ravel(), (SI + o).ravel(), (SJ + o).ravel()
