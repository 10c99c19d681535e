"""Restart adaptor of the dycore state: one netCDF file per rank in the layout the reference driver
writes and reads back,

    <restart_path>/restart_dycore_state_<rank>.nc

[REF driver/pace/driver/state.py:114-123 ``DriverState.save_state``; :154-172 ``_overwrite_state_from_restart``:
``state.<field>.data[:] = ds[<field>].data[:]`` for every field that carries units].  Each variable is the Quantity's full
``data`` array -- halo included, index order (i, j, k) -- with its ``units`` attribute and dimension names from the
Quantity's dims; the reference goes through xarray / netCDF4, this build writes classic netCDF-3 with scipy (64-bit
offsets), which xarray opens unchanged.  The files are keyed by the reference's GLOBAL rank number, so a run on N GPUs
(several sub-domains per process) and the reference's one-rank-per-process run exchange restarts freely.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional, Sequence

import numpy as np

from .dyn_core import STATE_NAMES, DycoreState

PREFIX = "restart_dycore_state"
_UNITS = {"u": "m/s", "v": "m/s", "w": "m/s", "ua": "m/s", "va": "m/s", "uc": "m/s", "vc": "m/s", "delp": "Pa", "delz": "m", "pt": "K", "pe": "Pa", "pk": "Pa^kappa",
          "peln": "ln(Pa)", "pkz": "Pa^kappa", "q_con": "kg/kg", "omga": "Pa/s", "cappa": "", "mfxd": "unknown", "mfyd": "unknown", "cxd": "", "cyd": "", "diss_estd": "unknown",
          "phis": "m^2 s^-2"}


def _path(restart_path: str, rank: int) -> str:
    return os.path.join(restart_path, f"{PREFIX}_{rank}.nc")


def save_state(state: DycoreState, local_ranks: Sequence[int], restart_path: str = "RESTART", names: Optional[Iterable[str]] = None) -> list:
    """Write one file per local rank; returns the paths.  ``local_ranks[i]`` is the global rank of sub-domain i."""
    from scipy.io import netcdf_file

    os.makedirs(restart_path, exist_ok=True)
    names = list(names or (STATE_NAMES + ["phis"]))
    out = []
    for i, rank in enumerate(local_ranks):
        p = _path(restart_path, rank)
        with netcdf_file(p, "w", version=2) as f:
            f.history = "pace_amd.restart.save_state"
            f.rank = np.int32(rank)
            for n in names:
                q = getattr(state, n)
                a = q.numpy(i)  # (i, j[, k]) host copy of the full storage
                dims = []
                for d, length in zip(q.dims, a.shape):
                    dn = f"{d}_{length}"  # (a netCDF dimension has ONE length: cell / interface variants get their own)
                    if dn not in f.dimensions:
                        f.createDimension(dn, length)
                    dims.append(dn)
                v = f.createVariable(n, a.dtype, tuple(dims))
                v[:] = a
                v.units = q.units or _UNITS.get(n, "")
                v.dims = " ".join(q.dims)
        out.append(p)
    return out


def load_state(state: DycoreState, local_ranks: Sequence[int], restart_path: str = "RESTART", names: Optional[Iterable[str]] = None) -> DycoreState:
    """Overwrite ``state`` from the per-rank files (every field present in the file and in the state)."""
    from scipy.io import netcdf_file

    want = list(names or (STATE_NAMES + ["phis"]))
    for i, rank in enumerate(local_ranks):
        p = _path(restart_path, rank)
        if not os.path.exists(p):
            raise FileNotFoundError(f"{p}: no restart file for rank {rank}")
        with netcdf_file(p, "r", mmap=False) as f:
            for n in want:
                if n not in f.variables:
                    continue
                a = np.array(f.variables[n][:])
                a = a.astype(a.dtype.newbyteorder("="))  # (netCDF classic is big-endian)
                q = getattr(state, n)
                exp = q.numpy(i).shape
                if a.shape != exp:
                    raise ValueError(f"{p}: variable {n} has shape {a.shape}, the state expects {exp} (different nx / nz / halo)")
                q.set_numpy(a, i)
    return state
