"""Restart adaptor of the dycore state: one netCDF file per rank, named and laid out like the files the reference
driver writes,

    <restart_path>/restart_dycore_state_<rank>.nc

[REF driver/pace/driver/state.py:114-123 ``DriverState.save_state``; :154-172 ``_overwrite_state_from_restart``:
``state.<field>.data[:] = ds[<field>].data[:]`` for every field that carries units].  Each variable is the Quantity's full
``data`` array -- halo included, index order (i, j, k) -- with its ``units`` attribute and dimension names from the
Quantity's dims, keyed by the reference's GLOBAL rank number (a run on N GPUs with several sub-domains per process writes the
same files as a one-rank-per-process run).

What does and does not interchange with the reference (stated, not assumed):

* WRITING: classic netCDF-3, 64-bit offsets (scipy is the only netCDF writer in this image); xarray opens such files unchanged.
  The reference's reader indexes EVERY ``DycoreState`` field that has units (the tracers ``qvapor`` ... ``qcld`` included); this
  build's state holds the acoustic path's fields only, so a reference run can restart from these files only if the caller adds
  the remaining fields through ``extra`` (name -> Quantity: the harness passes its tracers).
* READING: the reference writes through ``xr_dataset.to_netcdf()``, i.e. netCDF-4 / HDF5 by default.  Such files need ``netCDF4``,
  ``h5netcdf`` or ``xarray`` in the environment (tried in that order); with none of them -- this image -- ``load_state`` says so
  instead of failing inside scipy.  Classic files (this build's own, or a reference file written with ``format="NETCDF3_64BIT"``)
  are read with scipy.
* ``load_state`` is strict: a requested variable that is missing from a file is an error (``allow_missing=True`` keeps what the
  state holds for it), so a restart can never silently continue from synthetic data.
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


def save_state(state: DycoreState, local_ranks: Sequence[int], restart_path: str = "RESTART", names: Optional[Iterable[str]] = None, extra=None) -> list:
    """Write one file per local rank; returns the paths.  ``local_ranks[i]`` is the global rank of sub-domain i.
    ``extra``: {name: Quantity} written beside the state's fields (the tracers)."""
    from scipy.io import netcdf_file

    os.makedirs(restart_path, exist_ok=True)
    names = list(names or (STATE_NAMES + ["phis"]))
    fields = [(n, getattr(state, n)) for n in names] + list((extra or {}).items())
    out = []
    for i, rank in enumerate(local_ranks):
        p = _path(restart_path, rank)
        with netcdf_file(p, "w", version=2) as f:
            f.history = "pace_amd.restart.save_state"
            f.rank = np.int32(rank)
            for n, q in fields:
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


class MissingValuesError(RuntimeError):
    """a restart variable holds missing values (masked entries, or entries equal to its _FillValue / missing_value attribute)"""


def _unmasked(a, path, name, fill=()):
    """A restart variable as a plain array.  The netCDF4 reader hands back masked arrays where a value equals the variable's
    _FillValue; h5netcdf and xarray (opened without mask_and_scale) hand back the raw numbers, so the variable's ``_FillValue`` /
    ``missing_value`` attributes come in through ``fill`` and are compared explicitly.  A restart with missing values is not a
    state -- refuse it instead of loading the fill value (9.97e36: finite) as data."""
    if np.ma.isMaskedArray(a):
        if np.ma.is_masked(a):
            raise MissingValuesError(f"{path}: variable {name} has {int(np.ma.count_masked(a))} masked (missing / _FillValue) entries")
        a = a.filled()  # (nothing masked: filled() only drops the mask)
    a = np.asarray(a)
    if a.dtype.kind not in "fiu":  # (character / string variables -- time stamps, names -- carry no numeric fill value to compare)
        return a
    is_float = np.issubdtype(a.dtype, np.floating)
    for f in fill:
        if f is None:
            continue
        try:
            raw = np.asarray(f).ravel()
            if raw.dtype.kind not in "fiu":
                continue
            fv = raw.astype(a.dtype)
        except (TypeError, ValueError, OverflowError):
            continue
        for v0, v in zip(raw, fv):
            if not is_float and (not np.isfinite(v0) or int(v0) != int(v)):  # (a fill value the variable's own integer type cannot hold cannot occur in it)
                continue
            n = int(np.count_nonzero(np.isnan(a))) if (is_float and np.isnan(v)) else int(np.count_nonzero(a == v))
            if n:
                raise MissingValuesError(f"{path}: variable {name} has {n} masked (missing / _FillValue = {v!r}) entries")
    return a


def _fill_attrs(attrs):
    return tuple(attrs.get(k) for k in ("_FillValue", "missing_value") if k in attrs)


def _open_variables(path: str):
    """{name: ndarray (native byte order)} of one restart file, with whatever reader fits its format."""
    with open(path, "rb") as fh:
        magic = fh.read(4)
    if magic[:3] == b"CDF":  # classic netCDF (this build's files)
        from scipy.io import netcdf_file

        with netcdf_file(path, "r", mmap=False) as f:
            return {n: np.array(v[:]).astype(v[:].dtype.newbyteorder("=")) for n, v in f.variables.items()}
    if magic == b"\x89HDF":  # netCDF-4 / HDF5: what the reference's xarray writes by default
        try:
            import netCDF4

            reader = "netCDF4"
        except ImportError:
            reader = None
        if reader:
            try:
                with netCDF4.Dataset(path) as ds:
                    return {n: _unmasked(ds.variables[n][:], path, n) for n in ds.variables}
            except MissingValuesError:
                raise  # (not a reader failure: reported as what it is)
            except (OSError, RuntimeError) as e:
                raise RuntimeError(f"{path}: netCDF4 could not read the file ({e})") from e
        try:
            import h5netcdf

            reader = "h5netcdf"
        except ImportError:
            pass
        if reader:
            try:
                with h5netcdf.File(path, "r") as ds:
                    return {n: _unmasked(ds.variables[n][...], path, n, _fill_attrs(ds.variables[n].attrs)) for n in ds.variables}
            except MissingValuesError:
                raise
            except (OSError, RuntimeError, KeyError) as e:
                raise RuntimeError(f"{path}: h5netcdf could not read the file ({e})") from e
        try:
            import xarray as xr

            reader = "xarray"
        except ImportError:
            pass
        if reader:
            try:
                with xr.open_dataset(path, mask_and_scale=False) as ds:
                    return {n: _unmasked(ds[n].data, path, n, _fill_attrs(ds[n].attrs)) for n in ds.variables}
            except MissingValuesError:
                raise
            except (OSError, RuntimeError, ValueError) as e:
                raise RuntimeError(f"{path}: xarray could not read the file ({e})") from e
        raise RuntimeError(f"{path} is a netCDF-4 / HDF5 file (the reference's default restart format) and this environment has none of netCDF4, h5netcdf, "
                           "xarray to read it; convert it to classic netCDF (xarray: to_netcdf(format='NETCDF3_64BIT')) or install one of them")
    raise RuntimeError(f"{path}: not a netCDF file (magic {magic!r})")


def load_state(state: DycoreState, local_ranks: Sequence[int], restart_path: str = "RESTART", names: Optional[Iterable[str]] = None, extra=None,
               allow_missing: bool = False) -> DycoreState:
    """Overwrite ``state`` (and the ``extra`` Quantities, e.g. the tracers) from the per-rank files.  Every requested variable must
    be in every file unless ``allow_missing``."""
    want = [(n, getattr(state, n)) for n in (names or (STATE_NAMES + ["phis"]))] + list((extra or {}).items())
    for i, rank in enumerate(local_ranks):
        p = _path(restart_path, rank)
        if not os.path.exists(p):
            raise FileNotFoundError(f"{p}: no restart file for rank {rank}")
        have = _open_variables(p)
        missing = [n for n, _ in want if n not in have]
        if missing and not allow_missing:
            raise KeyError(f"{p}: variables missing from the restart file: {', '.join(missing)} (allow_missing=True keeps the state's own values for them)")
        for n, q in want:
            if n not in have:
                continue
            a = have[n]
            exp = q.numpy(i).shape
            if a.shape != exp:
                raise ValueError(f"{p}: variable {n} has shape {a.shape}, the state expects {exp} (different nx / nz / halo)")
            q.set_numpy(a, i)
    return state
