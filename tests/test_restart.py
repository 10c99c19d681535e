"""Restart adaptor: the reference's per-rank restart layout [REF driver/pace/driver/state.py:114-123,154-172]."""
import os

import numpy as np
import pytest


def _harness(backend, **kw):
    from pace_amd._testing import harness_for

    return harness_for(backend)(12, nz=6, layout=(1, 1), dt_atmos=225.0, k_split=1, n_split=1, **kw)


def test_restart_round_trip_continues_bitwise(backend, tmp_path):
    """Two steps in one go == one step, save, load into a fresh state, one more step (bit for bit), and the files carry the
    reference's names / units / (i, j, k) full-storage shapes.  On the host emulation and -- device state -> files -> fresh device
    state -> continuation -- on the MI355X (-m gpu)."""
    from scipy.io import netcdf_file

    from pace_amd import restart
    from pace_amd.dyn_core import STATE_NAMES

    a = _harness(backend)
    a.step()
    a.step()
    b = _harness(backend)
    b.step()
    paths = restart.save_state(b.state, b.layout.local_ranks, str(tmp_path / "RESTART"))
    assert [os.path.basename(p) for p in paths] == [f"restart_dycore_state_{r}.nc" for r in range(6)]
    with netcdf_file(paths[3], "r", mmap=False) as f:
        assert f.variables["delp"].shape == (19, 19, 7) and f.variables["phis"].shape == (19, 19)
        assert f.variables["delp"].units == b"Pa" and f.variables["u"].dims == b"x y_interface z"
        assert set(STATE_NAMES) <= set(f.variables)
    c = _harness(backend, seed=1)  # a different state, fully overwritten by the restart
    restart.load_state(c.state, c.layout.local_ranks, str(tmp_path / "RESTART"))
    c.dyn._bind(c.state)  # (zs depends on phis)
    c.step()
    for n in ("delp", "pt", "u", "v", "w", "delz", "q_con"):
        for r in range(6):
            assert np.array_equal(getattr(a.state, n).numpy(r), getattr(c.state, n).numpy(r)), (n, r)


def test_restart_with_tracers_and_remap_continues_bitwise(backend, tmp_path):
    """The body of step_dynamics (acoustic call + tracer advection + remap) across a restart: the tracers travel in the same
    files (``extra``), and the continued run equals the uninterrupted one bit for bit."""
    from pace_amd import restart

    kw = dict(n_tracers=2, hord_tr=8, remap=True)
    a = _harness(backend, **kw)
    a.step()
    a.step()
    b = _harness(backend, **kw)
    b.step()
    restart.save_state(b.state, b.layout.local_ranks, str(tmp_path), extra=b.tracers)
    c = _harness(backend, seed=1, **kw)
    restart.load_state(c.state, c.layout.local_ranks, str(tmp_path), extra=c.tracers)
    c.dyn._bind(c.state)
    c.step()
    a.synchronize()
    c.synchronize()
    for r in range(6):
        for n in ("delp", "pt", "u", "v", "w", "delz"):
            assert np.array_equal(getattr(a.state, n).numpy(r), getattr(c.state, n).numpy(r)), (n, r)
        for n in a.tracers:
            assert np.array_equal(a.tracers[n].numpy(r), c.tracers[n].numpy(r)), (n, r)


def test_restart_is_strict_about_missing_variables_and_foreign_formats(hostemu, tmp_path):
    from pace_amd import restart

    b = _harness("hostemu", n_tracers=1)
    restart.save_state(b.state, b.layout.local_ranks, str(tmp_path))  # (without the tracers)
    with pytest.raises(KeyError, match="tracer0"):
        restart.load_state(b.state, b.layout.local_ranks, str(tmp_path), extra=b.tracers)
    restart.load_state(b.state, b.layout.local_ranks, str(tmp_path), extra=b.tracers, allow_missing=True)
    # a netCDF-4 / HDF5 file (the reference's default) is recognised; without an HDF5 reader the error says what to do
    p = tmp_path / "h5"
    p.mkdir()
    for r in range(6):
        (p / f"restart_dycore_state_{r}.nc").write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    import importlib.util

    have_reader = any(importlib.util.find_spec(m) is not None for m in ("netCDF4", "h5netcdf", "xarray"))
    # no reader: the message names the format and what to install; a reader that chokes on the truncated file: its error is wrapped
    with pytest.raises(RuntimeError, match="could not read the file" if have_reader else "netCDF-4 / HDF5"):
        restart.load_state(b.state, b.layout.local_ranks, str(p))


def test_restart_refuses_masked_values():
    """A _FillValue in a restart variable comes back masked from the netCDF readers: not a state."""
    import numpy as np

    from pace_amd import restart

    a = np.ma.masked_array(np.arange(6.0).reshape(2, 3), mask=False)
    assert isinstance(restart._unmasked(a, "f.nc", "u"), np.ndarray) and not np.ma.isMaskedArray(restart._unmasked(a, "f.nc", "u"))
    a.mask[1, 2] = True
    with pytest.raises(RuntimeError, match="masked"):
        restart._unmasked(a, "f.nc", "u")
    # readers that do not mask (h5netcdf, xarray without mask_and_scale) hand over the raw numbers + the variable's attributes
    raw = np.arange(6.0).reshape(2, 3)
    raw[1, 1] = 9.969209968386869e36
    with pytest.raises(restart.MissingValuesError, match="masked"):
        restart._unmasked(raw, "f.nc", "u", restart._fill_attrs({"_FillValue": np.float64(9.969209968386869e36)}))
    assert restart._unmasked(raw, "f.nc", "u", restart._fill_attrs({"units": "m"})) is not None
    with pytest.raises(restart.MissingValuesError):
        restart._unmasked(np.array([1.0, np.nan]), "f.nc", "u", (np.nan,))
    # a character variable with a _FillValue (time stamps, names) is passed through, not compared; a fill value its integer type cannot hold cannot occur in it
    chars = np.array([b"a", b"b"], dtype="S1")
    assert restart._unmasked(chars, "f.nc", "stamp", (b" ",)) is chars
    assert restart._unmasked(np.array([1, 2], dtype=np.int8), "f.nc", "flag", (np.int64(-32767),)) is not None
    with pytest.raises(restart.MissingValuesError):
        restart._unmasked(np.array([1, -127], dtype=np.int8), "f.nc", "flag", (np.int8(-127),))


def test_restart_shape_mismatch_is_refused(hostemu, tmp_path):
    from pace_amd import restart
    from pace_amd._testing import hostemu_harness

    b = _harness("hostemu")
    restart.save_state(b.state, b.layout.local_ranks, str(tmp_path))
    other = hostemu_harness(12, nz=5, layout=(1, 1))
    with pytest.raises(ValueError, match="shape"):
        restart.load_state(other.state, other.layout.local_ranks, str(tmp_path))
    with pytest.raises(FileNotFoundError):
        restart.load_state(b.state, [7], str(tmp_path))
