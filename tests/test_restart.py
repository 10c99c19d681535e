"""Restart adaptor: the reference's per-rank restart layout [REF driver/pace/driver/state.py:114-123,154-172]."""
import os

import numpy as np
import pytest


def _harness(backend, **kw):
    from pace_amd.harness import DycoreHarness

    return DycoreHarness(12, nz=6, layout=(1, 1), dt_atmos=225.0, k_split=1, n_split=1, backend=backend, **kw)


def test_restart_round_trip_continues_bitwise(hostemu, tmp_path):
    """Two steps in one go == one step, save, load into a fresh state, one more step (bit for bit), and the files carry the
    reference's names / units / (i, j, k) full-storage shapes."""
    from scipy.io import netcdf_file

    from pace_amd import restart
    from pace_amd.dyn_core import STATE_NAMES

    a = _harness("hostemu")
    a.step()
    a.step()
    b = _harness("hostemu")
    b.step()
    paths = restart.save_state(b.state, b.layout.local_ranks, str(tmp_path / "RESTART"))
    assert [os.path.basename(p) for p in paths] == [f"restart_dycore_state_{r}.nc" for r in range(6)]
    with netcdf_file(paths[3], "r", mmap=False) as f:
        assert f.variables["delp"].shape == (19, 19, 7) and f.variables["phis"].shape == (19, 19)
        assert f.variables["delp"].units == b"Pa" and f.variables["u"].dims == b"x y_interface z"
        assert set(STATE_NAMES) <= set(f.variables)
    c = _harness("hostemu", seed=1)  # a different state, fully overwritten by the restart
    restart.load_state(c.state, c.layout.local_ranks, str(tmp_path / "RESTART"))
    c.dyn._bind(c.state)  # (zs depends on phis)
    c.step()
    for n in ("delp", "pt", "u", "v", "w", "delz", "q_con"):
        for r in range(6):
            assert np.array_equal(getattr(a.state, n).numpy(r), getattr(c.state, n).numpy(r)), (n, r)


def test_restart_shape_mismatch_is_refused(hostemu, tmp_path):
    from pace_amd import restart
    from pace_amd.harness import DycoreHarness

    b = _harness("hostemu")
    restart.save_state(b.state, b.layout.local_ranks, str(tmp_path))
    other = DycoreHarness(12, nz=5, layout=(1, 1), backend="hostemu")
    with pytest.raises(ValueError, match="shape"):
        restart.load_state(other.state, other.layout.local_ranks, str(tmp_path))
    with pytest.raises(FileNotFoundError):
        restart.load_state(b.state, [7], str(tmp_path))
