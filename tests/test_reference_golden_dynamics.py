"""Reference savepoints beyond c_sw / d_sw: Tracer2D1L, Remapping and the whole FVDynamics step
[REF tests/savepoint/thresholds/fv_dynamics.yaml:171-360; tests/savepoint/test_checkpoints.py:141-158].

`tools/gen_golden.py` records them where pyFV3 imports (under ``mpirun -n 6`` for the multi-rank ones); none exists in this tree's
containers, so the `*_against_reference_savepoints` tests skip with "reference parity unpinned".  The `*_checker_on_oracle_files`
twins run the same consumers (tests/savepoint_checkers.py), with the reference's own per-variable thresholds, on files the numpy /
C oracle writes in the generator's format -- the consumers are exercised, the oracle <-> reference side stays open.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle"))
import savepoint_checkers as sc  # noqa: E402
from helpers import oracle_cube  # noqa: E402
from pace_amd.constants import get_constants  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "golden_c12")
SKIP = "reference parity unpinned: tests/golden/golden_c12/{}-In_call0_rank*.npz is absent (generate it with tools/gen_golden.py where pyFV3 imports)"


def _write_oracle_savepoints(tmp_path, nz=6, k_split=1, n_split=2, n_tracers=1, courant_boost=None):
    """One oracle step_dynamics of a C12 cube, every savepoint written like tools/gen_golden.py writes the reference's."""
    import json

    from fv3_oracle.step_dynamics import step_dynamics
    from pace_amd.dyn_core import STATE_NAMES

    part, cfg, grids, ost, phis, odyn = oracle_cube(12, (1, 1), nz, dict(n_split=n_split, k_split=k_split))
    for s in ost:
        s["pkz"][...] = 1.0
    tr = [{f"q{t}": np.ascontiguousarray(s["q_con"] * 1e3 * (t + 1) + 1e-3) for t in range(n_tracers)} for s in ost]
    calls = {}

    def rec(sp, r, d):
        n = calls.get((sp, r), 0)
        calls[(sp, r)] = n + 1
        d = {k: np.array(v, dtype=np.float64) for k, v in d.items()}
        if sp == "FVDynamics-In":  # the whole state rides along (the reference checkpoints nine variables of it)
            for k in STATE_NAMES:
                if k not in d:
                    d["state_" + k] = np.array(ost[r][k])
            d["state_phis"] = np.array(phis[r])
        np.savez(tmp_path / f"{sp}_call{n}_rank{r}.npz", **d)

    step_dynamics(odyn, get_constants(), ost, tr, 225.0, k_split, 8, record=rec)
    json.dump({"config": dict(k_split=k_split, n_split=n_split, dt_atmos=225.0, hord_tr=8)}, open(tmp_path / "meta.json", "w"))
    return part


def test_tracer_2d_1l_against_reference_savepoints(hostemu):
    if not sc.ranks_present(GOLDEN, "Tracer2D1L-In"):
        pytest.skip(SKIP.format("Tracer2D1L"))
    errs, _ = sc.check_tracer_2d_1l_savepoints(GOLDEN, "hostemu")
    bad = {k: v for k, v in errs.items() if v > 1.0 and k not in ("mfxd", "mfyd")}  # (mfxd / mfyd: the reference's own threshold is "anything", relative 100)
    assert not bad, f"tracer_2d_1l leaves the reference's thresholds (excess factors): {bad} (all: {errs})"


def test_tracer_2d_1l_checker_on_oracle_files(hostemu, tmp_path):
    _write_oracle_savepoints(tmp_path, n_tracers=2)
    errs, ns = sc.check_tracer_2d_1l_savepoints(str(tmp_path), "hostemu")
    assert set(errs) == {"dp1", "mfxd", "mfyd", "cxd", "cyd", "tracer_q0", "tracer_q1"} and ns >= 1
    assert max(errs.values()) <= 1.0, errs  # inside the reference's thresholds of every variable


def test_remapping_against_reference_savepoints(hostemu):
    if not sc.ranks_present(GOLDEN, "Remapping-In"):
        pytest.skip(SKIP.format("Remapping"))
    errs = sc.check_remapping_savepoints(GOLDEN, "hostemu")
    bad = {k: v for k, v in errs.items() if v > 1.0}
    assert not bad, f"the remap leaves the reference's thresholds (excess factors): {bad} (all: {errs}); moist reference data differ in pt / pkz by construction (DESIGN §8)"


def test_remapping_checker_on_oracle_files(hostemu, tmp_path):
    _write_oracle_savepoints(tmp_path, nz=12, n_tracers=1)
    errs = sc.check_remapping_savepoints(str(tmp_path), "hostemu")
    assert set(sc.REMAP_OUT) <= set(errs) and "tracer_q0" in errs
    assert max(errs.values()) <= 1.0, errs  # inside the reference's thresholds (~10 ulp of each field) on every variable


def test_fv_dynamics_against_reference_savepoints(hostemu):
    if sc.ranks_present(GOLDEN, "FVDynamics-In") != list(range(6)):
        pytest.skip(SKIP.format("FVDynamics") + " for all six ranks (mpirun -n 6)")
    errs = sc.check_fv_dynamics_savepoints(GOLDEN, "hostemu")
    bad = {k: v for k, v in errs.items() if v > 1.0 and k not in ("ua", "va")}  # (ua / va: the reference's threshold is 1280 absolute: fill values)
    assert not bad, f"step_dynamics leaves the reference's thresholds (excess factors): {bad} (all: {errs})"


def test_fv_dynamics_checker_on_oracle_files(hostemu, tmp_path):
    """The whole step (2 x [acoustic call of 2 sub-steps, tracer advection, remap], six ranks, this build's halo exchange) against
    the oracle's, through the FVDynamics consumer."""
    _write_oracle_savepoints(tmp_path, nz=8, k_split=2, n_split=2, n_tracers=1)
    errs = sc.check_fv_dynamics_savepoints(str(tmp_path), "hostemu")
    assert set(sc.FVDYN_OUT) <= set(errs) and "tracer_q0" in errs
    # thresholds of the reference (u 2e-11 m/s, w 1.5e-12 m/s, delz 1.3e-10 m, uc 6e-13): after two acoustic calls + remaps the
    # library is inside every one of them (observed: w at 0.06 of its threshold -- the log / exp of the Riemann solvers --, the rest 0)
    assert max(errs.values()) <= 1.0, errs
