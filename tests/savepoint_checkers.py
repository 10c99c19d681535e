"""Consumers of REFERENCE savepoints for the rows behind c_sw / d_sw: ``Tracer2D1L-In/Out``, ``Remapping-In/Out`` and the whole
``FVDynamics-In/Out`` step [REF tests/savepoint/thresholds/fv_dynamics.yaml:171-360], in the file format of tools/gen_golden.py
(``<savepoint>_call<N>_rank<R>.npz``, arrays [i, j, k] in the reference's storage shape).  Each checker feeds the ``-In`` arrays to
the library's operator through the same host classes the driver uses and compares with ``-Out`` under the reference's OWN
per-variable thresholds (tests/golden/reference_thresholds_fv_dynamics.json: |a - b| <= absolute + relative * |b|, the
``assert_allclose`` form of the reference's ValidationCheckpointer).  tests/test_reference_golden_dynamics.py runs them on the
reference's files when they exist (else: skip, "reference parity unpinned") and on oracle-written files of the same format.
"""
import glob
import json
import os

import numpy as np

from pace_amd.config import AcousticDynamicsConfig
from pace_amd.constants import get_constants
from pace_amd._testing import stencil_factory_for
from pace_amd.context import StencilFactory
from pace_amd.grid import make_grid
from pace_amd.halo import Layout
from pace_amd.stencils import FiniteVolumeTransport, LagrangianToEulerian, TracerAdvection
from pace_amd.topology import CubedSpherePartitioner

HERE = os.path.dirname(os.path.abspath(__file__))
THRESHOLDS = json.load(open(os.path.join(HERE, "golden", "reference_thresholds_fv_dynamics.json")))

# compute regions (offsets around 1..n: i0, i1_off, j0, j1_off) by staggering
CELL, XI, YI = (1, 0, 1, 0), (1, 1, 1, 0), (1, 0, 1, 1)
REGION = {"u": YI, "v": XI, "uc": XI, "vc": YI, "mfxd": XI, "mfyd": YI, "cxd": XI, "cyd": YI}


def ranks_present(path, savepoint, call=0):
    return sorted(int(f.rsplit("rank", 1)[1].split(".")[0]) for f in glob.glob(os.path.join(path, f"{savepoint}_call{call}_rank*.npz")))


def load(path, savepoint, rank, call=0):
    return dict(np.load(os.path.join(path, f"{savepoint}_call{call}_rank{rank}.npz")))


def meta(path):
    f = os.path.join(path, "meta.json")
    return json.load(open(f)) if os.path.exists(f) else {}


def excess(savepoint, var, got, want, floor=None):
    """max over the points of |got - want| / (absolute + relative * |want|) with the reference's thresholds of this variable
    (<= 1 passes).  ``floor``: thresholds to use for variables the reference does not list (e.g. the recorded tracers)."""
    t = THRESHOLDS.get(f"{savepoint}/{var}", floor)
    if t is None:
        raise KeyError(f"no reference threshold for {savepoint}/{var}")
    a, r = float(t["absolute"]), float(t["relative"])
    if not np.isfinite(a):
        a = 0.0
    allow = a + r * np.abs(want)
    d = np.abs(got - want)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.where(allow > 0, d / allow, np.where(d > 0, np.inf, 0.0))
    return float(q.max())


def _grids(path, ranks, nx, nz, layout=(1, 1)):
    part = CubedSpherePartitioner(nx, layout)
    grids = []
    for r in ranks:
        g = make_grid(part, r, nz=nz)
        gf = os.path.join(path, f"grid_rank{r}.npz")
        if os.path.exists(gf):  # the reference's own metric terms and hybrid coordinate
            ref = np.load(gf)
            for name in ref.files:
                if name in g.fields and ref[name].shape == g.fields[name].shape:
                    g.fields[name] = np.array(ref[name])
            for name in ("ak", "bk"):
                if name in ref.files:
                    setattr(g, name, np.array(ref[name]).ravel())
        grids.append(g)
    return part, grids


def _pad3(a, nzp):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 2:
        return a
    out = np.zeros(a.shape[:2] + (nzp,))
    k = min(a.shape[2], nzp)
    out[:, :, :k] = a[:, :, :k]
    return out


def _sl(nx, ny, nk, reg):
    i0, di, j0, dj = reg
    o = 2  # local index 1 sits at python index n_halo = 3
    return (slice(i0 + o, nx + di + o + 1), slice(j0 + o, ny + dj + o + 1), slice(0, nk))


def _cfg(path, nx, nz, layout=(1, 1), **kw):
    m = meta(path).get("config", {})
    known = {k: m[k] for k in ("hord_dp", "hord_mt", "hord_tm", "hord_vt", "hord_tr", "nord", "d4_bg", "d2_bg", "d2_bg_k1", "d2_bg_k2", "d_con", "dddmp", "vtdm4", "ke_bg",
                               "p_fac", "rf_fast", "rf_cutoff", "tau", "delt_max", "do_vort_damp", "n_sponge", "k_split", "n_split", "dt_atmos") if k in m}
    known.update(kw)
    from dataclasses import fields

    ok = {f.name for f in fields(AcousticDynamicsConfig)}
    return AcousticDynamicsConfig(npx=nx + 1, npy=nx + 1, npz=nz, layout=tuple(layout), **{k: v for k, v in known.items() if k in ok})


def check_tracer_2d_1l_savepoints(path, backend, nx=12, call=0):
    """Tracer2D1L-In -> TracerAdvection -> Tracer2D1L-Out.  The reference checkpoints dp1, mfxd, mfyd, cxd, cyd (the operator scales
    the accumulated fluxes / Courant numbers by 1 / n_split in place); ``tracer_*`` arrays, where the generator recorded them from
    the state, are advected and compared too (threshold: the reference's for qvapor at FVDynamics-Out)."""
    ranks = ranks_present(path, "Tracer2D1L-In", call)
    inp = [load(path, "Tracer2D1L-In", r, call) for r in ranks]
    out = [load(path, "Tracer2D1L-Out", r, call) for r in ranks]
    nz = inp[0]["dp1"].shape[2] - 1  # (the reference's storages are padded to the interface shape)
    part, grids = _grids(path, ranks, nx, nz)
    cfg = _cfg(path, nx, nz)
    sf = stencil_factory_for(backend)(grids, cfg, get_constants())
    qf = sf.quantity_factory
    nzp = nz + 1
    Q = {n: qf.from_array([_pad3(x[n], nzp) for x in inp], ("x", "y", "z")) for n in ("dp1", "mfxd", "mfyd", "cxd", "cyd")}
    tnames = sorted(k for k in inp[0] if k.startswith("tracer_"))
    T = {n: qf.from_array([_pad3(x[n], nzp) for x in inp], ("x", "y", "z")) for n in tnames}
    if not T:  # nothing recorded: any positive field rides along (the compared variables do not depend on it)
        T = {"tracer_dummy": qf.from_array([np.full(x["dp1"].shape[:2] + (nzp,), 1.0e-3) for x in inp], ("x", "y", "z"))}
    lay = Layout(part, 1, 0) if len(ranks) == part.total_ranks else None
    op = TracerAdvection(sf, qf, FiniteVolumeTransport(sf, qf, grids, hord=int(meta(path).get("config", {}).get("hord_tr", 8))), grids, lay, T)
    if lay is None:
        op._halo = False  # (a subset of the ranks: no exchange is possible; only n_split = 1 can be checked)
    op(T, Q["dp1"], Q["mfxd"], Q["mfyd"], Q["cxd"], Q["cyd"])
    if lay is None and op.n_split > 1:
        raise RuntimeError(f"the savepoints need {op.n_split} sub-cycles: all six ranks' files are required for the tracer halo updates")
    errs = {}
    for i, r in enumerate(ranks):
        for var in ("dp1", "mfxd", "mfyd", "cxd", "cyd"):
            sl = _sl(part.nx, part.ny, nz, REGION.get(var, CELL))
            errs[var] = max(errs.get(var, 0.0), excess("Tracer2D1L-Out", var, Q[var].numpy(i)[sl], _pad3(out[i][var], nzp)[sl]))
        for var in tnames:
            sl = _sl(part.nx, part.ny, nz, CELL)
            errs[var] = max(errs.get(var, 0.0), excess("Tracer2D1L-Out", var, T[var].numpy(i)[sl], _pad3(out[i][var], nzp)[sl], floor=THRESHOLDS["FVDynamics-Out/qvapor"]))
    return errs, op.n_split


REMAP_OUT = ("delp", "delz", "pe", "peln", "pk", "pkz", "pt", "u", "v", "w")


def check_remapping_savepoints(path, backend, nx=12, call=0):
    """Remapping-In -> LagrangianToEulerian -> Remapping-Out under the reference's thresholds.  What this build remaps is the DRY
    configuration (no moist_cv / saturation adjustment / energy fixer: DESIGN §8), so reference files of the moist case are
    expected to differ in pt / pkz / cappa -- the checker reports every variable, the caller decides."""
    ranks = ranks_present(path, "Remapping-In", call)
    inp = [load(path, "Remapping-In", r, call) for r in ranks]
    out = [load(path, "Remapping-Out", r, call) for r in ranks]
    nz = inp[0]["delp"].shape[2] - 1
    part, grids = _grids(path, ranks, nx, nz)
    cfg = _cfg(path, nx, nz)
    sf = stencil_factory_for(backend)(grids, cfg, get_constants())
    qf = sf.quantity_factory
    nzp = nz + 1
    names = ("pt", "delp", "delz", "peln", "pe", "pk", "pkz", "u", "v", "w", "cappa")
    Q = {n: qf.from_array([_pad3(x[n], nzp) for x in inp], ("x", "y", "z")) for n in names}
    tnames = sorted(k for k in inp[0] if k.startswith("tracer_"))
    T = {n: qf.from_array([_pad3(x[n], nzp) for x in inp], ("x", "y", "z")) for n in tnames}
    ps = qf.zeros(("x", "y"))
    W = qf.from_array([np.asarray(x["wsd"], dtype=np.float64).reshape(x["delp"].shape[:2]) for x in inp], ("x", "y"))
    LagrangianToEulerian(sf, qf, grids)(T, *[Q[n] for n in names], ps, W)
    errs = {}
    for i, r in enumerate(ranks):
        for var in REMAP_OUT:
            if var not in out[i]:
                continue
            nk = nz + 1 if var in ("pe", "peln", "pk") else nz
            sl = _sl(part.nx, part.ny, nk, REGION.get(var, CELL))
            errs[var] = max(errs.get(var, 0.0), excess("Remapping-Out", var, Q[var].numpy(i)[sl], _pad3(out[i][var], nzp)[sl]))
        for var in tnames:
            sl = _sl(part.nx, part.ny, nz, CELL)
            errs[var] = max(errs.get(var, 0.0), excess("Remapping-Out", var, T[var].numpy(i)[sl], _pad3(out[i][var], nzp)[sl], floor=THRESHOLDS["FVDynamics-Out/qvapor"]))
    return errs


FVDYN_OUT = ("u", "v", "w", "delz", "ua", "va", "uc", "vc")


def check_fv_dynamics_savepoints(path, backend, nx=12):
    """FVDynamics-In -> k_split x [AcousticDynamics, TracerAdvection, LagrangianToEulerian] with this build's halo exchange ->
    FVDynamics-Out, all six ranks (the step needs its neighbours).  The generator stores the WHOLE prognostic state beside the
    nine variables the reference checkpoints (``state_*`` arrays), the tracers as ``tracer_*``."""
    from pace_amd.dyn_core import STATE_NAMES
    from pace_amd._testing import harness_for
    from pace_amd.harness import DycoreHarness

    ranks = ranks_present(path, "FVDynamics-In")
    if ranks != list(range(6)):
        raise RuntimeError(f"FVDynamics needs the files of all six ranks (found {ranks}): run tools/gen_golden.py under mpirun -n 6")
    inp = [load(path, "FVDynamics-In", r) for r in ranks]
    out = [load(path, "FVDynamics-Out", r) for r in ranks]
    nz = inp[0]["u"].shape[2] - 1
    m = meta(path).get("config", {})
    tnames = sorted(k for k in inp[0] if k.startswith("tracer_"))
    over = {k: m[k] for k in ("hord_dp", "hord_mt", "hord_tm", "hord_vt", "nord", "d4_bg", "d2_bg", "d2_bg_k1", "d2_bg_k2", "d_con", "dddmp", "vtdm4", "ke_bg", "p_fac", "rf_fast",
                              "rf_cutoff", "tau", "delt_max", "do_vort_damp", "n_sponge") if k in m}
    h = harness_for(backend)(nx, nz=nz, layout=(1, 1), dt_atmos=float(m.get("dt_atmos", 225.0)), k_split=int(m.get("k_split", 1)), n_split=int(m.get("n_split", 1)),
                      config_overrides=over, n_tracers=len(tnames), hord_tr=int(m.get("hord_tr", 8)), remap=True)
    nzp = nz + 1
    for i in ranks:
        for n in STATE_NAMES + ["phis"]:
            src = inp[i].get(n, inp[i].get("state_" + n))
            if src is None:
                raise KeyError(f"FVDynamics-In rank {i}: the state variable {n} is neither checkpointed nor recorded as state_{n}")
            getattr(h.state, n).set_numpy(_pad3(src, nzp), i)
        for t, n in enumerate(tnames):
            h.tracers[f"tracer{t}"].set_numpy(_pad3(inp[i][n], nzp), i)
    h.step()
    h.synchronize()
    errs = {}
    for i in ranks:
        for var in FVDYN_OUT:
            sl = _sl(h.part.nx, h.part.ny, nz, REGION.get(var, CELL))
            errs[var] = max(errs.get(var, 0.0), excess("FVDynamics-Out", var, getattr(h.state, var).numpy(i)[sl], _pad3(out[i][var], nzp)[sl]))
        for t, n in enumerate(tnames):
            sl = _sl(h.part.nx, h.part.ny, nz, CELL)
            errs[n] = max(errs.get(n, 0.0), excess("FVDynamics-Out", "qvapor", h.tracers[f"tracer{t}"].numpy(i)[sl], _pad3(out[i][n], nzp)[sl]))
    return errs
