"""Parity against savepoints of the REFERENCE (pyFV3 numpy backend), when they exist.

`tools/gen_golden.py` writes them (C_SW-In/Out, D_SW-In/Out of the first acoustic sub-step) inside an environment where
pyFV3 imports; none exists in this tree's containers, so the `*_against_reference_savepoints` tests skip with "reference
parity unpinned" (SURVEY §8c).  The `test_*_checker_on_oracle_generated_files` tests run the very same checkers on savepoint
files written from the numpy oracle, so the checkers themselves are exercised.
"""
import glob
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
from helpers import assert_close  # noqa: E402
from pace_amd.config import AcousticDynamicsConfig  # noqa: E402
from pace_amd.constants import get_constants  # noqa: E402
from pace_amd._testing import stencil_factory_for
from pace_amd.context import StencilFactory  # noqa: E402
from pace_amd.grid import make_grid  # noqa: E402
from pace_amd.topology import CubedSpherePartitioner  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "golden_c12")

# savepoint variable -> (our name, staggered region to compare: (i0, i1_off, j0, j1_off) around 1..n)
CSW_OUT = {"ucd": ("uc", (1, 1, 1, 0)), "vcd": ("vc", (1, 0, 1, 1)), "uad": ("ua", (0, 1, 0, 1)), "vad": ("va", (0, 1, 0, 1)),
           "divgdd": ("divgd", (1, 1, 1, 1))}


def _pad(a, nzp):
    out = np.zeros(a.shape[:2] + (nzp,))
    out[:, :, : min(a.shape[2], nzp)] = a[:, :, :nzp]
    return out


def check_c_sw_savepoints(path, backend, rank=0, nx=12, grid_file=True):
    """Feed C_SW-In of call 0 to fv3_c_sw and compare with C_SW-Out; returns {var: max field-relative error}."""
    inp = dict(np.load(os.path.join(path, f"C_SW-In_call0_rank{rank}.npz")))
    out = dict(np.load(os.path.join(path, f"C_SW-Out_call0_rank{rank}.npz")))
    nz = inp["delpd"].shape[2] - 1
    part = CubedSpherePartitioner(nx, (1, 1))
    g = make_grid(part, rank, nz=nz)
    grid_diffs = {}
    gf = os.path.join(path, f"grid_rank{rank}.npz")
    if grid_file and os.path.exists(gf):
        ref = np.load(gf)
        for name in ref.files:
            if name in g.fields and ref[name].shape == g.fields[name].shape:
                sc = np.abs(ref[name]).max()
                grid_diffs[name] = float(np.abs(ref[name] - g.fields[name])[3:-4, 3:-4].max() / (sc if sc > 0 else 1.0))
                g.fields[name] = np.array(ref[name])  # the operators run on the reference's own metric terms
        for name in ("ak", "bk"):
            if name in ref.files:
                setattr(g, name, np.array(ref[name]).ravel())
    cfg = AcousticDynamicsConfig(npx=nx + 1, npy=nx + 1, npz=nz, layout=(1, 1))
    sf = stencil_factory_for(backend)([g], cfg, get_constants())
    qf = sf.quantity_factory
    nzp = nz + 1
    Q = {n: qf.from_array([_pad(inp[v], nzp)], ("x", "y", "z")) for n, v in (("delp", "delpd"), ("pt", "ptd"), ("u", "ud"), ("v", "vd"), ("w", "wd"))}
    T = {n: qf.zeros(("x", "y", "z")) for n in ("uc", "vc", "ua", "va", "ut", "vt", "divgd", "omga", "delpc", "ptc")}
    dt2 = 0.5 * cfg.dt_atmos / cfg.k_split / cfg.n_split
    sf.call("c_sw", Q["delp"].fref, Q["pt"].fref, Q["u"].fref, Q["v"].fref, Q["w"].fref, T["uc"].fref, T["vc"].fref, T["ua"].fref, T["va"].fref, T["ut"].fref,
            T["vt"].fref, T["divgd"].fref, T["omga"].fref, T["delpc"].fref, T["ptc"].fref, dt2)
    errs = {}
    o = 2  # storage offset of local index 1 is n_halo = 3 -> python index 3; regions below are given for local indices
    for var, (ours, (i0, di, j0, dj)) in CSW_OUT.items():
        if var not in out:
            continue
        sl = (slice(i0 + o, nx + di + o + 1), slice(j0 + o, nx + dj + o + 1), slice(0, nz))
        got, want = T[ours].numpy(0)[sl], out[var][sl]
        sc = np.abs(want).max()
        errs[var] = float(np.abs(got - want).max() / (sc if sc > 0 else 1.0))
    return errs, grid_diffs


# D_SW savepoint variable -> (our name, region offsets as above); variable list [REF tests/savepoint/thresholds/fv_dynamics.yaml:76-170]
DSW_OUT = {"delpd": ("delp", (1, 0, 1, 0)), "ptd": ("pt", (1, 0, 1, 0)), "wd": ("w", (1, 0, 1, 0)), "ud": ("u", (1, 0, 1, 1)), "vd": ("v", (1, 1, 1, 0)),
           "mfxd": ("mfxd", (1, 1, 1, 0)), "mfyd": ("mfyd", (1, 0, 1, 1)), "xfxd": ("xfx", (1, 1, 1, 0)), "yfxd": ("yfx", (1, 0, 1, 1)),
           "divgdd": ("divgd", (1, 1, 1, 1))}


def check_d_sw_savepoints(path, backend, rank=0, nx=12, grid_file=True):
    """Feed D_SW-In of call 0 to fv3_d_sw and compare with D_SW-Out; returns {var: max field-relative error}.  The reference's
    checkpoint carries no q_con / cx / cy (dry run: q_con = 0; the accumulated Courant numbers are not compared)."""
    inp = dict(np.load(os.path.join(path, f"D_SW-In_call0_rank{rank}.npz")))
    out = dict(np.load(os.path.join(path, f"D_SW-Out_call0_rank{rank}.npz")))
    nz = inp["delpd"].shape[2] - 1
    part = CubedSpherePartitioner(nx, (1, 1))
    g = make_grid(part, rank, nz=nz)
    gf = os.path.join(path, f"grid_rank{rank}.npz")
    if grid_file and os.path.exists(gf):
        ref = np.load(gf)
        for name in ref.files:
            if name in g.fields and ref[name].shape == g.fields[name].shape:
                g.fields[name] = np.array(ref[name])
        for name in ("ak", "bk"):
            if name in ref.files:
                setattr(g, name, np.array(ref[name]).ravel())
    cfg = AcousticDynamicsConfig(npx=nx + 1, npy=nx + 1, npz=nz, layout=(1, 1))
    sf = stencil_factory_for(backend)([g], cfg, get_constants())
    qf = sf.quantity_factory
    nzp = nz + 1
    src = {"delpc": "delpcd", "delp": "delpd", "pt": "ptd", "u": "ud", "v": "vd", "w": "wd", "uc": "ucd", "vc": "vcd", "ua": "uad", "va": "vad", "divgd": "divgdd",
           "mfxd": "mfxd", "mfyd": "mfyd", "zh": "zhd"}
    Q = {n: qf.from_array([_pad(inp[v], nzp)], ("x", "y", "z")) for n, v in src.items() if v in inp}
    for n in ("zh", "delpc"):
        Q.setdefault(n, qf.zeros(("x", "y", "z")))
    T = {n: qf.zeros(("x", "y", "z")) for n in ("cxd", "cyd", "crx", "cry", "xfx", "yfx", "q_con", "heat", "diss")}
    dt = cfg.dt_atmos / cfg.k_split / cfg.n_split
    sf.call("d_sw", Q["delpc"].fref, *[Q[n].fref for n in ("delp", "pt", "u", "v", "w", "uc", "vc", "ua", "va", "divgd", "mfxd", "mfyd")], T["cxd"].fref, T["cyd"].fref,
            T["crx"].fref, T["cry"].fref, T["xfx"].fref, T["yfx"].fref, T["q_con"].fref, Q["zh"].fref, T["heat"].fref, T["diss"].fref, dt)
    Q.update(T)
    errs = {}
    o = 2
    for var, (ours, (i0, di, j0, dj)) in DSW_OUT.items():
        if var not in out:
            continue
        sl = (slice(i0 + o, nx + di + o + 1), slice(j0 + o, nx + dj + o + 1), slice(0, nz))
        got, want = Q[ours].numpy(0)[sl], out[var][sl]
        sc = np.abs(want).max()
        errs[var] = float(np.abs(got - want).max() / (sc if sc > 0 else 1.0))
    return errs


def test_d_sw_against_reference_savepoints(hostemu):
    if not glob.glob(os.path.join(GOLDEN, "D_SW-In_call0_rank*.npz")):
        pytest.skip("reference parity unpinned: tests/golden/golden_c12 is absent (generate it with tools/gen_golden.py where pyFV3 imports)")
    errs = check_d_sw_savepoints(GOLDEN, "hostemu")
    # magnitudes of the reference's own thresholds [REF tests/savepoint/thresholds/fv_dynamics.yaml:125-170]
    bad = {k: v for k, v in errs.items() if v > 1e-9}
    assert not bad, f"d_sw differs from the reference savepoints: {bad} (all: {errs})"


def test_d_sw_checker_on_oracle_generated_files(hostemu, tmp_path):
    """Write D_SW-In/Out files in the generator's format from the numpy oracle and run the checker on them."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle"))
    from fv3_oracle import d_sw as o_dsw
    from fv3_oracle.util import Dom
    from pace_amd.init import synthetic_state

    nx, nz = 12, 5
    part = CubedSpherePartitioner(nx, (1, 1))
    g = make_grid(part, 0, nz=nz)
    cfg = AcousticDynamicsConfig(npx=nx + 1, npy=nx + 1, npz=nz, layout=(1, 1))
    D = Dom(g, get_constants())
    x = {k: v[:, :, :nz].copy() for k, v in synthetic_state(g, seed=5, rank=0).items() if k != "phis"}
    x["uc"], x["vc"] = 0.7 * x["v"], 0.7 * x["u"]
    x["ua"], x["va"] = 0.9 * np.roll(x["u"], 1, 1), 0.9 * np.roll(x["v"], 1, 0)
    x["divgd"], x["q_con"] = 1e-6 * x["w"], np.zeros_like(x["u"])
    x["mfxd"], x["mfyd"], x["cxd"], x["cyd"] = (np.zeros_like(x["u"]) for _ in range(4))
    pad = lambda a: _pad(a, nz + 1)  # noqa: E731
    z = lambda: np.zeros_like(x["u"])  # noqa: E731
    np.savez(tmp_path / "D_SW-In_call0_rank0.npz", delpcd=pad(z()), zhd=pad(z()), **{v: pad(x[n]) for n, v in (("delp", "delpd"), ("pt", "ptd"), ("u", "ud"), ("v", "vd"),
             ("w", "wd"), ("uc", "ucd"), ("vc", "vcd"), ("ua", "uad"), ("va", "vad"), ("divgd", "divgdd"), ("mfxd", "mfxd"), ("mfyd", "mfyd"))})
    o = dict(delpc=z(), crx=z(), cry=z(), xfx=z(), yfx=z(), heat=z(), diss=z())
    dt = cfg.dt_atmos / cfg.k_split / cfg.n_split
    o_dsw.d_sw(D, cfg, o_dsw.get_column_namelist(cfg, nz), o["delpc"], x["delp"], x["pt"], x["u"], x["v"], x["w"], x["uc"], x["vc"], x["ua"], x["va"], x["divgd"], x["mfxd"],
               x["mfyd"], x["cxd"], x["cyd"], o["crx"], o["cry"], o["xfx"], o["yfx"], x["q_con"], None, o["heat"], o["diss"], dt)
    np.savez(tmp_path / "D_SW-Out_call0_rank0.npz", delpd=pad(x["delp"]), ptd=pad(x["pt"]), wd=pad(x["w"]), ud=pad(x["u"]), vd=pad(x["v"]), mfxd=pad(x["mfxd"]),
             mfyd=pad(x["mfyd"]), xfxd=pad(o["xfx"]), yfxd=pad(o["yfx"]), divgdd=pad(x["divgd"]))
    errs = check_d_sw_savepoints(str(tmp_path), "hostemu", grid_file=False)
    assert set(errs) == set(DSW_OUT) and max(errs.values()) < 1e-11, errs


def test_c_sw_against_reference_savepoints(hostemu):
    if not glob.glob(os.path.join(GOLDEN, "C_SW-In_call0_rank*.npz")):
        pytest.skip("reference parity unpinned: tests/golden/golden_c12 is absent (generate it with tools/gen_golden.py where pyFV3 imports)")
    errs, grid_diffs = check_c_sw_savepoints(GOLDEN, "hostemu")
    worst_grid = {k: v for k, v in grid_diffs.items() if v > 1e-9}
    assert not worst_grid, f"metric terms differ from the reference's MetricTerms: {worst_grid}"
    # magnitudes of the reference's own thresholds [REF tests/savepoint/thresholds/fv_dynamics.yaml:2-75]
    bad = {k: v for k, v in errs.items() if v > 1e-10}
    assert not bad, f"c_sw differs from the reference savepoints: {bad} (all: {errs})"


def test_savepoint_checker_on_oracle_generated_files(hostemu, tmp_path):
    """Write C_SW-In/Out files in the generator's format from the numpy oracle and run the checker on them."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle"))
    from fv3_oracle import c_sw as o_csw
    from fv3_oracle.util import Dom
    from pace_amd.init import synthetic_state

    nx, nz = 12, 5
    part = CubedSpherePartitioner(nx, (1, 1))
    g = make_grid(part, 0, nz=nz)
    cfg = AcousticDynamicsConfig(npx=nx + 1, npy=nx + 1, npz=nz, layout=(1, 1))
    D = Dom(g, get_constants())
    s = {k: v[:, :, :nz].copy() for k, v in synthetic_state(g, seed=3, rank=0).items() if k != "phis"}
    pad = lambda a: _pad(a, nz + 1)  # noqa: E731
    np.savez(tmp_path / "C_SW-In_call0_rank0.npz", delpd=pad(s["delp"]), ptd=pad(s["pt"]), ud=pad(s["u"]), vd=pad(s["v"]), wd=pad(s["w"]))
    ut, vt, div = np.zeros_like(s["u"]), np.zeros_like(s["u"]), np.zeros_like(s["u"])
    dt2 = 0.5 * cfg.dt_atmos / cfg.k_split / cfg.n_split
    o_csw.c_sw(D, s["delp"], s["pt"], s["u"], s["v"], s["w"], s["uc"], s["vc"], s["ua"], s["va"], ut, vt, div, s["omga"], dt2, nord=cfg.nord)
    np.savez(tmp_path / "C_SW-Out_call0_rank0.npz", ucd=pad(s["uc"]), vcd=pad(s["vc"]), uad=pad(s["ua"]), vad=pad(s["va"]), divgdd=pad(div))
    errs, _ = check_c_sw_savepoints(str(tmp_path), "hostemu", grid_file=False)
    assert set(errs) == set(CSW_OUT) and max(errs.values()) < 1e-12, errs
