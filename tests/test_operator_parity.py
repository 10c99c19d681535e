"""Operator-level parity: every operator of the acoustic sub-step is called ALONE through the C ABI on the
inputs the oracle's own sequence hands to that operator, and every output (the intermediates ws3, gz, pkc,
wsd, zh, pk3, uc / vc, heat_source included) is compared with what the oracle's operator produced from
the same inputs.  [REF docs/testing.rst:18-23: one test per module, inputs / outputs at the module boundary]

The inputs come from a *recording* run of the oracle sequence (oracle/fv3_oracle/dyn_core.py): each oracle
operator is wrapped so that its arguments are copied before and after the call.  An error therefore cannot
hide behind a later operator that overwrites or smooths the field, and uc / vc / the interface fields --
dead data after a full call -- are compared where they are live.
"""
import numpy as np
import pytest
import torch

from helpers import assert_close, oracle_cube
from pace_amd.constants import get_constants
from pace_amd._testing import stencil_factory_for
from pace_amd.context import StencilFactory

import fv3_oracle.dyn_core as o_dyn

OPS = ("update_dz_c", "riem_solver_c", "p_grad_c", "update_dz_d", "riem_solver3", "nh_p_grad", "ray_fast", "del2_cubed", "apply_diffusive_heating", "pk3_halo", "pe_halo")


class Recorder:
    """Wraps the operators the oracle sequence calls (module attributes of dyn_core._nh / _d_sw / _c_sw)."""

    def __init__(self):
        self.calls = {}
        self._saved = []

    def _wrap(self, mod, name):
        fn = getattr(mod, name)
        rec = self

        def w(D, *args, **kw):
            cp = lambda a: a.copy() if isinstance(a, np.ndarray) else a  # noqa: E731
            ins = [cp(a) for a in args]
            ret = fn(D, *args, **kw)
            outs = [cp(a) for a in args]
            rec.calls.setdefault(name, []).append(dict(ins=ins, outs=outs, ret=ret, D=D))
            return ret

        self._saved.append((mod, name, fn))
        setattr(mod, name, w)

    def __enter__(self):
        for n in OPS:
            self._wrap(o_dyn._nh, n)
        self._wrap(o_dyn._d_sw, "d_sw")
        self._wrap(o_dyn._c_sw, "c_sw")
        return self

    def __exit__(self, *a):
        for mod, name, fn in self._saved:
            setattr(mod, name, fn)


@pytest.fixture(params=["hostemu", pytest.param("hip:gfx950", marks=pytest.mark.gpu)])
def backend(request):
    request.getfixturevalue("hostemu" if request.param == "hostemu" else "gpu_backend")
    return request.param


_CACHE = {}


def recorded(layout, nz=8):
    """One recorded oracle call (n_split = 1) on the whole C12 cube: calls[name][rank]."""
    key = (layout, nz)
    if key not in _CACHE:
        part, cfg, grids, ost, phis, odyn = oracle_cube(12, layout, nz, dict(n_split=1))
        with Recorder() as rec:
            odyn(ost, 225.0, 1)
        _CACHE[key] = (part, cfg, grids, rec.calls, odyn)
    return _CACHE[key]


class Dev:
    """All ranks of the cube in one device context; Quantities from lists of per-rank oracle arrays."""

    def __init__(self, backend, grids, cfg):
        self.sf = stencil_factory_for(backend)(grids, cfg, get_constants())
        self.qf = self.sf.quantity_factory
        self.nz = grids[0].nz

    def q(self, arrays):
        a0 = np.asarray(arrays[0])
        two_d = a0.ndim == 2 or a0.shape[2] == 1
        if two_d:
            return self.qf.from_array([np.asarray(a).reshape(a0.shape[0], a0.shape[1]) for a in arrays], ("x", "y"))
        if a0.shape[2] == self.nz:  # level views of the oracle -> padded storage
            arrays = [np.concatenate([a, a[:, :, -1:]], axis=2) for a in arrays]
        return self.qf.from_array([np.asarray(a) for a in arrays], ("x", "y", "z"))


def _cmp(name, q, calls, idx, region, tol, kk=None, two_d=False):
    worst = 0.0
    for r, c in enumerate(calls):
        D = c["D"]
        want = c["outs"][idx].copy()
        got = q.numpy(r).copy()
        R = region(D)
        if region is RING1:
            # the cube-corner halo cell of a cell-centred field belongs to no neighbour: never read, not compared
            for (ci, cj), has in (((0, 0), D.sw), ((D.nx + 1, 0), D.se), ((D.nx + 1, D.ny + 1), D.ne), ((0, D.ny + 1), D.nw)):
                if has:
                    got[D.sl(ci, ci, cj, cj)] = 0.0
                    want[D.sl(ci, ci, cj, cj)] = 0.0
        if two_d:
            want = want.reshape(want.shape[0], want.shape[1])
            worst = max(worst, assert_close(f"{name} rank {r}", got[R], want[R], tol, 0.0))
        else:
            n = want.shape[2] if kk is None else kk
            worst = max(worst, assert_close(f"{name} rank {r}", got[R][:, :, :n], want[R][:, :, :n], tol, 0.0))
    return worst


CELLS = lambda D: D.sl(1, D.nx, 1, D.ny)  # noqa: E731
RING1 = lambda D: D.sl(0, D.nx + 1, 0, D.ny + 1)  # noqa: E731
LAYOUTS = [(1, 1), (2, 2)]


@pytest.mark.parametrize("layout", LAYOUTS)
def test_update_dz_c(backend, layout):
    """a3: gz (interface heights on the compute domain + 1 ring) and ws3.  update_dz_c(D, dp_ref, zs, ut, vt, gz, ws, dt)"""
    part, cfg, grids, calls, _ = recorded(layout)
    cl = calls["update_dz_c"]
    dv = Dev(backend, grids, cfg)
    I = lambda i: dv.q([c["ins"][i] for c in cl])  # noqa: E731
    zs, ut, vt, gz, ws = I(1), I(2), I(3), I(4), I(5)
    dv.sf.call("update_dz_c", zs.fref, ut.fref, vt.fref, gz.fref, ws.fref, float(cl[0]["ins"][6]))
    _cmp("gz", gz, cl, 4, RING1, 1e-13)
    _cmp("ws3", ws, cl, 5, RING1, 1e-11, two_d=True)  # (zs - gz) / dt: a difference of two ~1e4 m heights scaled to ~1 m/s


@pytest.mark.parametrize("layout", LAYOUTS)
def test_riem_solver_c(backend, layout):
    """a4: gz and pef (= pkc) on compute + 1 ring.  riem_solver_c(D, dt2, cappa, ptop, phis, ws, ptc, q_con, delpc, gz, pef, w3, p_fac)"""
    part, cfg, grids, calls, _ = recorded(layout)
    cl = calls["riem_solver_c"]
    dv = Dev(backend, grids, cfg)
    I = lambda i: dv.q([c["ins"][i] for c in cl])  # noqa: E731
    a = cl[0]["ins"]
    cappa, phis, ws, ptc, q_con, delpc, gz, pef, w3 = I(1), I(3), I(4), I(5), I(6), I(7), I(8), I(9), I(10)
    dv.sf.call("riem_solver_c", float(a[0]), cappa.fref, float(a[2]), phis.fref, ws.fref, ptc.fref, q_con.fref, delpc.fref, gz.fref, pef.fref, w3.fref)
    _cmp("gz", gz, cl, 8, RING1, 1e-12)
    _cmp("pef", pef, cl, 9, RING1, 1e-12)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_p_grad_c(backend, layout):
    """a5: uc, vc after the C-grid pressure gradient.  p_grad_c(D, rdxc, rdyc, uc, vc, delpc, pkc, gz, dt2)"""
    part, cfg, grids, calls, _ = recorded(layout)
    cl = calls["p_grad_c"]
    dv = Dev(backend, grids, cfg)
    I = lambda i: dv.q([c["ins"][i] for c in cl])  # noqa: E731
    uc, vc, delpc, pkc, gz = I(2), I(3), I(4), I(5), I(6)
    dv.sf.call("p_grad_c", uc.fref, vc.fref, delpc.fref, pkc.fref, gz.fref, float(cl[0]["ins"][7]))
    _cmp("uc", uc, cl, 2, lambda D: D.sl(1, D.nx + 1, 1, D.ny), 1e-13, kk=dv.nz)
    _cmp("vc", vc, cl, 3, lambda D: D.sl(1, D.nx, 1, D.ny + 1), 1e-13, kk=dv.nz)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_d_sw_on_recorded_inputs(backend, layout):
    """a6 on the inputs of the real sequence (c_sw winds, halo-updated fields), 1 x 1 and 2 x 2 ranks: every output incl.
    the Courant numbers / area fluxes, the accumulated mass fluxes and the heat source."""
    part, cfg, grids, calls, _ = recorded(layout)
    cl = calls["d_sw"]
    dv = Dev(backend, grids, cfg)
    nz = dv.nz
    # d_sw(D, cfg, col, delpc, delp, pt, u, v, w, uc, vc, ua, va, divgd, mfx, mfy, cx, cy, crx, cry, xfx, yfx, q_con, zh, heat_source, diss_est, dt)
    names = "delpc delp pt u v w uc vc ua va divgd mfx mfy cx cy crx cry xfx yfx q_con zh heat_source diss_est".split()
    Q = {}
    for n, name in enumerate(names):
        Q[name] = dv.q([c["ins"][2 + n] for c in cl])
    dv.sf.call("d_sw", *[Q[n].fref for n in names], float(cl[0]["ins"][25]))
    idx = {n: 2 + i for i, n in enumerate(names)}
    U = lambda D: D.sl(1, D.nx, 1, D.ny + 1)  # noqa: E731
    V = lambda D: D.sl(1, D.nx + 1, 1, D.ny)  # noqa: E731
    for n in ("delp", "pt", "w", "q_con"):
        _cmp(n, Q[n], cl, idx[n], CELLS, 1e-13, kk=nz)
    _cmp("heat_source", Q["heat_source"], cl, idx["heat_source"], CELLS, 1e-12, kk=nz)
    _cmp("u", Q["u"], cl, idx["u"], U, 1e-13, kk=nz)
    _cmp("v", Q["v"], cl, idx["v"], V, 1e-13, kk=nz)
    for n, R in (("mfx", V), ("cx", lambda D: D.sl(1, D.nx + 1, D.jsd, D.jed)), ("crx", lambda D: D.sl(1, D.nx + 1, D.jsd, D.jed)), ("xfx", lambda D: D.sl(1, D.nx + 1, D.jsd, D.jed)),
                 ("mfy", U), ("cy", lambda D: D.sl(D.isd, D.ied, 1, D.ny + 1)), ("cry", lambda D: D.sl(D.isd, D.ied, 1, D.ny + 1)), ("yfx", lambda D: D.sl(D.isd, D.ied, 1, D.ny + 1))):
        _cmp(n, Q[n], cl, idx[n], R, 1e-13, kk=nz)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_update_dz_d(backend, layout):
    """a7: zh on the compute domain and wsd.  update_dz_d(D, cfg, col, dp_ref, zs, zh, crx, cry, xfx, yfx, ws, dt)"""
    part, cfg, grids, calls, _ = recorded(layout)
    cl = calls["update_dz_d"]
    dv = Dev(backend, grids, cfg)
    I = lambda i: dv.q([c["ins"][i] for c in cl])  # noqa: E731
    zs, zh, crx, cry, xfx, yfx, ws = I(3), I(4), I(5), I(6), I(7), I(8), I(9)
    dv.sf.call("update_dz_d", zs.fref, zh.fref, crx.fref, cry.fref, xfx.fref, yfx.fref, ws.fref, float(cl[0]["ins"][10]))
    _cmp("zh", zh, cl, 4, CELLS, 1e-13)
    _cmp("wsd", ws, cl, 9, CELLS, 1e-10, two_d=True)  # (zs - zh) / dt as above; dt = 225 s here


@pytest.mark.parametrize("layout", LAYOUTS)
def test_riem_solver3_on_recorded_inputs(backend, layout):
    """a8 on the real sequence's inputs.  riem_solver3(D, last_call, dt, cappa, ptop, zs, ws, delz, q_con, delp, pt, zh, pe, ppe, pk3, pk, peln, w, p_fac)"""
    part, cfg, grids, calls, _ = recorded(layout)
    cl = calls["riem_solver3"]
    dv = Dev(backend, grids, cfg)
    I = lambda i: dv.q([c["ins"][i] for c in cl])  # noqa: E731
    a = cl[0]["ins"]
    Q = {n: I(i) for n, i in (("cappa", 2), ("zs", 4), ("ws", 5), ("delz", 6), ("q_con", 7), ("delp", 8), ("pt", 9), ("zh", 10), ("pe", 11), ("ppe", 12), ("pk3", 13), ("pk", 14), ("peln", 15), ("w", 16))}
    dv.sf.call("riem_solver3", int(bool(a[0])), float(a[1]), Q["cappa"].fref, float(a[3]), Q["zs"].fref, Q["ws"].fref, Q["delz"].fref, Q["q_con"].fref, Q["delp"].fref, Q["pt"].fref,
               Q["zh"].fref, Q["pe"].fref, Q["ppe"].fref, Q["pk3"].fref, Q["pk"].fref, Q["peln"].fref, Q["w"].fref)
    nz = dv.nz
    # ppe is the perturbation pressure (~1e2 Pa against pe ~1e5 Pa): a difference of exp/log results, compared on ITS scale
    for n, i, tol, kk in (("w", 16, 1e-10, nz), ("delz", 6, 1e-12, nz), ("zh", 10, 1e-13, nz + 1), ("ppe", 12, 1e-9, nz + 1), ("pk3", 13, 1e-13, nz + 1), ("pe", 11, 1e-14, nz + 1), ("pk", 14, 1e-13, nz + 1),
                          ("peln", 15, 1e-14, nz + 1)):
        _cmp(n, Q[n], cl, i, CELLS, tol, kk=kk)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_pk3_halo_and_edge_pe(backend, layout):
    """a9: the 2-wide ring of pk3 and the 1-wide ring of pe, straight after riem_solver3."""
    part, cfg, grids, calls, _ = recorded(layout)
    dv = Dev(backend, grids, cfg)
    cl = calls["pk3_halo"]
    pk3, delp = dv.q([c["ins"][0] for c in cl]), dv.q([c["ins"][1] for c in cl])
    dv.sf.call("pk3_halo", pk3.fref, delp.fref, float(cl[0]["ins"][2]), float(cl[0]["ins"][3]))
    for r, c in enumerate(cl):
        D = c["D"]
        R2 = D.sl(-1, D.nx + 2, -1, D.ny + 2)
        assert_close("pk3 ring", pk3.numpy(r)[R2][:, :, 1:], c["outs"][0][R2][:, :, 1:], 1e-13, 0.0)
    cl = calls["pe_halo"]
    pe, delp = dv.q([c["ins"][0] for c in cl]), dv.q([c["ins"][1] for c in cl])
    dv.sf.call("edge_pe", pe.fref, delp.fref, float(cl[0]["ins"][2]))
    _cmp("pe ring", pe, cl, 0, RING1, 1e-14)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_nh_p_grad(backend, layout):
    """a10 + a11: u, v after the non-hydrostatic pressure gradient (gz = g * zh formed by the caller, as in the reference).
    nh_p_grad(D, u, v, pp, gz, pk3, delp, dt, ptop, akap)"""
    part, cfg, grids, calls, _ = recorded(layout)
    cl = calls["nh_p_grad"]
    dv = Dev(backend, grids, cfg)
    I = lambda i: dv.q([c["ins"][i] for c in cl])  # noqa: E731
    a = cl[0]["ins"]
    u, v, pp, gz, pk3, delp = I(0), I(1), I(2), I(3), I(4), I(5)
    dv.sf.call("nh_p_grad", u.fref, v.fref, pp.fref, gz.fref, pk3.fref, delp.fref, float(a[6]), float(a[7]), float(a[8]))
    _cmp("u", u, cl, 0, lambda D: D.sl(1, D.nx, 1, D.ny + 1), 1e-12, kk=dv.nz)
    _cmp("v", v, cl, 1, lambda D: D.sl(1, D.nx + 1, 1, D.ny), 1e-12, kk=dv.nz)


def test_ray_fast(backend):
    """a12 at L79 (the reference's level set: ~10 levels above rf_cutoff).  ray_fast(D, cfg, u, v, w, dp, pfull, dt, ptop)"""
    part, cfg, grids, calls, _ = recorded((1, 1), nz=79)
    cl = calls["ray_fast"]
    dv = Dev(backend, grids, cfg)
    I = lambda i: dv.q([c["ins"][i] for c in cl])  # noqa: E731
    a = cl[0]["ins"]
    u, v, w = I(1), I(2), I(3)
    dv.sf.call("ray_fast", u.fref, v.fref, w.fref, float(a[6]), float(a[7]))
    changed = max(np.abs(c["outs"][1] - c["ins"][1]).max() for c in cl)
    assert changed > 0.0, "the recorded case does not exercise the Rayleigh layer"
    _cmp("u", u, cl, 1, lambda D: D.sl(1, D.nx, 1, D.ny + 1), 1e-13, kk=79)
    _cmp("v", v, cl, 2, lambda D: D.sl(1, D.nx + 1, 1, D.ny), 1e-13, kk=79)
    _cmp("w", w, cl, 3, CELLS, 1e-13, kk=79)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_del2_cubed_and_diffusive_heating(backend, layout):
    """a13: the smoothed heat source and pt after the heating.  del2_cubed(D, q, cd, nmax); apply_diffusive_heating(D, delp, delz, cappa, heat_source, pt, f)"""
    part, cfg, grids, calls, _ = recorded(layout)
    dv = Dev(backend, grids, cfg)
    cl = calls["del2_cubed"]
    q = dv.q([c["ins"][0] for c in cl])
    dv.sf.call("del2_cubed", q.fref, float(cl[0]["ins"][1]), 3)
    assert max(np.abs(c["ins"][0]).max() for c in cl) > 0.0
    _cmp("heat_source", q, cl, 0, CELLS, 1e-12, kk=dv.nz)
    cl = calls["apply_diffusive_heating"]
    I = lambda i: dv.q([c["ins"][i] for c in cl])  # noqa: E731
    delp, delz, cappa, hs, pt = I(0), I(1), I(2), I(3), I(4)
    dv.sf.call("apply_diffusive_heating", delp.fref, delz.fref, cappa.fref, hs.fref, pt.fref, float(cl[0]["ins"][5]))
    _cmp("pt", pt, cl, 4, CELLS, 1e-14, kk=dv.nz)


def test_pair_debug_report(hostemu, capsys):
    """tests/pair_debug.py (the side-by-side run mode, REF driver.py:83-87 pair_debug): every operator replayed against the
    oracle on a small cube; the report attributes the worst difference (libm round-off in the Riemann solvers) to its operator."""
    import pair_debug

    assert pair_debug.main(["--nx", "12", "--nz", "6", "--backend", "hostemu", "--n-split", "2", "--tol", "1e-10"]) == 0
    out = capsys.readouterr().out
    assert "riem_solver3" in out and "d_sw" in out and "worst difference" in out


def test_operator_wrappers_refuse_columns_that_differ_from_the_context():
    """dp_ref / pfull / ks / rdxc / rdyc live in the context (uploaded once); the reference passes them at every call.  The
    wrappers accept the context's values (or None) and refuse anything else instead of silently using their own copy."""
    from helpers import Case

    from pace_amd import stencils as st

    cs = Case(nx_tile=12, nz=8)
    sf, g = cs.sf, cs.grids[0]
    s = cs.states[0]
    u, v, w = cs.q([s["u"]]), cs.q([s["v"]]), cs.q([s["w"]])
    ray = st.RayleighDamping(sf)
    ray(u, v, w, g.dp_ref, g.pfull, 10.0, g.ptop, ks=g.ks)  # the context's values: accepted
    ray(u, v, w, None, None, 10.0, g.ptop)
    with pytest.raises(ValueError, match="pfull"):
        ray(u, v, w, g.dp_ref, g.pfull * 1.001, 10.0, g.ptop)
    with pytest.raises(ValueError, match="dp_ref"):
        ray(u, v, w, g.dp_ref[:-1], g.pfull, 10.0, g.ptop)
    with pytest.raises(ValueError, match="ks"):
        ray(u, v, w, g.dp_ref, g.pfull, 10.0, g.ptop, ks=g.ks + 1)
    pg = st.PGradC(sf)
    uc, vc, dpc, pkc, gz = (cs.q() for _ in range(5))
    dpc.storage.fill_(1.0)
    pg(sf.grid_fields["rdxc"], sf.grid_fields["rdyc"], uc, vc, dpc, pkc, gz, 5.0)
    other = cs.qf.zeros(("x", "y"))
    other.storage.copy_(sf.grid_fields["rdxc"].storage * 2.0)
    with pytest.raises(ValueError, match="rdxc"):
        pg(other, sf.grid_fields["rdyc"], uc, vc, dpc, pkc, gz, 5.0)
    with pytest.raises(ValueError, match="dp_ref"):
        st.UpdateGeopotentialHeightOnCGrid(sf)(g.dp_ref * 2.0, None, None, None, None, None, 1.0)
