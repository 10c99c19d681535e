"""TracerAdvection (tracer_2d_1l) -- the oracle's properties and the library against the oracle.
Reference call shape: tracer_advection(tracers, dp1, mfxd, mfyd, cxd, cyd) [REF examples/notebooks/functions.py:1037-1044]
with a FiniteVolumeTransport of hord 6 (as in that notebook, :933)."""
import numpy as np
import pytest

from helpers import assert_close, oracle_cube
from pace_amd.constants import get_constants
from pace_amd._testing import stencil_factory_for
from pace_amd.context import StencilFactory
from pace_amd.halo import Layout
from pace_amd.stencils import FiniteVolumeTransport, TracerAdvection

from fv3_oracle import tracer_2d_1l as o_t2


@pytest.fixture(params=["hostemu", pytest.param("hip:gfx950", marks=pytest.mark.gpu)])
def backend(request):
    request.getfixturevalue("hostemu" if request.param == "hostemu" else "gpu_backend")
    return request.param


def _inputs(n, layout, nz, want_split, n_tracers):
    """Accumulated fluxes of one acoustic call (2 sub-steps) of the oracle as the advection's inputs; the Courant numbers are
    scaled so that the operator needs ``want_split`` sub-cycles."""
    part, cfg, grids, ost, phis, odyn = oracle_cube(n, layout, nz, dict(n_split=2))
    dp1 = [s["delp"][:, :, :nz].copy() for s in ost]  # the air mass BEFORE the acoustic call
    odyn(ost, 900.0, 1)
    V = lambda a: a[:, :, :nz].copy()  # noqa: E731
    rng = np.random.default_rng(3)
    tr = []
    for s in ost:
        d = {}
        for t in range(n_tracers):
            base = V(s["q_con"]) * 1e3 if t % 2 == 0 else V(s["pt"]) * 0.01
            d[f"q{t}"] = base * (1.0 + 0.1 * t) + 0.5
        tr.append(d)
    base = max(max(np.abs(V(s["cxd"])[D.sl(1, D.nx, 1, D.ny)]).max(), np.abs(V(s["cyd"])[D.sl(1, D.nx, 1, D.ny)]).max()) for D, s in zip(odyn.doms, ost))
    courant_scale = 1.0 if want_split == 1 else (want_split - 0.6) / base  # n_split = int(1 + cmax), cmax = max|c| + 1 - sin_sg5, the last term in [0, 0.14]
    F = dict(dp1=dp1, mfx=[V(s["mfxd"]) for s in ost], mfy=[V(s["mfyd"]) for s in ost], cx=[V(s["cxd"]) * courant_scale for s in ost],
             cy=[V(s["cyd"]) * courant_scale for s in ost])
    for name in tr[0]:
        odyn.ex.scalar([t_[name] for t_ in tr])
    odyn.ex.scalar(F["dp1"])
    return part, cfg, grids, odyn, tr, F


def test_oracle_keeps_a_constant_tracer_constant_and_conserves_tracer_mass():
    """q == const stays const (dp2 is built from the same mass fluxes the tracer rides on) and sum(q dp area) is conserved
    over the cube -- with sub-cycling (Courant numbers scaled up until n_split = 3)."""
    nz = 4
    part, cfg, grids, odyn, tr, F = _inputs(12, (1, 1), nz, 3, 1)
    for t_ in tr:
        t_["one"] = np.full_like(t_["q0"], 0.75)
    doms = odyn.doms

    def mass(name, dp):
        return sum(float((t_[name][D.sl(1, D.nx, 1, D.ny)] * d[D.sl(1, D.nx, 1, D.ny)] * D.m.area[D.sl(1, D.nx, 1, D.ny)]).sum()) for t_, d, D in zip(tr, dp, doms))

    m0 = mass("q0", F["dp1"])
    mfx, mfy = [a.copy() for a in F["mfx"]], [a.copy() for a in F["mfy"]]
    ns = o_t2.tracer_2d_1l(doms, tr, F["dp1"], mfx, mfy, F["cx"], F["cy"], 6, halo_update=lambda fs: odyn.ex.scalar(fs))
    assert ns == 3
    for t_, D in zip(tr, doms):
        C = D.sl(1, D.nx, 1, D.ny)
        assert np.abs(t_["one"][C] - 0.75).max() < 1e-14
    # final air mass = dp1 (before the last sub-cycle) + the last sub-cycle's flux divergence
    dp_end = []
    for r, D in enumerate(doms):
        C, Ce, Cn = D.sl(1, D.nx, 1, D.ny), D.sl(2, D.nx + 1, 1, D.ny), D.sl(1, D.nx, 2, D.ny + 1)
        d = F["dp1"][r].copy()
        d[C] = d[C] + (mfx[r][C] - mfx[r][Ce] + mfy[r][C] - mfy[r][Cn]) * D.m.rarea[C]
        dp_end.append(d)
    m1 = mass("q0", dp_end)
    assert abs(m1 - m0) <= 1e-13 * abs(m0)


@pytest.mark.parametrize("hord", [6, 8])
@pytest.mark.parametrize("n, layout, want_split, n_tracers", [(12, (1, 1), 1, 2), (12, (1, 1), 2, 3), (12, (2, 2), 3, 1), (65, (1, 1), 1, 2)])
def test_tracer_advection_matches_the_oracle(backend, n, layout, want_split, n_tracers, hord):
    """n_split = 1 and > 1 (Courant numbers scaled up), even / odd tracer counts, 2 x 2 ranks, multi-strip sub-domains."""
    nz = 4
    part, cfg, grids, odyn, tr, F = _inputs(n, layout, nz, want_split, n_tracers)
    sf = stencil_factory_for(backend)(grids, cfg, get_constants())
    qf = sf.quantity_factory
    pad = lambda a: np.concatenate([a, a[:, :, -1:]], axis=2)  # noqa: E731
    Q = {k: qf.from_array([pad(a) for a in v], ("x", "y", "z")) for k, v in F.items()}
    T = {name: qf.from_array([pad(t_[name]) for t_ in tr], ("x", "y", "z")) for name in tr[0]}
    op = TracerAdvection(sf, qf, FiniteVolumeTransport(sf, qf, grids, hord=hord), grids, Layout(part, 1, 0), T)
    op(T, Q["dp1"], Q["mfx"], Q["mfy"], Q["cx"], Q["cy"])
    ns = o_t2.tracer_2d_1l(odyn.doms, tr, F["dp1"], F["mfx"], F["mfy"], F["cx"], F["cy"], hord, halo_update=lambda fs: odyn.ex.scalar(fs))
    assert op.n_split == ns == want_split
    for r, D in enumerate(odyn.doms):
        C = D.sl(1, D.nx, 1, D.ny)
        for name in tr[0]:
            assert_close(f"{name} rank {r}", T[name].numpy(r)[:, :, :nz][C], tr[r][name][C], 1e-13, 0.0)
        assert_close("dp1", Q["dp1"].numpy(r)[:, :, :nz][C], F["dp1"][r][C], 1e-14, 0.0)
        assert_close("mfxd", Q["mfx"].numpy(r)[:, :, :nz][D.sl(1, D.nx + 1, 1, D.ny)], F["mfx"][r][D.sl(1, D.nx + 1, 1, D.ny)], 1e-14, 0.0)
        assert_close("cxd", Q["cx"].numpy(r)[:, :, :nz][D.sl(1, D.nx + 1, D.jsd, D.jed)], F["cx"][r][D.sl(1, D.nx + 1, D.jsd, D.jed)], 1e-14, 0.0)


def test_acoustic_call_plus_tracer_advection_is_mass_consistent(backend):
    """DynamicalCore.step_dynamics-shaped sequence (k_split = 2 acoustic calls, each followed by the tracer advection, no
    remap): the mass fluxes d_sw accumulated over a call rebuild the air mass the call ended with -- dp1 + div(mfxd, mfyd) == delp
    to round-off after EVERY call (which needs the accumulators emptied per call) -- so a constant tracer stays constant and
    tracer mass is conserved through the whole step."""
    from pace_amd._testing import harness_for
    from pace_amd.harness import DycoreHarness

    h = harness_for(backend)(24, nz=6, layout=(1, 1), dt_atmos=450.0, k_split=2, n_split=3, n_tracers=3, hord_tr=8)
    h.tracers["tracer2"].storage.fill_(0.25)
    nx, nz, nh = 24, 6, 3
    area = h.sf.grid_fields["area"].storage[:, nh : nh + nx, nh : nh + nx]
    rarea = h.sf.grid_fields["rarea"].storage[:, nh : nh + nx, nh : nh + nx]
    C = lambda q: q.storage[:, :nz, nh : nh + nx, nh : nh + nx]  # noqa: E731

    def tmass(name, dp):
        return float((C(h.tracers[name]) * dp * area[:, None]).sum())

    m0 = tmass("tracer0", C(h.state.delp))
    dt = 450.0 / 2
    for k in range(2):
        h.dp1.storage.copy_(h.state.delp.storage)
        h.dyn(h.state, dt, n_map=k + 1)
        mfx, mfy = h.state.mfxd.storage, h.state.mfyd.storage
        div = (mfx[:, :nz, nh : nh + nx, nh : nh + nx] - mfx[:, :nz, nh : nh + nx, nh + 1 : nh + nx + 1] + mfy[:, :nz, nh : nh + nx, nh : nh + nx]
               - mfy[:, :nz, nh + 1 : nh + nx + 1, nh : nh + nx]) * rarea[:, None]
        dp2 = C(h.dp1) + div
        err = float(((dp2 - C(h.state.delp)).abs() / C(h.state.delp)).max())
        assert err < 1e-13, f"call {k + 1}: accumulated mass fluxes do not rebuild the air mass ({err:.2e})"
        h._tracer_halo.update()
        h.tracer_advection(h.tracers, h.dp1, h.state.mfxd, h.state.mfyd, h.state.cxd, h.state.cyd)
    assert float((C(h.tracers["tracer2"]) - 0.25).abs().max()) < 1e-14
    m1 = tmass("tracer0", C(h.state.delp))
    assert abs(m1 - m0) <= 1e-12 * abs(m0), (m0, m1)
