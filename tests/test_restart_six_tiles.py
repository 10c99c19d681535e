"""The six-tile FV3 restart the reference tree holds (C12 L63, ``tests/golden/c12_restart_6tiles.npz``, made by
``tools/make_fixtures.py`` from ``tests/main/data/c12_restart/fv_core.res.tile[1-6].nc``) as a pin of the cube topology that is
NOT self-referential: the data were written by the Fortran model, so the adjacency / rotation / sign tables of
``pace_amd/topology.py`` either agree with FV3's or these tests fail.  [REF tests/main/driver/test_restart_fortran.py:21-67 reads
the same files; the halo machinery they exercise is docs/util/communication.rst:43-109]

* D-grid winds on an edge shared by two tiles are stored by BOTH tiles: ``synchronize_vector_interfaces`` (owner -> other side,
  with the rotation sign) must therefore be a bit-exact no-op on the real ``u`` / ``v`` -- and it stops being one as soon as an
  orientation or a sign of the map is changed (negative controls below).
* Scalar halos (T, delp, phis) filled from the neighbouring tile continue the tile's own field across all 12 edges better than
  the same halos with the along-edge order reversed.
* Vector halos: the along-edge wind component is continuous across every edge with the map's sign and jumps by ~2|V| with the
  sign flipped; so does the cross-edge component where the coordinate lines of the two tiles are parallel (edge mid-points).
* The real global state with its real terrain through the acoustic dynamics (+ tracer advection + vertical remap): the HIP
  library against the oracle for one call, then a dozen model steps -- finite, inside the reference's SafetyChecker bounds,
  air mass conserved, no grid-scale noise growing along the tile edges / at the cube corners.
"""
import os

import numpy as np
import pytest

from helpers import compare_cubes, oracle_cube, run_device_cube  # noqa: F401  (path setup)

from fv3_oracle.dyn_core import OracleAcousticDynamics, OracleExchange
from pace_amd.config import AcousticDynamicsConfig
from pace_amd.constants import get_constants
from pace_amd.grid import make_grid
from pace_amd.init import restart_state
from pace_amd.topology import CubedSpherePartitioner

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c12_restart_6tiles.npz")
N, NZ, NH = 12, 63, 3


@pytest.fixture(scope="module")
def data():
    return np.load(GOLDEN)


def _cube(data, name, ex_x=0, ex_y=0):
    """per-tile [i, j, k] arrays with NaN halos"""
    out = []
    for t in range(6):
        a = np.full((N + 2 * NH + 1, N + 2 * NH + 1, NZ), np.nan)
        a[NH : NH + N + ex_x, NH : NH + N + ex_y, :] = np.transpose(data[name][t], (2, 1, 0))
        out.append(a)
    return out


def test_shared_interface_winds_are_single_valued_under_the_topology_tables(data):
    part = CubedSpherePartitioner(N, (1, 1))
    ex = OracleExchange(part, NH)
    u, v = _cube(data, "u", 0, 1), _cube(data, "v", 1, 0)
    u0, v0 = [a.copy() for a in u], [a.copy() for a in v]
    ex.synchronize_vector_interfaces(u, v)
    touched = 0
    for t in range(6):
        m = ex._map("sync_dgrid", t)
        touched += len(m.dst_flat)
        assert np.array_equal(u[t][NH : NH + N, NH : NH + N + 1], u0[t][NH : NH + N, NH : NH + N + 1]), f"tile {t}: u changed"
        assert np.array_equal(v[t][NH : NH + N + 1, NH : NH + N], v0[t][NH : NH + N + 1, NH : NH + N]), f"tile {t}: v changed"
    # every shared edge is rewritten from its owner: 12 edges x 12 faces
    assert touched == 12 * N
    # the winds there are not trivially equal (the no-op is a statement about the map, not about the data)
    assert np.nanstd(u0[0][NH : NH + N, NH + N]) > 1.0

    # negative controls: the same map with (a) the along-edge order of the sources reversed, (b) the signs flipped
    for mode in ("reverse", "sign"):
        ex2 = OracleExchange(part, NH)
        for t in range(6):
            m = ex2._map("sync_dgrid", t)
            if mode == "reverse":
                for sr in np.unique(m.src_rank):
                    for dc in (0, 1):
                        sel = np.nonzero((m.src_rank == sr) & (m.dst_comp == dc))[0]
                        m.src_flat[sel] = m.src_flat[sel][::-1].copy()
            else:
                m.sign[:] = -m.sign
        u2, v2 = [a.copy() for a in u0], [a.copy() for a in v0]
        ex2.synchronize_vector_interfaces(u2, v2)
        changed = sum(int((u2[t] != u0[t])[NH : NH + N, NH : NH + N + 1].sum() + (v2[t] != v0[t])[NH : NH + N + 1, NH : NH + N].sum()) for t in range(6))
        assert changed > 0.8 * touched * NZ, f"control '{mode}' should break the identity ({changed} values changed)"


def _edge_rows(a):
    """(first halo row, first interior row) along the W, E, S, N edges of a[i, j]"""
    s = slice(NH, NH + N)
    return [(a[NH - 1, s], a[NH, s]), (a[NH + N, s], a[NH + N - 1, s]), (a[s, NH - 1], a[s, NH]), (a[s, NH + N], a[s, NH + N - 1])]


def test_scalar_halos_continue_the_real_fields_across_all_twelve_edges(data):
    part = CubedSpherePartitioner(N, (1, 1))
    ex = OracleExchange(part, NH)
    temp, delp = _cube(data, "T"), _cube(data, "delp")
    phis = [np.full((N + 2 * NH + 1, N + 2 * NH + 1, 1), np.nan) for _ in range(6)]
    for t in range(6):
        phis[t][NH : NH + N, NH : NH + N, 0] = data["phis"][t].T
    for f in (temp, delp, phis):
        ex.scalar(f)
        for t in range(6):  # every edge halo cell was filled (the 3 x 3 blocks beyond a cube corner are not part of any update)
            a = f[t]
            assert np.isfinite(a[:NH, NH : NH + N]).all() and np.isfinite(a[NH + N : NH + N + NH, NH : NH + N]).all()
            assert np.isfinite(a[NH : NH + N, :NH]).all() and np.isfinite(a[NH : NH + N, NH + N : NH + N + NH]).all()

    def jumps(fields, levels):
        good, bad = [], []
        for t in range(6):
            for k in levels:
                for h, i in _edge_rows(fields[t][:, :, k]):
                    good.append(np.mean(np.abs(h - i)))
                    bad.append(np.mean(np.abs(h[::-1] - i)))
        return np.array(good), np.array(bad)

    for name, f, levels, frac in (("T", temp, range(5, NZ), 0.75), ("delp", delp, range(40, NZ), 0.6), ("phis", phis, [0], 0.6)):
        good, bad = jumps(f, levels)
        assert good.mean() < frac * bad.mean(), f"{name}: mean cross-edge jump {good.mean():.3g} vs {bad.mean():.3g} with the along-edge order reversed"
    # and the jump is what the field does from one cell to the next anyway (C12 cells are 800 km wide)
    g_int = np.mean([np.abs(temp[t][NH : NH + N - 1, NH : NH + N, 5:] - temp[t][NH + 1 : NH + N, NH : NH + N, 5:]).mean() for t in range(6)])
    good, _ = jumps(temp, range(5, NZ))
    assert good.mean() < 1.5 * g_int


def test_vector_halos_carry_the_right_rotation_and_sign(data):
    part = CubedSpherePartitioner(N, (1, 1))
    ex = OracleExchange(part, NH)
    u, v = _cube(data, "u", 0, 1), _cube(data, "v", 1, 0)
    ex.vector(u, v, "dgrid")
    s = slice(NH, NH + N)
    mid = slice(NH + 3, NH + N - 3)  # edge mid-points: the coordinate lines of the two tiles are (nearly) parallel there
    levels = range(10, 40)           # the jets

    def stats(sign):
        along_j, along_g, cross_j, cross_g = [], [], [], []
        for t in range(6):
            for k in levels:
                a, b = v[t][:, :, k], u[t][:, :, k]
                # along-edge component: v at the x-interfaces beside the W / E edges, u at the y-interfaces beside S / N;
                # the edge interface itself is index NH (W / S) and NH + N (E / N)
                for f, lo in ((a, True), (a, False)):
                    e = NH if lo else NH + N
                    h, i = (e - 1, e + 1) if lo else (e + 1, e - 1)
                    along_j.append(np.abs(sign * f[h, s] - f[e, s]))
                    along_g.append(np.abs(f[i, s] - f[e, s]))
                for f, lo in ((b, True), (b, False)):
                    e = NH if lo else NH + N
                    h, i = (e - 1, e + 1) if lo else (e + 1, e - 1)
                    along_j.append(np.abs(sign * f[s, h] - f[s, e]))
                    along_g.append(np.abs(f[s, i] - f[s, e]))
                # cross-edge component: u in the cells beside the W / E edges, v in the cells beside S / N
                cross_j += [np.abs(sign * b[NH - 1, mid] - b[NH, mid]), np.abs(sign * b[NH + N, mid] - b[NH + N - 1, mid]),
                            np.abs(sign * a[mid, NH - 1] - a[mid, NH]), np.abs(sign * a[mid, NH + N] - a[mid, NH + N - 1])]
                cross_g += [np.abs(b[NH, mid] - b[NH + 1, mid]), np.abs(b[NH + N - 1, mid] - b[NH + N - 2, mid]),
                            np.abs(a[mid, NH] - a[mid, NH + 1]), np.abs(a[mid, NH + N - 1] - a[mid, NH + N - 2])]
        return [float(np.mean(x)) for x in (along_j, along_g, cross_j, cross_g)]

    aj, ag, cj, cg = stats(1.0)
    faj, _, fcj, _ = stats(-1.0)
    # seen from the other tile, "first halo interface minus edge" IS "first interior interface minus edge": the two means agree to
    # round-off if and only if every halo value arrived with the magnitude and sign its owner holds
    assert abs(aj - ag) < 1e-12 * ag, (aj, ag)
    assert faj > 3.0 * aj, f"along-edge component: jump {aj:.2f} m/s, {faj:.2f} with the sign flipped"
    assert cj < 2.0 * cg and fcj > 2.0 * cj, f"cross-edge component: jump {cj:.2f} m/s (interior gradient {cg:.2f}), {fcj:.2f} with the sign flipped"


def _restart_cube(data, layout=(1, 1)):
    c = get_constants()
    part = CubedSpherePartitioner(N, layout)
    cfg = AcousticDynamicsConfig(npx=N + 1, npy=N + 1, npz=NZ, layout=layout, n_split=2)
    grids = [make_grid(part, r, nz=NZ, ak=data["ak"], bk=data["bk"]) for r in range(part.total_ranks)]
    states = [restart_state(g, data, part.tile_index(r), part.origin(r), c) for r, g in enumerate(grids)]
    phis = [s.pop("phis") for s in states]
    return c, part, cfg, grids, states, phis


@pytest.mark.parametrize("layout", [(1, 1), (2, 2)])
def test_real_global_state_one_acoustic_call_matches_the_oracle(backend, data, layout):
    """The six real tiles with their real terrain (not tile 1 replicated): HIP library = oracle after one acoustic call of two
    sub-steps, to the tolerances of the synthetic-state parity tests; on 6 ranks and on 24 (6 x 6 cells each)."""
    from test_parity import TOL, W_ATOL

    c, part, cfg, grids, states, phis = _restart_cube(data, layout)
    ost = [{k: v.copy() for k, v in s.items()} for s in states]
    odyn = OracleAcousticDynamics(part, grids, cfg, c, phis)
    odyn(ost, 60.0, 1)
    got, *_ = run_device_cube(backend, part, cfg, grids, states, phis, 60.0)
    compare_cubes(got, ost, part, NZ, ("delp", "pt", "u", "v", "w", "delz", "q_con"), TOL, atol=W_ATOL)


def _edge_band_roughness(h, name, k):
    """RMS of the 2-dx (checkerboard) component of a cell-centred field at level k: cells within 2 of a tile edge, and the rest"""
    band, inner = [], []
    for i in range(len(h.grids)):
        a = getattr(h.state, name).numpy(i)[NH : NH + N, NH : NH + N, k]
        lap = a[1:-1, 1:-1] - 0.25 * (a[:-2, 1:-1] + a[2:, 1:-1] + a[1:-1, :-2] + a[1:-1, 2:])
        mask = np.zeros_like(lap, dtype=bool)
        mask[:1, :] = mask[-1:, :] = mask[:, :1] = mask[:, -1:] = True  # (cells 2 from the tile edge: their stencil touches cell 1)
        band.append(lap[mask] ** 2)
        inner.append(lap[~mask] ** 2)
    return float(np.sqrt(np.mean(np.concatenate(band)))), float(np.sqrt(np.mean(np.concatenate(inner))))


def test_real_global_state_a_dozen_model_steps_stay_sane(backend, data):
    """12 model steps of [acoustic call (3 sub-steps) + tracer advection + vertical remap] from the real state: finite, inside the
    reference's SafetyChecker bounds [REF driver/pace/driver/driver.py:557-560], global air mass conserved to round-off, and the
    grid-scale roughness of the fields next to the tile edges does not grow against the interior's (a wrong edge / corner
    formula shows up exactly there)."""
    from pace_amd._testing import harness_for
    from pace_amd.harness import DycoreHarness

    h = harness_for(backend)(N, nz=NZ, layout=(1, 1), dt_atmos=450.0, k_split=1, n_split=3, init="restart", init_data=data, ak=data["ak"], bk=data["bk"],
                      n_tracers=1, hord_tr=8, remap=True)
    area = [g.area[NH : NH + N, NH : NH + N] for g in h.grids]

    def air_mass():
        return sum((h.state.delp.numpy(i)[NH : NH + N, NH : NH + N, :NZ].sum(axis=2) * area[i]).sum() for i in range(6))

    m0 = air_mass()
    r0 = {n: _edge_band_roughness(h, n, k) for n, k in (("pt", 40), ("delp", 50), ("w", 40))}
    for _ in range(12):
        h.step()
    h.synchronize()
    assert abs(air_mass() - m0) <= 1e-12 * m0
    s = h.sanity()
    for name, (lo, hi, ok) in s.items():
        assert ok, f"{name} is not finite"
    assert -200.0 < s["u"][0] and s["u"][1] < 200.0 and -200.0 < s["v"][0] and s["v"][1] < 200.0
    assert -1.0 < s["delp"][0] and s["delp"][1] < 4000.0
    assert max(abs(s["w"][0]), abs(s["w"][1])) < 5.0
    for i in range(6):
        temp = h.state.pt.numpy(i)[NH : NH + N, NH : NH + N, :NZ] * h.state.pkz.numpy(i)[NH : NH + N, NH : NH + N, :NZ]
        assert 100.0 < temp.min() and temp.max() < 380.0, (temp.min(), temp.max())
    for (n, k), (b0, i0) in zip((("pt", 40), ("delp", 50)), (r0["pt"], r0["delp"])):
        b1, i1 = _edge_band_roughness(h, n, k)
        assert b1 / i1 < 2.0 * max(b0 / i0, 1.0), f"{n}: 2-dx roughness next to the tile edges grew: band / interior {b0 / i0:.2f} -> {b1 / i1:.2f}"
    bw, iw = _edge_band_roughness(h, "w", 40)
    assert bw < 5.0 * iw + 1e-3, f"w: roughness next to the tile edges {bw:.3g} vs interior {iw:.3g}"
