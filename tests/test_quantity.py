"""GridIndexing / Quantity semantics restated from the reference's own middleware tests
[REF tests/main/fv3core/test_grid.py:56-804]; gtscript.I[0]+a is the plain offset a from the
call origin, gtscript.I[-1]-a is (call_domain - 1 - a)."""
import numpy as np
import pytest
import torch

from pace_amd.constants import X_DIM, X_INTERFACE_DIM, Y_DIM, Y_INTERFACE_DIM, Z_DIM, Z_INTERFACE_DIM
from pace_amd.quantity import GridIndexing, GridSizer, Quantity, QuantityFactory

EDGES = [(True, True, True, True), (False, False, False, False), (True, False, False, True)]


def _grid(domain, n_halo, s, n, w, e):
    return GridIndexing(domain=domain, n_halo=n_halo, south_edge=s, north_edge=n, west_edge=w, east_edge=e)


@pytest.mark.parametrize("s", [True, False])
@pytest.mark.parametrize("n", [True, False])
@pytest.mark.parametrize("w", [True, False])
@pytest.mark.parametrize("e", [True, False])
@pytest.mark.parametrize("o_off, d_off, i_s, i_e, j_s, j_e", [((0, 0), (0, 0), 0, 0, 0, 0), ((-1, -1), (2, 2), 1, 1, 1, 1), ((-1, 0), (2, 0), 1, 1, 0, 0)])
def test_axis_offsets(s, n, w, e, o_off, d_off, i_s, i_e, j_s, j_e):
    domain, h = (4, 4, 4), 3
    g = _grid(domain, h, s, n, w, e)
    origin = (h + o_off[0], h + o_off[1], 0)
    call_domain = (domain[0] + d_off[0], domain[1] + d_off[1], domain[2])
    ao = g.axis_offsets(origin, call_domain)
    big = np.iinfo(np.int16).max
    assert ao["i_start"] == (i_s if w else -big)
    assert ao["i_end"] == ((call_domain[0] - 1 - i_e) if e else big)
    assert ao["j_start"] == (j_s if s else -big)
    assert ao["j_end"] == ((call_domain[1] - 1 - j_e) if n else big)


@pytest.mark.parametrize("add", [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, 0, 1)])
def test_origins_and_domains(add):
    g = _grid((4, 4, 4), 3, True, True, True, True)
    assert g.origin_full(add) == add
    assert g.origin_compute(add) == (3 + add[0], 3 + add[1], add[2])
    assert g.domain_full(add) == (10 + add[0], 10 + add[1], 4 + add[2])
    assert g.domain_compute(add) == (4 + add[0], 4 + add[1], 4 + add[2])


@pytest.mark.parametrize("edges", EDGES)
@pytest.mark.parametrize(
    "domain, dims, halos, o_exp, d_exp",
    [
        ((4, 4, 7), [X_DIM, Y_DIM, Z_DIM], (0, 0, 0), (3, 3, 0), (4, 4, 7)),
        ((4, 4, 7), [X_DIM, Y_DIM, Z_DIM], tuple(), (3, 3, 0), (4, 4, 7)),
        ((4, 4, 7), [X_DIM, Y_DIM, Z_DIM], (3, 3), (0, 0, 0), (10, 10, 7)),
        ((4, 4, 7), [X_INTERFACE_DIM, Y_DIM, Z_DIM], (0, 3), (3, 0, 0), (5, 10, 7)),
        ((4, 4, 7), [Y_INTERFACE_DIM, X_DIM, Z_INTERFACE_DIM], (1, 1), (2, 2, 0), (7, 6, 8)),
    ],
)
def test_get_origin_domain(edges, domain, dims, halos, o_exp, d_exp):
    g = _grid(domain, 3, *edges)
    o, d = g.get_origin_domain(dims, halos)
    assert o == o_exp and d == d_exp


@pytest.mark.parametrize(
    "h, domain, dims, halos, exp",
    [
        (3, (5, 6, 7), [X_DIM, Y_DIM, Z_DIM], (0, 0, 0), (8, 9, 7)),
        (3, (5, 6, 7), [X_DIM, Y_DIM, Z_DIM], tuple(), (8, 9, 7)),
        (3, (5, 6, 7), [Z_DIM, Y_DIM, X_DIM], (0, 0, 0), (7, 9, 8)),
        (0, (4, 4, 7), [X_DIM, Y_DIM, Z_DIM], (0, 0, 0), (4, 4, 7)),
        (3, (4, 4, 7), [X_INTERFACE_DIM, Y_DIM, Z_DIM], (0, 0, 0), (8, 7, 7)),
        (3, (4, 4, 7), [X_DIM, Y_DIM, Z_DIM], (3, 3), (10, 10, 7)),
    ],
)
def test_get_shape(h, domain, dims, halos, exp):
    assert _grid(domain, h, True, True, True, True).get_shape(dims, halos) == exp


@pytest.mark.parametrize("edges", EDGES)
@pytest.mark.parametrize("h", [0, 2])
def test_restrict_vertical(edges, h):
    g = _grid((3, 4, 10), h, *edges)
    r = g.restrict_vertical(k_start=2)
    assert r.max_shape == g.max_shape and r.origin[2] == 2 and r.domain[2] == 8
    for k0, nk in ((0, 10), (2, 8)):
        r = g.restrict_vertical(k_start=k0, nk=nk)
        assert (r.origin[2], r.domain[2]) == (k0, nk)
        for k1, nk1 in ((0, 8), (2, 4)):
            r2 = r.restrict_vertical(k_start=k1, nk=nk1)
            assert (r2.origin[2], r2.domain[2]) == (k0 + k1, nk1)
    for k0, nk in ((-2, 10), (2, 10)):
        with pytest.raises(ValueError):
            g.restrict_vertical(k_start=k0, nk=nk)


def test_quantity_views_share_memory_and_are_ijk():
    qf = QuantityFactory(GridSizer(5, 6, 7, 3, n_sub=2), "cpu", torch.float64)
    q = qf.zeros([X_DIM, Y_INTERFACE_DIM, Z_DIM], "m/s")
    assert q.storage.shape == (2, 8, 13, 12)  # [n_sub, nk, nj, ni]: i fastest
    assert q.data.shape == (2, 12, 13, 8)
    assert q.origin == (3, 3, 0) and q.extent == (5, 7, 7) and q.units == "m/s"
    q.view[:] = 1.0
    assert float(q.storage.sum()) == 2 * 5 * 7 * 7
    a = np.arange(12 * 13 * 8, dtype=np.float64).reshape(12, 13, 8)
    q.set_numpy(a, 1)
    assert np.array_equal(q.numpy(1), a) and np.array_equal(q.sub(1).data.numpy(), a)
    f = q.field
    assert (f.shape[0], f.shape[1], f.shape[2]) == (12, 13, 8) and f.stride[0] == 1 and f.n_sub == 2
    q2 = qf.zeros([X_DIM, Y_DIM])
    assert q2.is_2d and q2.field.shape[2] == 1


def test_the_product_surface_has_one_backend():
    """``StencilFactory`` / ``QuantityFactory.from_backend`` / ``DycoreHarness`` take ``"hip:gfx950"`` and nothing else: the host emulation the
    CPU tests run on is reachable through ``pace_amd._testing`` only, never by passing a backend string."""
    from pace_amd.config import AcousticDynamicsConfig
    from pace_amd.context import StencilFactory
    from pace_amd.harness import DycoreHarness
    from pace_amd.quantity import GridSizer, QuantityFactory

    cfg = AcousticDynamicsConfig(npx=13, npy=13, npz=4)
    for bad in ("hostemu", "numpy", "cuda"):
        with pytest.raises(ValueError):
            StencilFactory([], cfg, backend=bad)
        with pytest.raises(ValueError):
            QuantityFactory.from_backend(GridSizer(12, 12, 4, 3, 1), bad)
    with pytest.raises(ValueError):
        DycoreHarness(12, nz=4, backend="hostemu")
