import numpy as np
import pytest

from pace_amd.constants import get_constants
from pace_amd.grid import global_area_minima, make_grid
from pace_amd.topology import CubedSpherePartitioner


def test_areas_tile_the_sphere():
    c = get_constants()
    n = 12
    part = CubedSpherePartitioner(n, (1, 1))
    g = make_grid(part, 0, nz=8)
    assert abs(6 * g.area[3 : 3 + n, 3 : 3 + n].sum() / (4 * np.pi * c.RADIUS**2) - 1) < 1e-12
    ac = g.area_c[3 : 4 + n, 3 : 4 + n].copy()
    w = np.ones_like(ac)
    w[0, :] = w[-1, :] = w[:, 0] = w[:, -1] = 0.5
    w[0, 0] = w[0, -1] = w[-1, 0] = w[-1, -1] = 1 / 3
    assert abs(6 * (ac * w).sum() / (4 * np.pi * c.RADIUS**2) - 1) < 1e-12


def test_decomposition_independence():
    """Metric terms of a 2x2 rank equal the matching window of the 1x1 tile
    (the reference's grid test [REF tests/main/test_grid_init.py:30-84])."""
    n = 12
    g1 = make_grid(CubedSpherePartitioner(n, (1, 1)), 2, nz=8)
    part = CubedSpherePartitioner(n, (2, 2))
    for sub in range(4):
        r = 2 * 4 + sub
        g = make_grid(part, r, nz=8)
        x0, y0 = part.origin(r)
        for name in ("dx", "dy", "area", "dxa", "dya", "rarea_c", "sin_sg1", "sin_sg4", "cosa_s", "rsin2", "fC", "f0", "dxc", "dyc"):
            a = g.fields[name][3:9, 3:9]
            b = g1.fields[name][3 + x0 : 9 + x0, 3 + y0 : 9 + y0]
            assert np.allclose(a, b, rtol=1e-12, atol=0), name


@pytest.mark.parametrize("n", [12, 48])
def test_corner_patch_minima(n):
    c = get_constants()
    fast = global_area_minima(n, c)
    g = make_grid(CubedSpherePartitioner(n, (1, 1)), 0, nz=1, ak=np.array([1.0, 0.0]), bk=np.array([0.0, 1.0]), da_min=1.0, da_min_c=1.0)
    assert np.isclose(fast[0], g.area[3 : 3 + n, 3 : 3 + n].min(), rtol=1e-10)
    assert np.isclose(fast[1], g.area_c[3 : 4 + n, 3 : 4 + n].min(), rtol=1e-10)


def test_eta79_fixture():
    from pace_amd.grid import load_eta79

    ak, bk = load_eta79()
    assert ak.shape == (80,) and bk.shape == (80,)
    assert ak[0] == 300.0 and bk[-1] == 1.0 and bk[0] == 0.0
    assert np.all(np.diff(ak + bk * 1e5) > 0)


@pytest.mark.parametrize("rank", [0, 1, 4])
def test_grid_init_not_decomposition_dependent_3x3(rank):
    """The reference's own grid test [REF tests/main/test_grid_init.py:30-84]: C48, ranks 0 / 1 / 4 of a 3 x 3 layout against the
    matching window of the 1 x 1 tile, the same list of metric terms (corner and cell-centre positions, area, dx, dy, the four
    cos_sg / sin_sg, rarea, rdx, rdy) -- and, like the reference (``v1 == v2``), bit for bit on the compute domain."""
    n, lay = 48, (3, 3)
    g1 = make_grid(CubedSpherePartitioner(n, (1, 1)), 0, nz=4)
    part = CubedSpherePartitioner(n, lay)
    g = make_grid(part, rank, nz=4)
    x0, y0 = part.origin(rank)
    m = n // 3
    cells = ("area", "rarea", "lon_agrid", "lat_agrid", "cos_sg1", "cos_sg2", "cos_sg3", "cos_sg4", "sin_sg1", "sin_sg2", "sin_sg3", "sin_sg4")
    for name in cells + ("dx", "rdx", "dy", "rdy", "lon", "lat"):
        a_all = g.fields[name] if name in g.fields else getattr(g, name)
        b_all = g1.fields[name] if name in g1.fields else getattr(g1, name)
        ex = 1 if name in ("dy", "rdy", "lon", "lat") else 0  # x-interface / corner staggering
        ey = 1 if name in ("dx", "rdx", "lon", "lat") else 0
        a = a_all[3 : 3 + m + ex, 3 : 3 + m + ey]
        b = b_all[3 + x0 : 3 + x0 + m + ex, 3 + y0 : 3 + y0 + m + ey]
        assert a.shape == b.shape and np.array_equal(a, b), name
