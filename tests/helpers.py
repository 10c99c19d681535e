"""Shared helpers for the parity tests: build a StencilFactory + oracle Dom for a set of ranks,
move fields between oracle ([i, j, k] numpy) and device Quantities, compare with tolerances."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

from pace_amd.config import AcousticDynamicsConfig  # noqa: E402
from pace_amd.constants import get_constants  # noqa: E402
from pace_amd._testing import stencil_factory_for
from pace_amd.context import StencilFactory  # noqa: E402
from pace_amd.grid import make_grid  # noqa: E402
from pace_amd.init import synthetic_state  # noqa: E402
from pace_amd.topology import CubedSpherePartitioner  # noqa: E402

from fv3_oracle.util import Dom  # noqa: E402

DIMS3 = ("x", "y", "z")


class Case:
    """A few ranks of a cubed sphere, held both as oracle inputs and as one batched device context."""

    def __init__(self, nx_tile=12, layout=(1, 1), ranks=(0,), nz=8, backend="hostemu", cfg_kw=None, dtype=torch.float64, seed=7):
        self.c = get_constants()
        self.part = CubedSpherePartitioner(nx_tile, layout)
        kw = dict(npx=nx_tile + 1, npy=nx_tile + 1, npz=nz, layout=layout)
        kw.update(cfg_kw or {})
        self.cfg = AcousticDynamicsConfig(**kw)
        self.ranks = list(ranks)
        self.nz = nz
        self.grids = [make_grid(self.part, r, nz=nz) for r in self.ranks]
        self.doms = [Dom(g, self.c) for g in self.grids]
        self.sf = stencil_factory_for(backend)(self.grids, self.cfg, self.c, dtype=dtype)
        self.qf = self.sf.quantity_factory
        self.states = [synthetic_state(g, seed=seed, rank=r) for g, r in zip(self.grids, self.ranks)]
        self.shape = self.states[0]["u"].shape

    def q(self, arrays=None, dims=DIMS3):
        """Device Quantity from a list of per-rank oracle arrays (or zeros)."""
        if arrays is None:
            return self.qf.zeros(dims)
        return self.qf.from_array([np.asarray(a) for a in arrays], dims)

    def host(self, q, r):
        return q.numpy(r)


def box(D, i0, i1, j0, j1, k=None):
    sl = D.sl(i0, i1, j0, j1)
    return sl if k is None else sl + (k,)


def assert_close(name, got, want, rtol=1e-13, atol_scale=1e-13):
    got = np.asarray(got)
    want = np.asarray(want)
    scale = np.max(np.abs(want)) if want.size else 0.0
    err = np.max(np.abs(got - want)) if want.size else 0.0
    assert np.all(np.isfinite(got)), f"{name}: non-finite values"
    assert err <= atol_scale * scale + rtol * scale + 1e-300, f"{name}: max abs err {err:.3e} vs field scale {scale:.3e}"
    return err / scale if scale > 0 else 0.0


# ---------------------------------------------------------------------------------------------
# whole-cube helpers
# ---------------------------------------------------------------------------------------------
STAG_X = ("v", "uc", "mfxd", "cxd")  # fields with an extra x interface
STAG_Y = ("u", "vc", "mfyd", "cyd")
IFACE_K = ("pe", "pk", "peln")


def compute_slice(name, nx, ny, nz, nh=3):
    ex = 1 if name in STAG_X else 0
    ey = 1 if name in STAG_Y else 0
    kk = nz + 1 if name in IFACE_K else nz
    return (slice(nh, nh + nx + ex), slice(nh, nh + ny + ey), slice(0, kk))


def oracle_cube(nx_tile, layout, nz, cfg_kw=None, seed=7, noise=0.01):
    """(part, cfg, grids, states, OracleAcousticDynamics) with interface-consistent initial winds."""
    from fv3_oracle.dyn_core import OracleAcousticDynamics

    c = get_constants()
    part = CubedSpherePartitioner(nx_tile, layout)
    kw = dict(npx=nx_tile + 1, npy=nx_tile + 1, npz=nz, layout=layout)
    kw.update(cfg_kw or {})
    cfg = AcousticDynamicsConfig(**kw)
    grids = [make_grid(part, r, nz=nz) for r in range(part.total_ranks)]
    states = [synthetic_state(g, seed=seed, rank=r, noise=noise) for r, g in enumerate(grids)]
    phis = [s["phis"] for s in states]
    ost = [{k: v for k, v in s.items() if k != "phis"} for s in states]
    dyn = OracleAcousticDynamics(part, grids, cfg, c, phis)
    dyn.ex.synchronize_vector_interfaces([s["u"] for s in ost], [s["v"] for s in ost])
    return part, cfg, grids, ost, phis, dyn


def run_device_cube(backend, part, cfg, grids, ost_init, phis, timestep, n_calls=1, dtype=torch.float64, native=True, constants=None):
    """Run AcousticDynamics on every rank of the cube in one context; returns per-rank arrays."""
    from pace_amd.dyn_core import AcousticDynamics, DycoreState
    from pace_amd.halo import Layout

    sf = stencil_factory_for(backend)(grids, cfg, constants or get_constants(), dtype=dtype)
    per_rank = [dict(s, phis=p) for s, p in zip(ost_init, phis)]
    st = DycoreState.from_arrays(sf.quantity_factory, per_rank)
    dyn = AcousticDynamics(Layout(part, 1, 0), grids, sf, config=cfg, phis=st.phis, state=st)
    dyn.native = native
    for n in range(n_calls):
        dyn(st, timestep, n_map=n + 1)
    if backend != "hostemu":
        torch.cuda.synchronize()
    return st.to_arrays(), dyn, st, sf


def compare_cubes(got, want, part, nz, names, tol, atol=None):
    """Field-scale relative comparison: max|a - b| <= tol[name] * max|b| (+ atol[name], an absolute floor in the field's own
    unit for fields whose scale can be arbitrarily small -- w of a balanced state)."""
    atol = atol or {}
    worst = {}
    for r in range(part.total_ranks):
        for name in names:
            sl = compute_slice(name, part.nx, part.ny, nz)
            a, b = got[r][name][sl], want[r][name][sl]
            assert np.all(np.isfinite(a)), f"{name} rank {r}: non-finite"
            sc = np.abs(b).max()
            e = np.abs(a - b).max()
            e = max(0.0, e - atol.get(name, 0.0))
            worst[name] = max(worst.get(name, 0.0), e / sc if sc > 0 else e)
    bad = {k: v for k, v in worst.items() if v > tol.get(k, tol["default"])}
    assert not bad, f"field-scale relative errors above tolerance: {bad} (all: {worst})"
    return worst
