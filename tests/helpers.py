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
        self.sf = StencilFactory(self.grids, self.cfg, self.c, backend=backend, dtype=dtype)
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
