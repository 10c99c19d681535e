"""Test-only entry points.  The product surface -- ``StencilFactory`` / ``QuantityFactory.from_backend`` / ``DycoreHarness`` -- accepts
exactly one backend, ``"hip:gfx950"``.  The host emulation (the same kernel sources compiled with g++ under ``tests/_hostemu``: how the
kernel logic is checked against the oracle in a GPU-less container) is reachable only through this module, so that no caller of the
package can select a CPU path by passing a string.  The product loader refuses the emulation library in any case (``pace_amd/lib.py``)."""
from __future__ import annotations

_TOKEN = object()


def hostemu_factory(grids, config, constants=None, dtype=None, **kw):
    """A ``StencilFactory`` on the host-emulation library (tests only)."""
    import torch

    from .context import StencilFactory

    return StencilFactory(grids, config, constants, backend="hostemu", dtype=dtype or torch.float64, _testing_token=_TOKEN, **kw)


def hostemu_quantity_factory(sizer, dtype=None):
    import torch

    from .quantity import QuantityFactory

    return QuantityFactory(sizer, "cpu", dtype or torch.float64)


def hostemu_harness(*args, **kw):
    """A ``DycoreHarness`` on the host-emulation library (tests only)."""
    from .harness import DycoreHarness

    return DycoreHarness(*args, backend="hostemu", _testing_token=_TOKEN, **kw)


def stencil_factory_for(backend):
    """``backend`` of a parametrized test -> constructor with the ``StencilFactory(grids, config, constants, **kw)`` signature."""
    if backend == "hostemu":
        return hostemu_factory
    from .context import StencilFactory

    def make(grids, config, constants=None, **kw):
        return StencilFactory(grids, config, constants, backend=backend, **kw)

    return make


def harness_for(backend):
    """``backend`` of a parametrized test -> constructor with the ``DycoreHarness(...)`` signature (without ``backend``)."""
    if backend == "hostemu":
        return hostemu_harness
    from .harness import DycoreHarness

    def make(*args, **kw):
        return DycoreHarness(*args, backend=backend, **kw)

    return make
