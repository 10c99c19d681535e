"""Operator classes of the acoustic path -- same names, constructor shape and ``__call__``
argument order as the pyFV3 operators the reference constructs
(ctor/call evidence [REF examples/notebooks/functions.py:877-891,935-951],
class names [REF tests/main/fv3core/test_config.py:10-16]).  Each call is one C-ABI entry of
``libfv3_mi355x`` enqueued on the factory's HIP stream; nothing is allocated or compiled at
call time [REF tests/main/fv3core/test_dycore_call.py:193-211].
"""
from __future__ import annotations

from typing import Optional

from .constants import X_DIM, X_INTERFACE_DIM, Y_DIM, Y_INTERFACE_DIM, Z_DIM, Z_INTERFACE_DIM
from .context import StencilFactory
from .quantity import Quantity

_CELL = (X_DIM, Y_DIM, Z_DIM)


def _ref(q: Optional[Quantity]):
    return None if q is None else q.fref


def _as_numpy(x):
    """Host values of a 1-D / 2-D argument given as numpy array, torch tensor or Quantity; None when it cannot be inspected cheaply."""
    import numpy as np

    if isinstance(x, Quantity):
        return None  # (per-sub-domain device field: see _check_metric)
    if hasattr(x, "detach"):
        return x.detach().cpu().numpy()
    try:
        return np.asarray(x, dtype=np.float64)
    except Exception:
        return None


def _check_owned(sf: StencilFactory, op: str, name: str, given, expected, rtol=1.0e-12):
    """The C side owns the level columns (``dp_ref``, ``pfull``, ``ks``) and the metric terms (``rdxc``, ``rdyc``): they are uploaded
    once with the context and the kernels read them there.  The reference passes them at every call; a caller that passes
    DIFFERENT values would silently get the context's -- refuse instead.  ``None`` = "use the context's" (accepted)."""
    import numpy as np

    if given is None:
        return
    if isinstance(given, Quantity):
        own = sf.grid_fields.get(name)
        if own is not None and given.storage.data_ptr() == own.storage.data_ptr():
            return  # the context's own field
        if own is not None and given.storage.shape == own.storage.shape:
            import torch

            if bool(torch.allclose(given.storage.double(), own.storage.double(), rtol=rtol, atol=0.0)):
                return
        raise ValueError(f"{op}: argument {name!r} differs from the metric term the context was created with (the kernels read the context's copy)")
    g = _as_numpy(given)
    if g is None or expected is None:
        return
    e = np.asarray(expected, dtype=np.float64)
    if g.shape != e.shape or not np.allclose(g, e, rtol=rtol, atol=0.0):
        raise ValueError(f"{op}: argument {name!r} differs from the values the context was created with (GridData.{name}; the kernels read the context's copy)")


class _Op:
    def __init__(self, stencil_factory: StencilFactory, quantity_factory=None, grid_data=None, *_, **__):
        self.sf = stencil_factory
        self.qf = quantity_factory or stencil_factory.quantity_factory
        self.grid_data = grid_data


class CGridShallowWaterDynamics(_Op):
    """``c_sw``; returns (delpc, ptc) like the reference."""

    def __init__(self, stencil_factory, quantity_factory=None, grid_data=None, nested=False, grid_type=0, nord=None):
        super().__init__(stencil_factory, quantity_factory, grid_data)
        self.delpc = self.qf.zeros(_CELL, "Pa")
        self.ptc = self.qf.zeros(_CELL, "K")

    def __call__(self, delp, pt, u, v, w, uc, vc, ua, va, ut, vt, divgd, omga, dt2):
        self.sf.call("c_sw", delp.fref, pt.fref, u.fref, v.fref, w.fref, uc.fref, vc.fref, ua.fref, va.fref, ut.fref, vt.fref, divgd.fref, omga.fref, self.delpc.fref, self.ptc.fref, float(dt2))
        return self.delpc, self.ptc


class UpdateGeopotentialHeightOnCGrid(_Op):
    def __call__(self, dp_ref, zs, ut, vt, gz, ws, dt):
        _check_owned(self.sf, "update_dz_c", "dp_ref", dp_ref, self.sf.grids[0].dp_ref)
        self.sf.call("update_dz_c", zs.fref, ut.fref, vt.fref, gz.fref, ws.fref, float(dt))


class RiemannSolverC(_Op):
    def __call__(self, dt2, cappa, ptop, phis, ws, ptc, q_con, delpc, gz, pef, w3):
        self.sf.call("riem_solver_c", float(dt2), cappa.fref, float(ptop), phis.fref, ws.fref, ptc.fref, q_con.fref, delpc.fref, gz.fref, pef.fref, w3.fref)


class PGradC(_Op):
    """The ``p_grad_c`` stencil of dyn_core."""

    def __call__(self, rdxc, rdyc, uc, vc, delpc, pkc, gz, dt2):
        _check_owned(self.sf, "p_grad_c", "rdxc", rdxc, None)
        _check_owned(self.sf, "p_grad_c", "rdyc", rdyc, None)
        self.sf.call("p_grad_c", uc.fref, vc.fref, delpc.fref, pkc.fref, gz.fref, float(dt2))


class FiniteVolumeFluxPrep(_Op):
    def __init__(self, stencil_factory, grid_data=None, grid_type=0):
        super().__init__(stencil_factory, None, grid_data)

    def __call__(self, uc, vc, crx, cry, x_area_flux, y_area_flux, uc_contra, vc_contra, dt):
        self.sf.call("fxadv", uc.fref, vc.fref, crx.fref, cry.fref, x_area_flux.fref, y_area_flux.fref, uc_contra.fref, vc_contra.fref, float(dt))


class FiniteVolumeTransport(_Op):
    """``fv_tp_2d``.  ``nord``/``damp_c`` enable the del-n damping fluxes (scalars here; the
    per-level columns of d_sw are handled inside ``fv3_d_sw``)."""

    def __init__(self, stencil_factory, quantity_factory=None, grid_data=None, damping_coefficients=None, grid_type=0, hord=6, nord=None, damp_c=None):
        super().__init__(stencil_factory, quantity_factory, grid_data)
        self.hord = int(hord)
        self.nord = -1 if nord is None else int(nord)
        self.damp_c = 0.0 if damp_c is None else float(damp_c)

    def __call__(self, q, crx, cry, x_area_flux, y_area_flux, q_x_flux, q_y_flux, x_mass_flux=None, y_mass_flux=None, mass=None):
        self.sf.call("fv_tp_2d", q.fref, crx.fref, cry.fref, x_area_flux.fref, y_area_flux.fref, q_x_flux.fref, q_y_flux.fref, _ref(x_mass_flux), _ref(y_mass_flux), _ref(mass), self.hord, self.nord, self.damp_c)


class AGrid2BGridFourthOrder(_Op):
    def __init__(self, stencil_factory, quantity_factory=None, grid_data=None, grid_type=0, z_dim=Z_DIM, replace=False):
        super().__init__(stencil_factory, quantity_factory, grid_data)
        self.replace = bool(replace)
        self.nk = stencil_factory.sizer.nz + (1 if z_dim == Z_INTERFACE_DIM else 0)

    def __call__(self, qin, qout, kstart=0, nk=None):
        self.sf.call("a2b_ord4", qin.fref, qout.fref, int(kstart), int(self.nk if nk is None else nk), int(self.replace))


class DGridShallowWaterLagrangianDynamics(_Op):
    def __init__(self, stencil_factory, quantity_factory=None, grid_data=None, damping_coefficients=None, column_namelist=None, nested=False, stretched_grid=False, config=None):
        super().__init__(stencil_factory, quantity_factory, grid_data)

    def __call__(self, delpc, delp, pt, u, v, w, uc, vc, ua, va, divgd, mfx, mfy, cx, cy, crx, cry, xfx, yfx, q_con, zh, heat_source, diss_est, dt):
        a = (delpc, delp, pt, u, v, w, uc, vc, ua, va, divgd, mfx, mfy, cx, cy, crx, cry, xfx, yfx, q_con, zh, heat_source, diss_est)
        self.sf.call("d_sw", *[x.fref for x in a], float(dt))


class UpdateHeightOnDGrid(_Op):
    def __call__(self, surface_height, height, courant_number_x, courant_number_y, x_area_flux, y_area_flux, ws, dt):
        self.sf.call("update_dz_d", surface_height.fref, height.fref, courant_number_x.fref, courant_number_y.fref, x_area_flux.fref, y_area_flux.fref, ws.fref, float(dt))


class RiemannSolver3(_Op):
    def __call__(self, last_call, dt, cappa, ptop, zs, wsd, delz, q_con, delp, pt, zh, pe, ppe, pk3, pk, peln, w):
        self.sf.call("riem_solver3", int(bool(last_call)), float(dt), cappa.fref, float(ptop), zs.fref, wsd.fref, delz.fref, q_con.fref, delp.fref, pt.fref, zh.fref, pe.fref, ppe.fref, pk3.fref, pk.fref, peln.fref, w.fref)


class PK3Halo(_Op):
    def __call__(self, pk3, delp, ptop, akap):
        self.sf.call("pk3_halo", pk3.fref, delp.fref, float(ptop), float(akap))


class EdgePE(_Op):
    def __call__(self, pe, delp, ptop):
        self.sf.call("edge_pe", pe.fref, delp.fref, float(ptop))


class NonHydrostaticPressureGradient(_Op):
    def __call__(self, u, v, pp, gz, pk3, delp, dt, ptop, akap):
        self.sf.call("nh_p_grad", u.fref, v.fref, pp.fref, gz.fref, pk3.fref, delp.fref, float(dt), float(ptop), float(akap))


class RayleighDamping(_Op):
    def __call__(self, u, v, w, dp, pfull, dt, ptop, ks=None):
        _check_owned(self.sf, "ray_fast", "dp_ref", dp, self.sf.grids[0].dp_ref)
        _check_owned(self.sf, "ray_fast", "pfull", pfull, self.sf.grids[0].pfull)
        if ks is not None and int(ks) != int(self.sf.grids[0].ks):
            raise ValueError(f"ray_fast: ks = {ks} differs from the context's ({self.sf.grids[0].ks}: number of pure-pressure layers of ak / bk)")
        self.sf.call("ray_fast", u.fref, v.fref, w.fref, float(dt), float(ptop))


class HyperdiffusionDamping(_Op):
    def __init__(self, stencil_factory, quantity_factory=None, damping_coefficients=None, rarea=None, nmax=3):
        super().__init__(stencil_factory, quantity_factory, None)
        self.nmax = int(nmax)

    def __call__(self, qdel, cd):
        self.sf.call("del2_cubed", qdel.fref, float(cd), self.nmax)


class ApplyDiffusiveHeating(_Op):
    def __call__(self, delp, delz, cappa, heat_source, pt, delt_time_factor):
        self.sf.call("apply_diffusive_heating", delp.fref, delz.fref, cappa.fref, heat_source.fref, pt.fref, float(delt_time_factor))


class TracerAdvection(_Op):
    """``tracer_2d_1l``: sub-cycled 2-D advection of the tracers with the mass fluxes / Courant numbers the
    acoustic sub-steps accumulated.  Constructor and call as the reference's
    ``TracerAdvection(stencil_factory, quantity_factory, transport, grid_data, comm, tracers)`` /
    ``tracer_advection(tracers, dp1, mfxd, mfyd, cxd, cyd)`` [REF examples/notebooks/functions.py:916-951, 1037-1044].
    ``transport`` is a :class:`FiniteVolumeTransport` (its ``hord`` is used: 5 / 6, or 8 = the monotone scheme of the
    reference's dycore configs, ``hord_tr: 8``), ``comm`` a :class:`pace_amd.halo.Layout` (or None: all ranks local)."""

    def __init__(self, stencil_factory, quantity_factory=None, transport=None, grid_data=None, comm=None, tracers=None):
        super().__init__(stencil_factory, quantity_factory, grid_data)
        self.hord = int(getattr(transport, "hord", 6))
        self.comm = comm
        self._halo = None
        self._updater = None
        self._bound = None
        self.n_split = None  # of the last call

    def _tracer_updater(self, tracers):
        from .halo import HaloExchanger, Layout
        from .topology import CubedSpherePartitioner

        key = tuple(id(q) for q in tracers.values())
        if self._bound != key:
            lay = self.comm
            if lay is None:
                cfg = self.sf.config
                lay = Layout(CubedSpherePartitioner(cfg.npx - 1, tuple(cfg.layout)), 1, 0)
            if self._halo is None:
                # the context's one exchanger (the acoustic dynamics' own when it exists): it owns the transport
                self._halo = HaloExchanger.shared(self.sf, lay, group=getattr(lay, "group", None))
            self._updater = self._halo.updater("cell", [(q,) for q in tracers.values()])
            self._bound = key
        return self._updater

    def __call__(self, tracers, dp1, x_mass_flux, y_mass_flux, x_courant, y_courant):
        import ctypes as C

        import torch

        from . import lib as _lib

        sf = self.sf
        cmax = C.c_double()
        st = sf.lib.fv3_tracer_2d_1l_cmax(sf.ctx, x_courant.fref, y_courant.fref, C.byref(cmax), sf.stream_handle)
        if st != 0:
            raise _lib.Fv3Error("fv3_tracer_2d_1l_cmax failed: " + sf.lib.fv3_last_error(sf.ctx).decode())
        cm = cmax.value
        lay = self.comm
        if lay is not None and lay.world_size > 1:  # the operator's one global quantity: all-reduce MAX over the processes
            import torch.distributed as dist

            t = torch.tensor([cm], dtype=torch.float64, device=sf.device if dist.get_backend(getattr(lay, "group", None)) == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=getattr(lay, "group", None))
            cm = float(t.item())
        n_split = int(1.0 + cm)
        self.n_split = n_split
        qs = list(tracers.values())
        arr = (_lib.F * max(len(qs), 1))(*[C.pointer(q.field) for q in qs])
        plan = None
        if n_split > 1:
            up = self._tracer_updater(tracers)
            if not up.ex.native:
                raise _lib.Fv3Error("tracer_2d_1l with sub-cycles needs the native halo plans (FV3_HALO_NATIVE=1)")
            plan = up._native_plan()
        st = sf.lib.fv3_tracer_2d_1l(sf.ctx, len(qs), arr, dp1.fref, x_mass_flux.fref, y_mass_flux.fref, x_courant.fref, y_courant.fref, n_split, self.hord, plan,
                                     sf.stream_handle)
        if st != 0:
            raise _lib.Fv3Error(f"fv3_tracer_2d_1l failed ({st}): " + sf.lib.fv3_last_error(sf.ctx).decode())


class LagrangianToEulerian(_Op):
    """The vertical remap that closes ``DynamicalCore.step_dynamics`` (reference operator pyFV3
    ``LagrangianToEulerian``, savepoint ``Remapping`` [REF tests/savepoint/thresholds/fv_dynamics.yaml:227-326]).  Call with the
    state's quantities; everything is remapped in place (see ``fv3_remap`` in include/fv3_mi355x.h for the configuration)."""

    def __call__(self, tracers, pt, delp, delz, peln, pe, pk, pkz, u, v, w, cappa, ps, wsd):
        import ctypes as C

        from . import lib as _lib

        qs = list(tracers.values()) if tracers else []
        arr = (_lib.F * max(len(qs), 1))(*[C.pointer(q.field) for q in qs])
        st = self.sf.lib.fv3_remap(self.sf.ctx, len(qs), arr, pt.fref, delp.fref, delz.fref, peln.fref, pe.fref, pk.fref, pkz.fref, u.fref, v.fref, w.fref, cappa.fref,
                                   ps.fref, wsd.fref, self.sf.stream_handle)
        if st != 0:
            raise _lib.Fv3Error(f"fv3_remap failed ({st}): " + self.sf.lib.fv3_last_error(self.sf.ctx).decode())
