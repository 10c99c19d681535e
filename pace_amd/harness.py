"""Dycore-only harness: builds grid, synthetic state and the acoustic dynamics for the ranks
one process owns and steps them -- the slice of ``Driver._critical_path_step_all`` that reaches
the hot path with ``dycore_only: true, disable_step_physics: true``
[REF driver/pace/driver/driver.py:627-662; .jenkins/driver_configs/baroclinic_c48_6ranks_dycore_only.yaml:1-2].
One "step" = ``k_split`` calls of AcousticDynamics, timed like the reference ("mainloop" timer, first step dropped
[REF .jenkins/print_performance_number.py:13-14]).  The headline workload is the acoustic loop alone; ``n_tracers`` adds the
sub-cycled tracer advection after every acoustic call and ``remap`` the Lagrangian-to-Eulerian remap after that (SURVEY §8f-3:
together the body of ``DynamicalCore.step_dynamics``; no physics, no moist thermodynamics).
"""
from __future__ import annotations

import time
from typing import Optional

import numpy as np
import torch

from .config import AcousticDynamicsConfig
from .constants import get_constants
from .context import StencilFactory
from .dyn_core import STATE_NAMES, AcousticDynamics, DycoreState
from .grid import make_grid
from .halo import Layout
from .init import synthetic_state, synthetic_state_device
from .topology import CubedSpherePartitioner

# dt_atmos / k_split / n_split per configuration
CONFIGS = {
    "c12": dict(nx_tile=12, nz=79, layout=(1, 1), dt_atmos=225.0, k_split=1, n_split=1),
    "c48": dict(nx_tile=48, nz=79, layout=(1, 1), dt_atmos=225.0, k_split=1, n_split=1),
    "c192": dict(nx_tile=192, nz=79, layout=(1, 1), dt_atmos=200.0, k_split=7, n_split=8),
    # not a reference config: 6 sub-domains of 384^2 = the per-GPU share of the C768 run on 4 GPUs (scaling studies on one GPU)
    "c384": dict(nx_tile=384, nz=79, layout=(1, 1), dt_atmos=225.0, k_split=2, n_split=6),
    # likewise 6 sub-domains of 272^2 ~ 3 of 384^2 = the per-GPU cell count of the C768 run on 8 GPUs
    "c272": dict(nx_tile=272, nz=79, layout=(1, 1), dt_atmos=225.0, k_split=2, n_split=6),
    "c768": dict(nx_tile=768, nz=79, layout=(2, 2), dt_atmos=225.0, k_split=2, n_split=6),
}


class DycoreHarness:
    def __init__(
        self,
        nx_tile: int,
        nz: int = 79,
        layout=(1, 1),
        dt_atmos: float = 225.0,
        k_split: int = 1,
        n_split: int = 1,
        world_size: int = 1,
        proc: int = 0,
        backend: str = "hip:gfx950",
        device: Optional[str] = None,
        dtype=torch.float64,
        group=None,
        seed: int = 20261002,
        noise: float = 0.01,
        verbose: bool = False,
        config_overrides: Optional[dict] = None,
        init: str = "synthetic",
        n_tracers: int = 0,
        hord_tr: int = 8,
        remap: bool = False,
        init_data=None,
        ak=None,
        bk=None,
        loopback: bool = False,
        _testing_token=None,
    ):
        self.c = get_constants()
        self.part = CubedSpherePartitioner(nx_tile, tuple(layout))
        self.cfg = AcousticDynamicsConfig(npx=nx_tile + 1, npy=nx_tile + 1, npz=nz, layout=tuple(layout), dt_atmos=dt_atmos, k_split=k_split, n_split=n_split,
                                          **(config_overrides or {}))
        self.layout = Layout(self.part, world_size, proc)
        self.layout.group = group
        self.layout.loopback = bool(loopback)  # this process plays `proc` of `world_size` alone, its messages looped back (timing runs)
        t0 = time.time()
        self.grids = [make_grid(self.part, r, nz=nz, ak=ak, bk=bk) for r in self.layout.local_ranks]
        if verbose:
            print(f"[harness] grid for ranks {self.layout.local_ranks} in {time.time() - t0:.1f}s", flush=True)
        self.sf = StencilFactory(self.grids, self.cfg, self.c, backend=backend, device=device, dtype=dtype, _testing_token=_testing_token)
        self.state = DycoreState(self.sf.quantity_factory)
        t0 = time.time()
        on_device = not self.sf.hostemu
        if init not in ("synthetic", "baroclinic", "restart"):
            raise ValueError(f"init {init!r}: 'synthetic' (SURVEY §8d recipe), 'baroclinic' (JW2006 wave) or 'restart' (init_data = the six-tile FV3 restart arrays)")
        for i, (g, r) in enumerate(zip(self.grids, self.layout.local_ranks)):
            if init == "baroclinic":
                from .init import baroclinic_state

                s = baroclinic_state(g, self.c)
                for n in STATE_NAMES + ["phis"]:
                    getattr(self.state, n).set_numpy(s[n], i)
            elif init == "restart":
                from .init import restart_state

                s = restart_state(g, init_data, self.part.tile_index(r), self.part.origin(r), self.c)
                for n in STATE_NAMES + ["phis"]:
                    getattr(self.state, n).set_numpy(s[n], i)
            elif on_device:
                # evaluate the recipe with torch on the GPU (numpy needs ~1 min per 384^2 x 79 rank)
                s = synthetic_state_device(g, self.sf.device, seed=seed, rank=r, noise=noise)
                for n in STATE_NAMES + ["phis"]:
                    q = getattr(self.state, n)
                    src = s[n]
                    q.storage[i].copy_((src.permute(1, 0) if src.dim() == 2 else src.permute(2, 1, 0)).to(q.storage.dtype))
            else:
                s = synthetic_state(g, seed=seed, rank=r, noise=noise)
                for n in STATE_NAMES + ["phis"]:
                    getattr(self.state, n).set_numpy(s[n], i)
            del s
        if verbose:
            print(f"[harness] {init} state in {time.time() - t0:.1f}s", flush=True)
        self.dyn = AcousticDynamics(self.layout, self.grids, self.sf, config=self.cfg, phis=self.state.phis, state=self.state)
        # shared D-grid interface winds must be single-valued across sub-domains (they are in any
        # physical state; the per-rank white noise of the synthetic recipe breaks it)
        if not loopback:
            self.dyn._updaters["interface_u__v"].update()
        # tracers advected after every acoustic call with the mass fluxes / Courant numbers it accumulated
        self.tracers = {}
        if n_tracers:
            from .stencils import FiniteVolumeTransport, TracerAdvection

            qf = self.sf.quantity_factory
            for t in range(n_tracers):
                q = qf.zeros(("x", "y", "z"), "kg/kg")
                q.storage.copy_(self.state.q_con.storage * (10.0 * (t + 1)) + 1.0e-3 * (t + 1))
                self.tracers[f"tracer{t}"] = q
            self.dp1 = qf.zeros(("x", "y", "z"), "Pa")
            self.tracer_advection = TracerAdvection(self.sf, qf, FiniteVolumeTransport(self.sf, qf, self.grids, hord=hord_tr), self.grids, self.layout, self.tracers)
            self._tracer_halo = self.dyn.halo.updater("cell", [(q,) for q in self.tracers.values()])
        self.remap = None
        if remap:
            from .stencils import LagrangianToEulerian

            self.remap = LagrangianToEulerian(self.sf, self.sf.quantity_factory, self.grids)
            self.ps = self.sf.quantity_factory.zeros(("x", "y"), "Pa")
        self.cells_local = self.part.nx * self.part.ny * nz * len(self.grids)
        self.cells_global = nx_tile * nx_tile * 6 * nz

    def step(self, timer=None):
        """One model step of the dycore-only driver: k_split acoustic-dynamics calls (each followed by tracer advection and the
        vertical remap where the harness was built with them).  ``timer`` (pace_amd.timer.Timer): the reference's timer names inside
        ``DynamicalCore.step_dynamics`` -- "DynCore" around the acoustic dynamics, "TracerAdvection", "Remapping"
        [REF tests/main/driver/test_driver.py:77-121]."""
        from .timer import NullTimer

        timer = timer or NullTimer()
        dt = self.cfg.dt_atmos / self.cfg.k_split
        for k in range(self.cfg.k_split):
            if self.tracers:
                self.dp1.storage.copy_(self.state.delp.storage)  # the air mass the accumulated mass fluxes start from
            with timer.clock("DynCore"):
                self.dyn(self.state, dt, n_map=k + 1)
            if self.tracers:
                with timer.clock("TracerAdvection"):
                    self._tracer_halo.update()
                    self.tracer_advection(self.tracers, self.dp1, self.state.mfxd, self.state.mfyd, self.state.cxd, self.state.cyd)
            if self.remap is not None:
                s = self.state
                with timer.clock("Remapping"):
                    self.remap(self.tracers, s.pt, s.delp, s.delz, s.peln, s.pe, s.pk, s.pkz, s.u, s.v, s.w, s.cappa, self.ps, self.dyn._wsd)

    def close(self):
        """Destroy the library context now (scratch, streams, the RCCL communicator) instead of at garbage collection -- the end of a
        multi-process run, while the process group still exists."""
        self.sf.close()

    def synchronize(self):
        if not self.sf.hostemu:
            torch.cuda.synchronize(self.sf.device)

    def checksum_parts(self):
        """Per-sub-domain float64 sums of the prognostic fields (local sub-domains, in rank order)."""
        out = {}
        for n in ("delp", "pt", "u", "v", "w", "delz", "q_con"):
            q = getattr(self.state, n)
            out[n] = [float(q.sub(i).view[...][..., : self.cfg.npz].double().sum().item()) for i in range(q.n_sub)]
        return out

    def checksum(self, gather=None):
        """Order-fixed float64 sums of the prognostic fields: the per-sub-domain sums added in global rank order, so the value is
        bitwise comparable between runs of one build whatever the number of processes (`gather`: a callable that returns the
        list of every process's `checksum_parts()` in process order; None = single process)."""
        parts = [self.checksum_parts()] if gather is None else gather(self.checksum_parts())
        out = {}
        for n in parts[0]:
            tot = 0.0
            for p in parts:
                for v in p[n]:
                    tot += v
            out[n] = tot
        return out

    def sanity(self):
        """SafetyChecker-style bounds [REF driver/pace/driver/driver.py:557-560] on the local state."""
        out = {}
        for n in ("delp", "pt", "u", "v", "w"):
            q = getattr(self.state, n)
            v = q.view[...]
            v = v[..., : self.cfg.npz]
            out[n] = (float(v.min()), float(v.max()), bool(torch.isfinite(v).all()))
        return out
