#!/usr/bin/env python3
"""Golden input/output vectors of the acoustic path FROM THE REFERENCE ITSELF (pyFV3, numpy backend).

This is the script that pins parity.  It needs an environment where the reference's
un-vendored dependencies import (`pyFV3`, `ndsl`, `gt4py`): that is neither this build container nor
the GPU box, so the fixtures it writes (`tests/golden/golden_c12/*.npz`) are absent from the tree and
`tests/test_reference_golden.py` skips with "reference parity unpinned" until someone runs

    python tools/gen_golden.py --out tests/golden/golden_c12          (inside a pace checkout)

What it records: EVERY checkpoint call of one `step_dynamics` -- `C_SW-In/Out`, `D_SW-In/Out`, `Tracer2D1L-In/Out`,
`Remapping-In/Out`, `FVDynamics-In/Out` (variable names as in tests/savepoint/thresholds/fv_dynamics.yaml:2-360) -- of the C12 L79
baroclinic case, set up exactly like the reference's tests/main/fv3core/test_dycore_call.py:29-134 (numpy backend, layout (1,1)).
One file per savepoint call: `<savepoint>_call<N>_rank<R>.npz` holding every checkpointed array as (i, j, k) float64, plus
`grid_rank<R>.npz` (the GridData / DampingCoefficients fields the operators read) and `meta.json` (config scalars).  Beside the
variables the reference passes to its checkpointer the recorder stores what a consumer needs to RE-RUN the operator: the tracers at
`Tracer2D1L-*` / `Remapping-*` / `FVDynamics-*` (`tracer_<name>`) and the whole prognostic state at `FVDynamics-In`
(`state_<name>`), read from the dycore state object.  Only data is written -- no reference source.

Single process (default): NullComm, rank `--rank`: the operator-level pairs (C_SW, D_SW, Tracer2D1L with one sub-cycle, Remapping)
do not depend on the halo transport.  `mpirun -n 6 python tools/gen_golden.py` (mpi4py): all six ranks with real halo updates --
what `FVDynamics-In/Out` needs.  `--dry` zeroes the water species of the initial state (this build remaps the dry configuration:
no moist_cv, saturation adjustment or energy fixer -- DESIGN §8), so that Remapping / FVDynamics are comparable as a whole.

Consumers: tests/test_reference_golden.py (C_SW, D_SW), tests/test_reference_golden_dynamics.py + tests/savepoint_checkers.py
(Tracer2D1L, Remapping, FVDynamics), each under the reference's own per-variable thresholds.
"""
import argparse
import json
import os
import sys
from collections import defaultdict

import numpy as np


class Recorder:
    """A pyFV3 checkpointer: called as checkpointer(savepoint_name, **arrays_or_quantities)."""

    WANTED = ("C_SW-In", "C_SW-Out", "D_SW-In", "D_SW-Out", "Tracer2D1L-In", "Tracer2D1L-Out", "Remapping-In", "Remapping-Out", "FVDynamics-In", "FVDynamics-Out")
    # prognostic fields of the dycore state a consumer needs to start a step from (names of pace_amd.dyn_core.STATE_NAMES)
    STATE = "u v w ua va uc vc delp delz pt pe pk peln pkz q_con omga cappa mfxd mfyd cxd cyd diss_estd phis".split()
    TRACERS = "qvapor qliquid qice qrain qsnow qgraupel qo3mr qsgs_tke qcld".split()

    def __init__(self, out, rank, wanted=WANTED):
        self.out, self.rank, self.wanted = out, rank, wanted
        self.calls = defaultdict(int)
        self.state = None  # the DycoreState object (set by main): source of the tracers / the full state

    def __call__(self, savepoint_name, **kwargs):
        if savepoint_name not in self.wanted:
            return
        n = self.calls[savepoint_name]
        self.calls[savepoint_name] += 1
        arrays = {}
        for name, value in kwargs.items():
            data = getattr(value, "data", value)
            arrays[name] = np.array(np.asarray(data), dtype=np.float64)
        if self.state is not None and savepoint_name.split("-")[0] in ("Tracer2D1L", "Remapping", "FVDynamics"):
            for name in self.TRACERS:
                q = getattr(self.state, name, None)
                if q is not None and f"tracer_{name}" not in arrays and name not in arrays:
                    arrays[f"tracer_{name}"] = np.array(np.asarray(getattr(q, "data", q)), dtype=np.float64)
            if savepoint_name == "FVDynamics-In":
                for name in self.STATE:
                    q = getattr(self.state, name, None)
                    if q is not None and name not in arrays:
                        arrays[f"state_{name}"] = np.array(np.asarray(getattr(q, "data", q)), dtype=np.float64)
        np.savez_compressed(os.path.join(self.out, f"{savepoint_name}_call{n}_rank{self.rank}.npz"), **arrays)

    # the reference's checkpointers are also context managers in places
    def trial(self):
        import contextlib

        return contextlib.nullcontext()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="tests/golden/golden_c12")
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--eta-file", default="tests/main/input/eta79.nc")
    ap.add_argument("--dry", action="store_true", help="zero the water species of the initial state (the dry configuration this build remaps)")
    ap.add_argument("--list-alts", action="store_true",
                    help="print the FV3_ALT names (the named alternatives of the uncertain restatements, DESIGN §2) one per line and exit: "
                         "`for a in '' $(python tools/gen_golden.py --list-alts); do FV3_ALT=$a pytest tests/test_reference_golden*.py; done` tries them in one pass")
    a = ap.parse_args()
    if a.list_alts:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        from fv3_oracle.util import ALT_NAMES

        print("\n".join(ALT_NAMES))
        return
    mpi = None
    try:
        from mpi4py import MPI

        if MPI.COMM_WORLD.Get_size() == 6:
            mpi = MPI.COMM_WORLD
            a.rank = mpi.Get_rank()
    except ImportError:
        pass
    try:
        from datetime import timedelta

        import ndsl.dsl.stencil
        import pyFV3
        import pyFV3.initialization.analytic_init as ai
        from ndsl.comm.communicator import CubedSphereCommunicator
        from ndsl.comm.null_comm import NullComm
        from ndsl.comm.partitioner import CubedSpherePartitioner, TilePartitioner
        from ndsl.dsl.dace.dace_config import DaceConfig
        from ndsl.dsl.stencil import GridIndexing
        from ndsl.grid import DampingCoefficients, GridData, MetricTerms
        from ndsl.initialization.allocator import QuantityFactory
        from ndsl.initialization.sizer import SubtileGridSizer
    except ImportError as e:
        raise SystemExit(f"the reference stack is not importable here ({e}); run this inside a pace checkout with its submodules installed")

    os.makedirs(a.out, exist_ok=True)
    backend = "numpy"
    cfg = dict(layout=(1, 1), npx=13, npy=13, npz=79, ntiles=6, nwat=6, dt_atmos=225, a_imp=1.0, beta=0.0, consv_te=False, d2_bg=0.0, d2_bg_k1=0.2,
               d2_bg_k2=0.1, d4_bg=0.15, d_con=1.0, d_ext=0.0, dddmp=0.5, delt_max=0.002, do_sat_adj=True, do_vort_damp=True, fill=True, hord_dp=6,
               hord_mt=6, hord_tm=6, hord_tr=8, hord_vt=6, hydrostatic=False, k_split=1, ke_bg=0.0, kord_mt=9, kord_tm=-9, kord_tr=9, kord_wz=9,
               n_split=1, nord=3, p_fac=0.05, rf_fast=True, rf_cutoff=3000.0, tau=10.0, vtdm4=0.06, z_tracer=True, do_qa=True)
    config = pyFV3.DynamicalCoreConfig(**cfg)
    if mpi is not None:
        from ndsl.comm.mpi import MPIComm

        mpi_comm = MPIComm()
    else:
        mpi_comm = NullComm(rank=a.rank, total_ranks=6, fill_value=0.0)
    partitioner = CubedSpherePartitioner(TilePartitioner(config.layout))
    communicator = CubedSphereCommunicator(mpi_comm, partitioner)
    stencil_config = ndsl.dsl.stencil.StencilConfig(
        compilation_config=ndsl.dsl.stencil.CompilationConfig(backend=backend, rebuild=False, validate_args=True),
        dace_config=DaceConfig(communicator=communicator, backend=backend),
    )
    sizer = SubtileGridSizer.from_tile_params(nx_tile=12, ny_tile=12, nz=79, n_halo=3, extra_dim_lengths={}, layout=config.layout,
                                              tile_partitioner=partitioner.tile, tile_rank=communicator.tile.rank)
    grid_indexing = GridIndexing.from_sizer_and_communicator(sizer=sizer, comm=communicator)
    quantity_factory = QuantityFactory.from_backend(sizer=sizer, backend=backend)
    metric_terms = MetricTerms(quantity_factory=quantity_factory, communicator=communicator, eta_file=a.eta_file)
    grid_data = GridData.new_from_metric_terms(metric_terms)
    damping = DampingCoefficients.new_from_metric_terms(metric_terms)
    state = ai.init_analytic_state(analytic_init_case="baroclinic", grid_data=grid_data, quantity_factory=quantity_factory, adiabatic=config.adiabatic,
                                   hydrostatic=config.hydrostatic, moist_phys=config.moist_phys, comm=communicator)
    stencil_factory = ndsl.dsl.stencil.StencilFactory(config=stencil_config, grid_indexing=grid_indexing)
    if a.dry:
        for name in Recorder.TRACERS:
            q = getattr(state, name, None)
            if q is not None:
                getattr(q, "data", q)[...] = 0.0
    rec = Recorder(a.out, a.rank)
    rec.state = state
    dycore = pyFV3.DynamicalCore(comm=communicator, grid_data=grid_data, stencil_factory=stencil_factory, quantity_factory=quantity_factory,
                                 damping_coefficients=damping, config=config, timestep=timedelta(seconds=config.dt_atmos), phis=state.phis,
                                 state=state, checkpointer=rec)
    dycore.step_dynamics(state)

    # the read-only inputs of the operators (names as in tests/mpi_54rank/test_grid_init.py:33-120)
    names = ("dx dy dxa dya dxc dyc rdx rdy rdxa rdya rdxc rdyc area rarea rarea_c cosa cosa_u cosa_v cosa_s sina_u sina_v rsin_u rsin_v rsina rsin2 "
             "sin_sg1 sin_sg2 sin_sg3 sin_sg4 cos_sg1 cos_sg2 cos_sg3 cos_sg4 fC f0 edge_w edge_e edge_s edge_n ak bk").split()
    grid = {}
    for n in names:
        v = getattr(grid_data, n, None)
        if v is not None:
            grid[n] = np.array(np.asarray(getattr(v, "data", v)), dtype=np.float64)
    for n in ("del6_u", "del6_v", "divg_u", "divg_v", "da_min", "da_min_c"):
        v = getattr(damping, n, None)
        if v is not None:
            grid[n] = np.array(np.asarray(getattr(v, "data", v)), dtype=np.float64)
    np.savez_compressed(os.path.join(a.out, f"grid_rank{a.rank}.npz"), **grid)
    if a.rank == 0 or mpi is None:
        json.dump({"config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}, "ranks": 6 if mpi is not None else [a.rank], "backend": backend,
                   "halo_transport": "mpi" if mpi is not None else "NullComm (halos = 0: operator-level pairs only)", "dry": bool(a.dry),
                   "savepoint_calls": dict(rec.calls)}, open(os.path.join(a.out, "meta.json"), "w"), indent=1)
    print(f"wrote {sum(rec.calls.values())} savepoints + grid to {a.out}")


if __name__ == "__main__":
    main()
