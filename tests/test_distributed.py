"""The N > 1 path on CPU: two gloo processes, each owning half of the cube's sub-domains
(hostemu kernels), must reproduce the single-process result bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, init_file, out_dir, nx_tile, layout, nz, native):
    sys.path.insert(0, ROOT)
    os.environ["FV3_HALO_NATIVE"] = native
    import torch.distributed as dist

    from pace_amd._testing import hostemu_harness

    torch.set_num_threads(1)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    h = hostemu_harness(nx_tile, nz, layout, dt_atmos=225.0, k_split=2, n_split=2, world_size=world, proc=rank, group=None)
    h.step()
    out = h.state.to_arrays(["delp", "pt", "u", "v", "w", "delz"])
    np.savez(os.path.join(out_dir, f"proc{rank}.npz"), **{f"{n}_{i}": a[n] for i, a in enumerate(out) for n in a})
    dist.barrier()
    dist.destroy_process_group()


# native = "1": the updaters are fv3_halo_plans driven by fv3_acoustic_step itself (no halo callback), the messages move
# through the host-driven transport (gloo) -- the plans, buffers and start / wait protocol the RCCL transport uses;
# native = "0": the torch.distributed reference path (Python callback per update)
@pytest.mark.parametrize("native", ["1", "0"])
@pytest.mark.parametrize("nx_tile, layout", [(12, (1, 1)), (12, (2, 2))])
def test_two_process_gloo_matches_single_process(hostemu, tmp_path, nx_tile, layout, native, monkeypatch, world=2):
    nz = 5
    monkeypatch.setenv("FV3_HALO_NATIVE", native)
    sys.path.insert(0, ROOT)
    from pace_amd._testing import hostemu_harness

    h = hostemu_harness(nx_tile, nz, layout, dt_atmos=225.0, k_split=2, n_split=2, world_size=1, proc=0)
    h.step()
    ref = h.state.to_arrays(["delp", "pt", "u", "v", "w", "delz"])
    init_file = str(tmp_path / "init")
    mp.spawn(_worker, args=(world, init_file, str(tmp_path), nx_tile, layout, nz, native), nprocs=world, join=True)
    per = len(ref) // world
    for p in range(world):
        got = np.load(tmp_path / f"proc{p}.npz")
        for i in range(per):
            for n in ("delp", "pt", "u", "v", "w", "delz"):
                a, b = got[f"{n}_{i}"], ref[p * per + i][n]
                sl = (slice(3, -4), slice(3, -4), slice(0, nz))
                assert np.array_equal(a[sl], b[sl]), (p, i, n, np.abs(a[sl] - b[sl]).max())



def test_eight_process_headline_partition_matches_single_process(hostemu, tmp_path, monkeypatch):
    """The partition of the headline configuration (layout 2 x 2 = 24 sub-domains over 8 processes, 3 per process, `3g .. 3g+2`
    -- the four sub-tiles of a cube tile straddle two processes, every process has on-device neighbours AND neighbours in two or more other
    processes) on a C24 cube (12^2 sub-domains): the native halo plans over the host-driven transport must reproduce the one-process
    result bit for bit, WHOLE FIELDS compared (compute domain of every sub-domain)."""
    test_two_process_gloo_matches_single_process(hostemu, tmp_path, 24, (2, 2), "1", monkeypatch, world=8)


def _tracer_run(h, scale):
    """one acoustic call, then the tracer advection on Courant numbers scaled up so that it sub-cycles"""
    dt = h.cfg.dt_atmos / h.cfg.k_split
    h.dp1.storage.copy_(h.state.delp.storage)
    h.dyn(h.state, dt, n_map=1)
    if scale is None:  # the reference run picks the scale: 2.5 / (largest accumulated Courant number)
        cm = max(float(h.state.cxd.storage.abs().max()), float(h.state.cyd.storage.abs().max()))
        scale = 2.5 / cm
    h.state.cxd.storage.mul_(scale)
    h.state.cyd.storage.mul_(scale)
    h._tracer_halo.update()
    h.tracer_advection(h.tracers, h.dp1, h.state.mfxd, h.state.mfyd, h.state.cxd, h.state.cyd)
    return scale


def _tracer_worker(rank, world, init_file, out_dir, nx_tile, layout, nz, scale):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from pace_amd._testing import hostemu_harness

    torch.set_num_threads(1)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    h = hostemu_harness(nx_tile, nz, layout, dt_atmos=225.0, k_split=1, n_split=2, world_size=world, proc=rank, group=None, n_tracers=2, hord_tr=8)
    _tracer_run(h, scale)
    assert h.tracer_advection.n_split >= 2, h.tracer_advection.n_split
    # the acoustic dynamics and the tracer advection share the context's one exchanger (a second one would re-initialise the transport)
    assert h.tracer_advection._halo is h.dyn.halo
    np.savez(os.path.join(out_dir, f"proc{rank}.npz"), **{f"{n}_{i}": q.numpy(i) for n, q in h.tracers.items() for i in range(len(h.grids))})
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_tracer_advection_with_sub_cycles_matches_single_process(hostemu, tmp_path):
    """Acoustic call + sub-cycled tracer advection (halo updates of the tracers between the sub-cycles, the Courant bound
    all-reduced) on two gloo processes = the single-process result, bit for bit: the tracers' halo plan runs through the same
    exchanger -- and the same host transport -- as the acoustic plans."""
    nx_tile, layout, nz = 12, (2, 2), 4
    sys.path.insert(0, ROOT)
    from pace_amd._testing import hostemu_harness

    h = hostemu_harness(nx_tile, nz, layout, dt_atmos=225.0, k_split=1, n_split=2, world_size=1, proc=0, n_tracers=2, hord_tr=8)
    scale = _tracer_run(h, None)
    assert h.tracer_advection.n_split >= 2
    ref = {n: [q.numpy(i) for i in range(len(h.grids))] for n, q in h.tracers.items()}
    init_file = str(tmp_path / "init")
    mp.spawn(_tracer_worker, args=(2, init_file, str(tmp_path), nx_tile, layout, nz, scale), nprocs=2, join=True)
    per = len(h.grids) // 2
    for p in range(2):
        got = np.load(tmp_path / f"proc{p}.npz")
        for i in range(per):
            for n in ref:
                a, b = got[f"{n}_{i}"], ref[n][p * per + i]
                sl = (slice(3, -4), slice(3, -4), slice(0, nz))
                assert np.array_equal(a[sl], b[sl]), (p, i, n, np.abs(a[sl] - b[sl]).max())


def test_loopback_share_runs_alone(hostemu):
    """bench.py --emulate-share: one process playing 1 of 8 alone (3 of the 24 sub-domains, its messages looped back) runs the
    multi-process plans and stays finite."""
    sys.path.insert(0, ROOT)
    from pace_amd._testing import hostemu_harness

    h = hostemu_harness(12, nz=4, layout=(2, 2), dt_atmos=60.0, k_split=1, n_split=2, world_size=8, proc=0, loopback=True)
    assert len(h.grids) == 3 and h.dyn.halo.transport_name.startswith("loopback")
    h.step()
    assert all(ok for _, _, ok in h.sanity().values())


def test_halo_exchanger_reports_a_released_factory_and_a_foreign_group(hostemu):
    """An updater kept beyond its StencilFactory gets a clear error (not a ReferenceError inside update()); a second user of
    the shared exchanger with another process group is refused (one transport per context)."""
    import gc

    from pace_amd.halo import HaloExchanger, Layout
    from pace_amd._testing import hostemu_harness

    h = hostemu_harness(12, nz=4, layout=(1, 1))
    ex = h.dyn.halo
    assert HaloExchanger.shared(h.sf, h.layout) is ex
    with pytest.raises(ValueError, match="process group"):
        HaloExchanger.shared(h.sf, h.layout, group=object())
    assert ex.sf is h.sf
    import weakref

    class Gone:
        pass

    g = Gone()
    ex._sf_ref = weakref.ref(g)  # what the exchanger sees once its factory has been collected
    del g
    gc.collect()
    with pytest.raises(RuntimeError, match="has been released"):
        ex.sf
