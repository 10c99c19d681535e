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

    from pace_amd.harness import DycoreHarness

    torch.set_num_threads(1)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    h = DycoreHarness(nx_tile, nz, layout, dt_atmos=225.0, k_split=2, n_split=2, world_size=world, proc=rank, backend="hostemu", group=None)
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
def test_two_process_gloo_matches_single_process(hostemu, tmp_path, nx_tile, layout, native, monkeypatch):
    nz = 5
    monkeypatch.setenv("FV3_HALO_NATIVE", native)
    sys.path.insert(0, ROOT)
    from pace_amd.harness import DycoreHarness

    h = DycoreHarness(nx_tile, nz, layout, dt_atmos=225.0, k_split=2, n_split=2, world_size=1, proc=0, backend="hostemu")
    h.step()
    ref = h.state.to_arrays(["delp", "pt", "u", "v", "w", "delz"])
    init_file = str(tmp_path / "init")
    mp.spawn(_worker, args=(2, init_file, str(tmp_path), nx_tile, layout, nz, native), nprocs=2, join=True)
    per = len(ref) // 2
    for p in range(2):
        got = np.load(tmp_path / f"proc{p}.npz")
        for i in range(per):
            for n in ("delp", "pt", "u", "v", "w", "delz"):
                a, b = got[f"{n}_{i}"], ref[p * per + i][n]
                sl = (slice(3, -4), slice(3, -4), slice(0, nz))
                assert np.array_equal(a[sl], b[sl]), (p, i, n, np.abs(a[sl] - b[sl]).max())
