"""GPU-only checks at sizes the oracle cannot reach quickly: size-independent properties
(global mass conservation, determinism / statelessness, no allocation at call time -- the
reference's dycore-call invariants [REF tests/main/fv3core/test_dycore_call.py:149-211]) and
the fp32 build against fp64."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _harness(nx, layout=(1, 1), nz=79, dtype=torch.float64, **kw):
    from pace_amd.harness import DycoreHarness

    return DycoreHarness(nx, nz, layout, dt_atmos=225.0, k_split=kw.pop("k_split", 1), n_split=kw.pop("n_split", 2), backend="hip:gfx950", device="cuda:0", dtype=dtype, **kw)


def _mass(h):
    nx, nz = h.part.nx, h.cfg.npz
    area = h.sf.grid_fields["area"].storage[:, 3 : 3 + nx, 3 : 3 + nx]
    delp = h.state.delp.storage[:, :nz, 3 : 3 + nx, 3 : 3 + nx]
    return float((delp * area[:, None]).sum(dtype=torch.float64))


def test_c48_mass_conservation_finite_and_bounds(gpu_backend):
    h = _harness(48)
    m0 = _mass(h)
    for _ in range(3):
        h.step()
    h.synchronize()
    m1 = _mass(h)
    assert abs(m1 - m0) / m0 < 1e-13
    s = h.sanity()
    assert all(v[2] for v in s.values()), s
    assert -1 < s["delp"][0] and s["delp"][1] < 4000 and abs(s["u"][0]) < 200 and abs(s["u"][1]) < 200


def test_c96_2x2_decomposition_identity_on_device(gpu_backend):
    """24 sub-domains (layout 2x2) and 6 sub-domains (1x1) give the same fields."""
    outs = []
    for layout in ((1, 1), (2, 2)):
        h = _harness(96, layout, nz=16, noise=0.0)
        h.step()
        h.synchronize()
        nx = h.part.nx
        G = torch.zeros((6, 96, 96, 16), dtype=torch.float64, device="cuda:0")
        for i, r in enumerate(h.layout.local_ranks):
            t = h.part.tile_index(r)
            x0, y0 = h.part.origin(r)
            G[t, x0 : x0 + nx, y0 : y0 + nx] = h.state.pt.sub(i).data[3 : 3 + nx, 3 : 3 + nx, :16]
        outs.append(G)
    assert torch.equal(outs[0], outs[1])


def test_stateless_deterministic_and_no_allocation(gpu_backend):
    h1, h2 = _harness(24, nz=20), _harness(24, nz=20)
    h1.step()
    h2.step()
    h1.synchronize()
    for n in ("delp", "pt", "u", "v", "w", "delz"):
        assert torch.equal(getattr(h1.state, n).storage, getattr(h2.state, n).storage), n
    # second call: no device allocation (scratch is owned by the context, buffers are cached).
    # (objects of earlier tests may be collected DURING the call and lower memory_allocated(): collect them first
    # and compare the allocation COUNT, which only ever grows)
    import gc

    gc.collect()
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    stats0 = torch.cuda.memory_stats()["allocation.all.allocated"]
    h1.step()
    h1.synchronize()
    assert torch.cuda.memory_allocated() <= before
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == stats0


def test_fp32_build_tracks_fp64(gpu_backend):
    """PACE_FLOAT_PRECISION=32 analogue: same step in fp32, stated tolerance 2e-4 field-relative
    (pressure sums and exp/log chains run in fp32 -- SURVEY §7 hard part 9)."""
    # C96: at coarser grids (da_min_c * d4_bg)**(nord+1) overflows fp32, as it would in a
    # 32-bit FV3 build; the fp32 configuration of BASELINE.json is C768
    h64 = _harness(96, nz=20, noise=0.0)
    h32 = _harness(96, nz=20, noise=0.0, dtype=torch.float32)
    h64.step()
    h32.step()
    h64.synchronize()
    for n, tol in (("delp", 2e-5), ("pt", 2e-5), ("u", 2e-4), ("v", 2e-4)):
        a = getattr(h64.state, n).view[...][..., :20]
        b = getattr(h32.state, n).view[...][..., :20].double()
        assert torch.isfinite(b).all()
        err = float((a - b).abs().max() / a.abs().max())
        assert err < tol, (n, err)


def _run(cmd, **kw):
    """subprocess.run(check=True) that SHOWS what the child printed when it fails or times out (a bare CalledProcessError says nothing)."""
    import subprocess

    kw.pop("check", None)
    want_out = kw.pop("capture_output", False)
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, **{k: v for k, v in kw.items() if k != "text"})
    except subprocess.TimeoutExpired as e:
        out = (e.stdout or b"")[-3000:] if isinstance(e.stdout, (bytes, bytearray)) else (e.stdout or "")[-3000:]
        err = (e.stderr or b"")[-3000:] if isinstance(e.stderr, (bytes, bytearray)) else (e.stderr or "")[-3000:]
        raise AssertionError(f"timed out after {e.timeout} s: {cmd}\n--- stdout\n{out}\n--- stderr\n{err}")
    assert r.returncode == 0, f"exit code {r.returncode}: {cmd}\n--- stdout\n{r.stdout[-3000:]}\n--- stderr\n{r.stderr[-3000:]}"
    return r if want_out else r


def _free_port():
    """A rendezvous port nobody holds right now (a fixed port fails when an earlier run's socket is still in TIME_WAIT)."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


@pytest.mark.gpu
@pytest.mark.parametrize("nx, layout, world", [(48, 1, 2), (24, 2, 4), (48, 1, 6), (24, 2, 8)])
def test_two_process_decomposition_is_bitwise_identical(tmp_path, nx, layout, world):
    """The same cube stepped by one process and by `world` processes (C48: 6 sub-domains split 3 + 3;
    C24 layout 2x2: 24 sub-domains, 6 per process, tiles straddling processes as on 4 / 8 GPUs; C48 on 6 processes: ONE
    sub-domain per process, the shape of BASELINE's "6 tiles -> 6 GPUs" configuration; C24 layout 2x2 on 8 processes: the partition of the
    headline run, 3 sub-domains per process, every tile straddling two processes;
    messages over gloo staged through pinned host memory because the box has one GPU; the 8-GPU
    bench uses RCCL with the identical pack / unpack plans) must give bitwise equal FIELDS (tools/multi_gpu_check.py compares a hash of every
    sub-domain's whole compute-domain array besides its sum and maximum)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "multi_gpu_check.py")
    env = dict(os.environ, FV3_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    w1, w2 = str(tmp_path / "w1.json"), str(tmp_path / "w2.json")
    size = ["--nx", str(nx), "--layout", str(layout)]
    _run([sys.executable, tool, "--out", w1] + size, check=True, env=env, timeout=1200)
    _run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
         _free_port(), tool, "--backend", "gloo", "--out", w2] + size,
        check=True, env=env, timeout=1200,
    )
    _run([sys.executable, tool, "--compare", w1, w2], check=True, timeout=60)


@pytest.mark.gpu
@pytest.mark.parametrize(
    "toggle",
    ["FV3_RIEM_MODE=columns", "FV3_EDGE_PROFILE_LDS=1", "FV3_EDGE_PROFILE_REG=1", "FV3_EDGE_PROFILE_GENERIC=1", "FV3_TP2D_MODE=staged", "FV3_DEL6_MODE=staged", "FV3_SEG=96", "FV3_SEG=32", "FV3_CSW_B_GENERIC=1", "FV3_CSW_MARCH=0", "FV3_CSW_MARCH=abc", "FV3_DIVDAMP_STAGED=1", "FV3_KE_STAGED=1", "FV3_DSW_SCALARS=separate", "FV3_AUX_STREAM=0", "FV3_GRID_SPLIT=0", "FV3_PINGPONG=0", "FV3_DSW_WIND_OVERLAP=1", "FV3_TP2D_FA=0", "FV3_RIEM_REGS=0", "FV3_DSW_MARCH=old", "FV3_TP2D_MARCH=old", "FV3_DSW_HEAT=separate", "FV3_CSW_WIN_OVERLAP=0", "FV3_ACC_STORE=0", "FV3_EP_ONE_LAUNCH=0", "FV3_DSW_WINDSTAGE=staged", "FV3_DSW_MARCH=coupled", "FV3_CSW_DEFER=1", "FV3_DEL2_FUSED=0", "FV3_DEL2_HEAT=fused", "FV3_GZ_FIRST=copy", "FV3_DSW_VORT_IN_KE=1", "FV3_DSW_SIDE=copy", "FV3_DSW_SPONGE_WIND=serial", "FV3_FRAME_LAUNCH=split", "FV3_GATHER_BATCH=0", "FV3_DZ_SCAN=separate", "FV3_SEQ_DELZ=every", "FV3_SEQ_UAVA=every", "FV3_ACC_DEFER=0"],
)
def test_alternative_kernel_forms_agree(tmp_path, toggle):
    """Every operator that has two device implementations (wave Riemann solver vs thread-per-column, the
    three edge_profile forms, marching vs staged transport / del-n fluxes; the sponge-layer launches beside the marches on the
    auxiliary stream vs in program order; sub-plane vs whole-plane workgroup mapping of the few-plane launches) must give the same step: per
    sub-domain sums and maxima of the prognostic fields after two C48 L79 steps agree to 1e-11 (the forms
    differ only in FMA association of a few sums; most are bitwise equal).  This is the check that caught a
    mis-compiled edge_profile variant (in-kernel selection among field pointers)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "multi_gpu_check.py")
    base, alt = str(tmp_path / "base.json"), str(tmp_path / "alt.json")
    size = ["--nx", "48", "--nz", "79"]
    env = {k: v for k, v in os.environ.items() if not k.startswith("FV3_")}
    _run([sys.executable, tool, "--out", base] + size, check=True, env=env, timeout=1200)
    name, val = toggle.split("=")
    _run([sys.executable, tool, "--out", alt] + size, check=True, env=dict(env, **{name: val}), timeout=1200)
    _run([sys.executable, tool, "--compare", base, alt, "--rtol", "1e-11"], check=True, timeout=60)


@pytest.mark.gpu
def test_bench_two_ranks_reproduce_the_single_process_state(tmp_path):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one rank per "GPU", barrier + max-over-ranks timing,
    rank 0 prints the line) -- on this 1-GPU box both ranks share GPU 0 and exchange over gloo (test hooks of bench.py; with one
    device per rank the same plans run over RCCL).  The line carries n_gpus = 2 and the SAME state checksum as the one-process
    run: the decomposition over processes does not change a bit of the result."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--config", "c48", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-op-timing"]
    env = {k: v for k, v in os.environ.items() if not k.startswith("FV3_")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    one = _run([sys.executable, os.path.join(root, "bench.py")] + args, check=True, env=env, timeout=1200, capture_output=True, text=True, cwd=str(tmp_path))
    two = _run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", _free_port(),
         os.path.join(root, "bench.py"), "--gpus", "2"] + args,
        check=True, env=dict(env, FV3_FORCE_DEVICE="0", FV3_DIST_BACKEND="gloo"), timeout=1200, capture_output=True, text=True, cwd=str(tmp_path))
    l1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    l2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert l1["n_gpus"] == 1 and l2["n_gpus"] == 2 and l2["finite"] and l2["scaling"] == "strong"
    assert l2["metric"] == l1["metric"] and l2["config"]["workload"].split(",")[0] == l1["config"]["workload"].split(",")[0]
    assert l2["state_checksum"] == l1["state_checksum"], (l1["state_checksum"], l2["state_checksum"])


@pytest.mark.gpu
def test_rccl_calls_execute_on_a_one_rank_communicator(gpu_backend, monkeypatch):
    """What one GPU can execute of the RCCL transport: a communicator of one rank (ncclGetUniqueId / ncclCommInitRank through the
    library's run-time binding) and every message of the 1/8 share's halo updates posted as ncclSend to / ncclRecv from rank 0 in
    the library's group-per-update protocol on its communication stream (FV3_LOOPBACK_TRANSPORT=rccl).  The state after two
    acoustic calls must be bitwise the one of the same run with device copies standing in for the messages."""
    from pace_amd.harness import DycoreHarness

    res = {}
    for mode in ("copy", "rccl"):
        monkeypatch.setenv("FV3_LOOPBACK_TRANSPORT", mode)
        h = DycoreHarness(48, nz=16, layout=(2, 2), dt_atmos=225.0, k_split=2, n_split=3, world_size=8, proc=0, backend=gpu_backend, loopback=True)
        name = h.dyn.halo.transport_name
        assert name.startswith("loopback-rccl" if mode == "rccl" else "loopback ("), name
        for _ in range(2):
            h.dyn(h.state, 112.5, n_map=1)
        h.synchronize()
        res[mode] = {n: getattr(h.state, n).storage.clone() for n in ("delp", "pt", "u", "v", "w", "delz", "q_con")}
        del h
    for n, a in res["copy"].items():
        assert torch.isfinite(a).all(), n
        assert torch.equal(a, res["rccl"][n]), f"{n}: the RCCL self-loop moved different bytes than the device copies"


@pytest.mark.gpu
def test_bench_init_and_teardown_order_with_both_rccl_users(tmp_path):
    """The 8-GPU run has TWO users of RCCL in one process: torch's NCCL process group (launcher-side barriers / reductions) and the
    library's own communicator (the halo messages).  bench.py creates the process group first, the communicator inside the harness,
    and at the end destroys the communicator (with the context) BEFORE the process group.  What one GPU can execute of that: both
    with ONE rank each (FV3_BENCH_FORCE_PG=1: the process group of a real launch, world size 1; --emulate-share 8 with
    FV3_LOOPBACK_TRANSPORT=rccl: the library's communicator, every message an ncclSend / ncclRecv to self), the share's acoustic
    steps in between, then the teardown -- the process must print its line and exit 0 within the timeout (an init-order or
    teardown hang is what the driver's 8-GPU run would otherwise find first)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("FV3_")}
    env.update(FV3_BENCH_FORCE_PG="1", FV3_LOOPBACK_TRANSPORT="rccl", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    r = _run([sys.executable, os.path.join(root, "bench.py"), "--config", "c48", "--emulate-share", "6", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, timeout=1200, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["finite"] and line["halo_transport"].startswith("loopback-rccl"), line["halo_transport"]
