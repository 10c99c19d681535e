#!/usr/bin/env python3
"""Headline benchmark of the MI355X FV3 acoustic dycore (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (all N): C768 L79 fp64, layout 2x2 = 24 sub-domains of 384^2 (BASELINE cfg-3; the
configuration the metric is quoted on -- it fits one MI355X: ~130 GB of fields), dt_atmos 225 s,
k_split 2, n_split 6, dycore-only.  The 24 sub-domains are split over the N processes
(24/N each, strong scaling); a "step" is one model step = k_split AcousticDynamics calls =
12 acoustic sub-steps.  value = simulated-days/day = dt_atmos / mean step seconds (max over ranks).

Extra objects on the JSON line:
  roofline     d_sw (all launches of one fv3_d_sw call), algorithmic bytes = 33 field passes x 8 B x
               local cells / mean HIP-event duration of the call, vs 8 TB/s HBM peak; beside it the two floors of the
               call: hbm_floor_ms (algorithmic bytes at 8 TB/s) and valu_floor_ms (VALU wave-instructions x 4 cycles / 1024 SIMDs /
               2.4 GHz, from the committed SQ-counter pass) -- the kernel is bound by whichever is larger
  cpu_baseline the numpy oracle on the host cores on a bounded sample: a C48 L79 cube, one pinned single-threaded process
               per tile (6 processes), a few acoustic sub-steps, scaled per cell to the C768 step -- baseline only
  operators    per-operator mean milliseconds per acoustic sub-step (HIP events recorded by fv3_acoustic_step)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
D_SW_PASSES = 33  # algorithmic field passes of d_sw


_CPU_WORKER = r"""
import json, os, sys, time
tile, core, budget, root = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
try:
    os.sched_setaffinity(0, {core})
except Exception:
    pass
for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[v] = "1"
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import numpy as np
from fv3_oracle.dyn_core import OracleAcousticDynamics
from fv3_oracle.util import Dom
from pace_amd.config import AcousticDynamicsConfig
from pace_amd.constants import get_constants
from pace_amd.grid import make_grid
from pace_amd.init import synthetic_state
from pace_amd.topology import CubedSpherePartitioner

class NoExchange:  # one tile per process: the halo cells keep their initial (smooth, finite) values -- same arithmetic per cell
    def scalar(self, *a, **k): pass
    def vector(self, *a, **k): pass
    def synchronize_vector_interfaces(self, *a, **k): pass

n, nz = 48, 79
c = get_constants()
part = CubedSpherePartitioner(n, (1, 1))
cfg = AcousticDynamicsConfig(npx=n + 1, npy=n + 1, npz=nz, n_split=1, k_split=1)
g = make_grid(part, tile, nz=nz)
s = synthetic_state(g, rank=tile)
phis = s.pop("phis")
one = CubedSpherePartitioner(n, (1, 1))
dyn = OracleAcousticDynamics.__new__(OracleAcousticDynamics)
OracleAcousticDynamics.__init__(dyn, part, [g] * 6, cfg, c, [phis] * 6)
dyn.nranks, dyn.doms, dyn.tmp, dyn.ex = 1, dyn.doms[:1], dyn.tmp[:1], NoExchange()
dyn.zs, dyn.phis = dyn.zs[:1], dyn.phis[:1]
states = [s]
dt_sub = 225.0 / 2 / 6
dyn(states, dt_sub, 1)  # warm-up (imports, page faults)
times = []
t_all = time.time()
while len(times) < 8 and time.time() - t_all < budget:
    t0 = time.time()
    dyn(states, dt_sub, 1)
    times.append(time.time() - t0)
ok = bool(np.isfinite(states[0]["delp"]).all())
print(json.dumps({"tile": tile, "core": core, "t_sub": float(np.mean(times)), "n": len(times), "finite": ok}))
"""


def cpu_baseline(seconds_budget=20.0):
    """The numpy oracle on the host cores, as SURVEY 8d asks: a C48 L79 cube, ONE single-threaded process per tile pinned
    to its own core (6 processes; the processes never touch the GPU), up to 8 acoustic sub-steps each (~10-20 s);
    wall time of a sub-step = the slowest process.  Checker code used as a *reported* baseline only."""
    import subprocess

    ncpu = os.cpu_count() or 1
    try:
        cores = sorted(os.sched_getaffinity(0))
    except Exception:
        cores = list(range(ncpu))
    nproc = 6
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, "-c", _CPU_WORKER, str(t), str(cores[t % len(cores)]), str(seconds_budget), ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=env) for t in range(nproc)]
    res = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        if p.returncode != 0:
            raise RuntimeError("cpu_baseline worker failed:\n" + err[-2000:])
        res.append(json.loads(out.strip().splitlines()[-1]))
    t_sub = max(r["t_sub"] for r in res)
    n, nz = 48, 79
    cells = 6 * n * n * nz
    model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    sample = (f"numpy {np.__version__} oracle, C{n} L{nz} cube ({cells} cells), one single-threaded process per tile pinned to its own core ({nproc} of {ncpu} host cores, "
              f"{model}), {min(r['n'] for r in res)} acoustic sub-steps, slowest process {t_sub:.2f} s / sub-step (per-process " + ", ".join("%.2f" % r["t_sub"] for r in res) + ")")
    return t_sub / cells, sample, nproc


def _flush_c_stdio():
    """Flush the C-level stdio buffers of this process (libraries that printf into a pipe are fully buffered: their text would
    otherwise appear at exit, after the JSON line the driver reads)."""
    import ctypes

    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def _hbm_used_gb():
    try:
        free, total = torch.cuda.mem_get_info()
        return round((total - free) / 1.0e9, 1)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c768", help="c768 (headline) | c384 | c272 | c192 | c48 | c12")
    ap.add_argument("--nz", type=int, default=None)
    ap.add_argument("--precision", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-op-timing", action="store_true")
    ap.add_argument("--k-split", type=int, default=None, help="override (profiling runs only; changes the workload)")
    ap.add_argument("--n-split", type=int, default=None, help="override (profiling runs only; changes the workload)")
    ap.add_argument("--tracers", type=int, default=0, help="also advect N tracers after every acoustic call (tracer_2d_1l, hord_tr 8): a separate, "
                    "clearly named workload -- the headline metric is the acoustic dynamics alone")
    ap.add_argument("--emulate-share", type=int, default=0, metavar="N", help="ONE process owning the first 1/N of the sub-domains (the per-GPU share of an "
                    "N-GPU run) with every inter-process message looped back on the device: the per-GPU compute + pack / unpack time of the N-GPU run without "
                    "the other GPUs.  A separately named workload, never the headline value")
    ap.add_argument("--remap", action="store_true", help="also run the Lagrangian-to-Eulerian vertical remap after every acoustic call (+ tracer advection): "
                    "with --tracers the body of DynamicalCore.step_dynamics; a separate, clearly named workload")
    ap.add_argument("--graph", action="store_true", help="capture one model step into a HIP graph after the warm-up and REPLAY it in the timed region (one process only: "
                    "device-local or looped-back halo transport -- nothing in a step touches the host there; the RCCL launch path of an N-GPU run stays exactly as "
                    "tested).  Same kernels, same values; what it removes is the launch path (~110 launches per sub-step), which matters for the small per-GPU shares. "
                    "Implies --no-op-timing (event pairs are not part of a replay); the line says graph_replay: true")
    a = ap.parse_args()

    from pace_amd.harness import CONFIGS, DycoreHarness

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    group = None
    # Test hook for a 1-GPU box (tests/test_gpu_invariants.py): FV3_BENCH_FORCE_PG=1 with WORLD_SIZE=1 creates the torch process group
    # of ONE rank in the order and with the arguments of a real N-GPU launch and runs the N-GPU teardown at the end; with
    # --emulate-share N and FV3_LOOPBACK_TRANSPORT=rccl the library's own one-rank RCCL communicator then lives beside torch's, is
    # created after it and destroyed before it -- the init-order / teardown sequence of the 8-GPU run, minus the peers.
    force_pg = world == 1 and os.environ.get("FV3_BENCH_FORCE_PG") == "1"
    if force_pg:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # (test hooks for a 1-GPU box: FV3_FORCE_DEVICE=0 puts every rank on GPU 0, FV3_DIST_BACKEND=gloo exchanges through
        #  pinned host memory -- RCCL needs one device per rank; tests/test_gpu_invariants.py runs the N = 2 path this way)
        backend = os.environ.get("FV3_DIST_BACKEND", "nccl")
        if "FV3_FORCE_DEVICE" in os.environ:
            local_rank = int(os.environ["FV3_FORCE_DEVICE"])
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)
    kw = dict(CONFIGS[a.config])
    if a.nz:
        kw["nz"] = a.nz
    if a.k_split:
        kw["k_split"] = a.k_split
    if a.n_split:
        kw["n_split"] = a.n_split
    if (6 * kw["layout"][0] * kw["layout"][1]) % world:
        sys.exit(f"{a.config} has {6 * kw['layout'][0] * kw['layout'][1]} sub-domains: not divisible over {world} GPUs")
    dtype = torch.float64 if a.precision == 64 else torch.float32
    graph_stream = None
    if a.graph:
        if world != 1:
            sys.exit("--graph: one process only (device-local / looped-back halo transport); the N-GPU launch path is not replayed from a graph")
        a.no_op_timing = True
        a.warmup = max(a.warmup, 2)  # (the sequencer's lazy allocations happen in the first two steps; a capture must not allocate)
        graph_stream = torch.cuda.Stream()  # (a capture needs a side stream; everything from the harness on runs on it)
        torch.cuda.set_stream(graph_stream)
    share = a.emulate_share if (a.emulate_share and world == 1) else 0
    if share and (6 * kw["layout"][0] * kw["layout"][1]) % share:
        sys.exit(f"--emulate-share {share}: {6 * kw['layout'][0] * kw['layout'][1]} sub-domains do not divide")
    h = DycoreHarness(world_size=share or world, proc=0 if share else rank, device=f"cuda:{local_rank}", dtype=dtype, group=group, verbose=(rank == 0), n_tracers=a.tracers,
                      remap=a.remap, loopback=bool(share), **kw)
    transport = h.dyn.halo.transport_name
    _flush_c_stdio()  # (RCCL's version banner sits in the C stdio buffer of a piped run: out now, not behind the result line at exit)
    # A fallback nobody asked for must not produce a number: with an nccl group the messages go through the library's RCCL
    # transport or the run fails (FV3_HALO_NATIVE=0 asks for the torch.distributed path explicitly; gloo groups are test hooks)
    if world > 1 and os.environ.get("FV3_DIST_BACKEND", "nccl") == "nccl" and os.environ.get("FV3_HALO_NATIVE", "1") != "0" and transport != "rccl-native":
        sys.exit(f"[bench] the native RCCL halo transport is not active on rank {rank} (transport: {transport}; reason: {h.dyn.halo.fallback_reason}) -- refusing to report; "
                 "set FV3_HALO_NATIVE=0 to benchmark the torch.distributed path on purpose")
    rccl_ranks = 0
    if world > 1:
        import torch.distributed as dist

        tt = torch.tensor([1 if transport == "rccl-native" else 0], device=f"cuda:{local_rank}", dtype=torch.int32)
        dist.all_reduce(tt)
        rccl_ranks = int(tt.item())

    # ---- per-operator HIP-event timing: fv3_acoustic_step brackets every operator with an event
    #      pair on the stream it launches on (fv3_ctx_set_profiling / fv3_profile_read)
    # (N > 1: d_sw only -- the roofline kernel; ~50 event pairs per sub-step cost up to 2 % of a 12 ms sub-step, measured on the emulated 1/8 share)
    if not a.no_op_timing:
        h.sf.set_profiling(1 if world == 1 else 2)

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        h.step()
    barrier()
    if not a.no_op_timing:
        h.sf.profile(reset=True)
    graph = None
    if a.graph:
        graph = torch.cuda.CUDAGraph()
        graph.capture_begin()
        h.step()
        graph.capture_end()
        barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        if graph is not None:
            graph.replay()
        else:
            h.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist

        tt = torch.tensor([elapsed], device=f"cuda:{local_rank}", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    sanity = h.sanity()
    ok = all(v[2] for v in sanity.values())

    def gather_parts(mine):
        import torch.distributed as dist

        every = [None] * world
        dist.all_gather_object(every, mine)
        return every

    checksum = h.checksum(gather_parts if world > 1 else None)
    if world > 1:
        import torch.distributed as dist

        flags = [None] * world
        dist.all_gather_object(flags, ok)
        ok = all(flags)
    if rank == 0:
        cfg = h.cfg
        s_per_step = elapsed / a.steps
        n_sub_steps = cfg.k_split * cfg.n_split
        sdpd = cfg.dt_atmos / s_per_step
        op_ms = {}
        if not a.no_op_timing:
            for name, (ms, calls) in h.sf.profile(reset=True).items():
                if name in ("glue", "halo", "diffusive_heating", "pk3_halo_edge_pe"):
                    op_ms[name] = ms / (a.steps * n_sub_steps)  # several launches per sub-step: quote the sum per sub-step
                else:
                    op_ms[name] = ms / calls
        line = {
            "metric": "simulated-days/day + acoustic-step ms, C768 L79 fp64, 1/2/4/8 MI355X",
            "value": sdpd,
            "unit": "simulated-days/day",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1000.0 * s_per_step,
            "acoustic_step_ms": 1000.0 * s_per_step / n_sub_steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if a.precision == 64 else "f32",
            "data": "synthetic",
            "config": {
                "workload": ("EMULATED SHARE 1/%d: " % share if share else "") + f"C{kw['nx_tile']} L{kw['nz']} layout {kw['layout'][0]}x{kw['layout'][1]} ({h.part.total_ranks} sub-domains of {h.part.nx}^2, {len(h.grids)} per GPU), "
                f"dt_atmos {cfg.dt_atmos:g} s, k_split {cfg.k_split}, n_split {cfg.n_split}, "
                + ("acoustic dynamics"
                   + (f" + tracer advection of {a.tracers} tracers after every acoustic call (tracer_2d_1l, hord_tr 8)" if a.tracers else "")
                   + (" + Lagrangian-to-Eulerian remap (kord 9) after every acoustic call" if a.remap else "")
                   + "; no physics" if (a.tracers or a.remap) else "dycore-only acoustic dynamics (no tracer/remap/physics)"),
                "sub_steps_per_step": n_sub_steps,
                "cells_global": h.cells_global,
            },
            "finite": ok,
            "state_checksum": checksum,
            "hbm_used_GB": _hbm_used_gb(),  # device memory in use on this rank when the timed region ended (hipMemGetInfo: fields, workspace, library scratch, torch)
            "halo_transport": transport,
            "rccl_ranks": rccl_ranks,
            "sub_domains_per_gpu": len(h.grids),
            "pingpong_scalars": os.environ.get("FV3_PINGPONG", "1") != "0",
            "graph_replay": bool(a.graph),
        }
        if a.tracers or a.remap:
            line["metric"] = ("simulated-days/day of the step_dynamics body (acoustic dynamics"
                              + (f" + tracer advection of {a.tracers} tracers" if a.tracers else "") + (" + vertical remap" if a.remap else "")
                              + ") -- a separately named workload, not the headline metric")
        if share:
            line["metric"] = f"EMULATED per-GPU share of an {share}-GPU run (one process alone, messages looped back) -- not the headline metric"
            line["emulated_share"] = {"of_gpus": share, "sub_domains": len(h.grids), "ideal_ms_per_substep_from_1gpu": None,
                                      "note": "compute + pack / unpack of the sub-domains one GPU owns in the N-GPU run; no inter-GPU transfer time, halo values are not the neighbours'"}
        if "d_sw" in op_ms:
            alg = D_SW_PASSES * (8 if a.precision == 64 else 4) * h.cells_local
            ach = alg / (op_ms["d_sw"] * 1e-3) / 1e9
            # HBM-side bytes of one fv3_d_sw call from the committed PMC passes (rocprofv3 cannot run
            # inside the bench); only quoted when the profile was taken on this very workload
            # Both files carry the hash of the kernel sources they were measured on (tools/pmc_traffic.py / pmc_sq.py); the loaded library carries
            # the hash of the sources it was built from (fv3_build_id).  A counter file of another tree is reported as null with the reason.
            build_id = h.sf.lib.fv3_build_id().decode()
            same_workload = a.config == "c768" and world == 1 and a.precision == 64 and not a.nz

            def committed(name, key):
                pth = os.path.join(ROOT, "profiles", name)
                if not os.path.exists(pth):
                    return None, f"profiles/{name} not found", None
                if not same_workload:
                    return None, f"profiles/{name} was measured on the C768 L79 fp64 one-GPU workload, not on this one", None
                rec = json.load(open(pth))
                if rec.get("csrc_hash") != build_id:
                    return None, f"profiles/{name} was measured on kernel sources {rec.get('csrc_hash')}, this library is built from {build_id}: stale, not quoted", None
                return rec[key], f"profiles/{name} (" + rec["source"] + f"; kernel sources {build_id})", rec

            traffic, traffic_src, trec = committed("traffic_d_sw.json", "bytes")
            # the rate fv3_copy (one field read, one written) reached in the SAME counter pass: what "bandwidth-bound" means on this chip for this access pattern
            copy_gbps = trec.get("copy_GBps") if trec else None
            valu_floor, valu_src, vrec = committed("valu_d_sw.json", "valu_floor_ms")
            valu_floor_clk = vrec.get("valu_floor_ms_measured_clock") if vrec else None
            clk = vrec.get("clock_hz_measured") if vrec else None
            hbm_floor = alg / (HBM_PEAK_GBPS * 1e9) * 1e3
            # bound: whichever floor of the call is higher -- the HBM floor of the bytes the call actually moves (the measured traffic when the committed
            # counter file is current, else the algorithmic bytes) against the VALU floor at the clock the chip held while the counters were taken
            vf = valu_floor_clk or valu_floor
            hbm_floor_traffic = (traffic / (HBM_PEAK_GBPS * 1e9) * 1e3) if traffic else None
            line["roofline"] = {
                "kernel": "d_sw (all launches of one fv3_d_sw call)",
                "bound": "hbm" if (vf is None or max(hbm_floor, hbm_floor_traffic or 0.0) >= vf) else "valu-issue (the HBM fraction below is still the algorithmic bytes over the HBM peak)",
                "achieved": ach,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBPS,
                "frac_of_measured_copy": (ach / copy_gbps) if copy_gbps else None,
                "measured_copy_GBps": copy_gbps,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_call": alg,
                "ms_per_call": op_ms["d_sw"],
                "hbm_floor_ms": hbm_floor,
                "hbm_floor_ms_at_measured_traffic": hbm_floor_traffic,
                "valu_floor_ms": valu_floor,
                "valu_floor_ms_at_measured_clock": valu_floor_clk,
                "shader_clock_hz_measured": clk,
                "valu_floor_source": valu_src,
                "library_build_id": build_id,
                # non-null: NOT the product build (extra compile flags / per-file overrides / a variant tag: pace_amd/build.py variant_suffix) -- the committed
                # counter files never match such an id, so traffic / VALU floors are null for it
                "library_variant": (build_id.partition("+")[2] or None),
            }
        # the same fraction for every operator with a pass count in SURVEY §8a (algorithmic bytes per cell = passes x sizeof(Real))
        passes = {"c_sw": 15, "update_dz_c": 4, "riem_solver_c": 8, "p_grad_c": 7, "d_sw": D_SW_PASSES, "update_dz_d": 6, "riem_solver3": 11, "nh_p_grad": 8}
        line["roofline_operators"] = {
            k: {"ms": op_ms[k], "algorithmic_GB": p_ * (8 if a.precision == 64 else 4) * h.cells_local / 1e9,
                "frac": p_ * (8 if a.precision == 64 else 4) * h.cells_local / (op_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBPS}
            for k, p_ in passes.items() if k in op_ms and op_ms[k] > 0
        }
        if "d_sw" in op_ms:
            # counter traffic of every operator (one sub-step, the same two PMC passes as roofline.traffic; null + reason when the file is of another tree / workload)
            orec_ops, orec_src, orec = committed("traffic_operators.json", "operators")
            for k, v in line["roofline_operators"].items():
                t_ = (orec_ops or {}).get(k, {}).get("traffic_GB")
                v["traffic_GB"] = t_
                v["traffic_over_algorithmic"] = (t_ / v["algorithmic_GB"]) if t_ else None
                cg = orec.get("copy_GBps") if orec else None
                v["frac_of_measured_copy"] = (v["algorithmic_GB"] / (v["ms"] * 1e-3) / cg) if cg else None
            line["roofline_operators_traffic_source"] = orec_src
        line["operators_ms_per_substep"] = op_ms
        # (measurement inside the measurement, stated: the per-operator numbers come from one HIP-event pair per operator recorded INSIDE the
        #  timed region -- 13 pairs per sub-step on the compute stream; `--no-op-timing` runs without them: 108.46 / 108.75 ms per sub-step with,
        #  108.70 / 108.44 without, at C768 on one box -- no difference above the run-to-run spread)
        line["operator_timing"] = ("off" if a.no_op_timing else "HIP event pair around every operator inside the timed region" if world == 1
                                   else "HIP event pair around d_sw (the roofline kernel) inside the timed region")
        if not a.no_cpu_baseline:
            per_cell, sample, cores = cpu_baseline()
            t_step_cpu = per_cell * h.cells_global * n_sub_steps
            line["cpu_baseline"] = {
                "value": cfg.dt_atmos / t_step_cpu,
                "unit": "simulated-days/day",
                "cores": cores,
                "kind": "port",
                "sample": sample + "; scaled per cell to the benchmarked step (own numpy restatement -- stands in for the reference numpy backend, which cannot run offline)",
            }
        _flush_c_stdio()
        print(json.dumps(line), flush=True)
    if world > 1 or force_pg:
        import gc

        import torch.distributed as dist

        # orderly teardown: every rank is done with its exchanges (barrier), the library's communicator goes with the context while
        # the process group still exists, then the process group -- nothing RCCL-related is left for interpreter shutdown
        torch.cuda.synchronize()
        dist.barrier()
        h.close()
        del h
        gc.collect()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
