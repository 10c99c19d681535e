"""Dycore-only driver: runs the acoustic dynamics from a pace driver yaml.

    python -m pace_amd.driver driver/examples/configs/baroclinic_c12.yaml [--steps N] [--out perf.json]

It reproduces the reference's time loop for ``dycore_only: true`` + ``disable_step_physics: true``
[REF driver/pace/driver/driver.py:627-662]: one timer entry per model step, a step being ``k_split`` x
[AcousticDynamics (+ tracer advection with ``--tracers N``, + the vertical remap with ``--remap``)], and writes
the per-step times in the layout the reference's performance collector uses
(``{"times": {<timer>: {"hits": n, "times": [[...per step...] per rank]}}}``, timers ``mainloop``, ``DynCore``,
``TracerAdvection``, ``Remapping`` [REF tests/main/driver/test_driver.py:77-121]).  With ``--tracers N --remap`` the step is the
body of ``DynamicalCore.step_dynamics`` and the outer timer is called ``mainloop``:
[REF .jenkins/print_performance_number.py:13-14] (mean of steps 2..N per rank) then runs on the file unchanged.  Without them
the outer timer is ``acoustic_mainloop`` (it times less than the reference's ``mainloop`` does).
The number of steps comes from ``seconds`` / ``minutes`` / ``hours`` / ``days`` and ``dt_atmos`` as in
the reference [REF driver/pace/driver/driver.py:305-337 (total_time / n_steps)].

Keys read from the yaml: ``nx_tile, nz, layout, dt_atmos, seconds|minutes|hours|days,
dycore_config.*`` (the fields of ``AcousticDynamicsConfig``), ``stencil_config.compilation_config.
{backend, device_sync}``, ``initialization.type`` (``analytic``/anything else -> the synthetic
recipe of SURVEY §8d; the analytic baroclinic state is not part of this build and the driver says
so).  ``backend`` values other than ``hip:gfx950`` are reported and replaced: this build has one
backend.  Multi-process runs take RANK / WORLD_SIZE / LOCAL_RANK from the environment like bench.py.
"""
from __future__ import annotations

import argparse
import dataclasses
import json
import os
import sys

import yaml


def load_config(path: str):
    with open(path) as f:
        y = yaml.safe_load(f)
    from .config import AcousticDynamicsConfig

    known = {f.name for f in dataclasses.fields(AcousticDynamicsConfig)}
    dy = {k: v for k, v in (y.get("dycore_config") or {}).items() if k in known}
    ignored = sorted(k for k in (y.get("dycore_config") or {}) if k not in known)
    total = 0.0
    for key, mult in (("seconds", 1.0), ("minutes", 60.0), ("hours", 3600.0), ("days", 86400.0)):
        total += float(y.get(key, 0) or 0) * mult
    dt_atmos = float(y["dt_atmos"])
    run = dict(
        nx_tile=int(y["nx_tile"]),
        nz=int(y["nz"]),
        layout=tuple(int(v) for v in y.get("layout", (1, 1))),
        dt_atmos=dt_atmos,
        n_steps=max(1, int(round(total / dt_atmos))) if total > 0 else 1,
        backend=((y.get("stencil_config") or {}).get("compilation_config") or {}).get("backend", "hip:gfx950"),
        device_sync=bool(((y.get("stencil_config") or {}).get("compilation_config") or {}).get("device_sync", False)),
        init=(y.get("initialization") or {}).get("type", "analytic"),
        case=(((y.get("initialization") or {}).get("config") or {}).get("case", "baroclinic")),
        dycore_only=bool(y.get("dycore_only", False)),
        disable_step_physics=bool(y.get("disable_step_physics", False)),
        experiment=((y.get("performance_config") or {}).get("experiment_name", os.path.splitext(os.path.basename(path))[0])),
    )
    return run, dy, ignored


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("config")
    ap.add_argument("--steps", type=int, default=None, help="override the number of model steps from the yaml")
    ap.add_argument("--out", default=None, help="performance json (default: <experiment>_fv3_mi355x.json)")
    ap.add_argument("--precision", type=int, default=64)
    ap.add_argument("--restart", default=None, help="start from restart_dycore_state_<rank>.nc files in this directory [REF driver/pace/driver/state.py:154-172]")
    ap.add_argument("--save-restart", default=None, help="write restart_dycore_state_<rank>.nc files there after the last step [REF state.py:114-123]")
    ap.add_argument("--tracers", type=int, default=0, help="advect N synthetic tracers after every acoustic call (TracerAdvection, hord_tr from the yaml)")
    ap.add_argument("--remap", action="store_true", help="Lagrangian-to-Eulerian remap after every acoustic call (with --tracers: the body of DynamicalCore.step_dynamics)")
    a = ap.parse_args(argv)
    run, dy, ignored = load_config(a.config)

    import torch

    from .harness import DycoreHarness

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    say = (lambda *x: print("[driver]", *x, flush=True)) if rank == 0 else (lambda *x: None)
    if run["backend"] not in ("hip:gfx950", "hip"):
        say(f"backend {run['backend']!r} requested by the yaml -> running 'hip:gfx950' (the only backend of this build)")
    if not (run["dycore_only"] and run["disable_step_physics"]):
        say("physics is outside this build: running the dycore-only loop")
    say("step = k_split x [acoustic dynamics" + (f", advection of {a.tracers} tracers" if a.tracers else "") + (", vertical remap" if a.remap else "") + "]"
        + ("" if (a.tracers and a.remap) else "  (--tracers N --remap add the rest of step_dynamics)"))
    if run["init"] == "analytic" and str(run["case"]).startswith("baroclinic"):
        init = "baroclinic"
        say("initialization: JW2006 baroclinic wave (pace_amd.init.baroclinic_state; restated from the paper, see its docstring)")
    else:
        init = "synthetic"
        say(f"initialization {run['init']!r}/{run['case']!r} is not available in this build: using the synthetic recipe of SURVEY §8d")
    if ignored:
        say("dycore_config keys not read by the acoustic path:", ", ".join(ignored))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    n_ranks = 6 * run["layout"][0] * run["layout"][1]
    if n_ranks % world:
        sys.exit(f"{n_ranks} sub-domains are not divisible over {world} processes")
    dtype = torch.float64 if a.precision == 64 else torch.float32
    kw = {k: dy[k] for k in ("k_split", "n_split") if k in dy}
    h = DycoreHarness(nx_tile=run["nx_tile"], nz=run["nz"], layout=run["layout"], dt_atmos=run["dt_atmos"], world_size=world, proc=rank,
                      device=f"cuda:{local_rank}", dtype=dtype, verbose=(rank == 0), init=init, config_overrides={k: v for k, v in dy.items() if k not in ("k_split", "n_split")},
                      n_tracers=a.tracers, hord_tr=int(dy.get("hord_tr", 8)), remap=a.remap, **kw)
    if run["device_sync"]:
        h.sf.set_device_sync(True)
    if a.restart:
        from . import restart

        restart.load_state(h.state, h.layout.local_ranks, a.restart, extra=h.tracers)
        h.dyn._bind(h.state)
        say(f"state loaded from {a.restart}")
    n_steps = a.steps or run["n_steps"]
    from .timer import Timer

    # The reference's timestep timer: one "mainloop" entry per model step [REF driver/pace/driver/driver.py:640], and inside it the
    # dycore's own clocks ("DynCore" = the acoustic dynamics, "TracerAdvection", "Remapping"), collected per step like its
    # performance collector does (times_per_step / hits_per_step [REF tests/main/driver/test_driver.py:77-121]).  When the step is
    # not the whole body of step_dynamics (no --tracers / --remap) the outer clock is called "acoustic_mainloop" instead, so that
    # nobody compares it with the reference's "mainloop".
    full_step = bool(a.tracers and a.remap)
    loop_name = "mainloop" if full_step else ("acoustic_mainloop" if not (a.tracers or a.remap) else "dynamics_mainloop")
    # (device synchronisation on the OUTER clock only: a synchronising nested clock would serialise the streams inside the timed step -- the nested
    #  dycore clocks then measure host-side enqueue intervals, and the json says so)
    timer = Timer(sync=h.synchronize, sync_names=(loop_name,))
    times_per_step, hits_per_step = [], []
    for step in range(n_steps):
        timer.reset()
        with timer.clock(loop_name):
            h.step(timer)  # dycore.step_dynamics for dycore_only + disable_step_physics
        times_per_step.append(timer.times)
        hits_per_step.append(timer.hits)
    times = [t[loop_name] for t in times_per_step]
    ok = all(v[2] for v in h.sanity().values())
    if a.save_restart:
        from . import restart

        restart.save_state(h.state, h.layout.local_ranks, a.save_restart, extra=h.tracers)
        say(f"restart files written to {a.save_restart}")
    # per reference rank (the ranks a process owns step together: they share its clocks)
    local = {r: {n: [t.get(n, 0.0) for t in times_per_step] for n in times_per_step[0]} for r in h.layout.local_ranks}
    hits = {n: sum(hh.get(n, 0) for hh in hits_per_step) for n in hits_per_step[0]}
    if world > 1:
        import torch.distributed as dist

        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        local = {}
        for g in gathered:
            local.update(g)
        dist.destroy_process_group()
    if rank == 0:
        names = list(local[min(local)])
        report = {n: {"hits": hits[n], "times": [local[r][n] for r in sorted(local)]} for n in names}  # TimeReport(hits, times[rank][step])
        per_rank = report[loop_name]["times"]
        mean = sum(per_rank[0][1:]) / max(1, len(per_rank[0]) - 1) if n_steps > 1 else per_rank[0][0]
        sdpd = run["dt_atmos"] / mean
        out = a.out or f"{run['experiment']}_fv3_mi355x.json"
        json.dump({"setup": {"experiment": run["experiment"], "nx_tile": run["nx_tile"], "nz": run["nz"], "layout": list(run["layout"]), "dt_atmos": run["dt_atmos"],
                             "k_split": h.cfg.k_split, "n_split": h.cfg.n_split, "n_gpus": world, "backend": "hip:gfx950", "dycore_only": True, "acoustic_only": not (a.tracers or a.remap), "tracers": a.tracers, "remap": bool(a.remap), "finite": ok,
                             "note": ("a step here is k_split AcousticDynamics calls; the reference's dycore_only mainloop (DynamicalCore.step_dynamics) also runs tracer "
                                      "advection and the Lagrangian-to-Eulerian remap (--tracers N --remap add them): not comparable with the reference's 'mainloop' timer")
                             if not (a.tracers and a.remap) else
                             "a step is k_split x [AcousticDynamics, tracer advection, vertical remap] = the body of DynamicalCore.step_dynamics without physics and "
                             "moist thermodynamics"},
                   # the reference collector's layout: times.<timer> = {hits, times[rank][step]}; "mainloop" only when the step is the
                   # whole body of step_dynamics (--tracers N --remap), then .jenkins/print_performance_number.py runs on this file as is
                   "times": report,
                   "times_note": ("only the outer clock ('" + loop_name + "') synchronises the device; the nested clocks (DynCore / TracerAdvection / Remapping) are host-side "
                                  "enqueue intervals. " + ("'mainloop' here = the dry body of step_dynamics with N synthetic tracers: no moist thermodynamics, no physics coupling "
                                                           "-- the reference's mainloop does more per step." if full_step else "")),
                   ("acoustic_simulated_days_per_day" if not (a.tracers or a.remap) else "dynamics_simulated_days_per_day"): sdpd}, open(out, "w"))
        say(f"{n_steps} steps of dt_atmos={run['dt_atmos']:g}s: acoustic mainloop mean (first step dropped) {mean * 1e3:.2f} ms -> {sdpd:.2f} simulated-days/day ({'acoustic dynamics only' if not (a.tracers or a.remap) else 'acoustic dynamics' + (f' + {a.tracers} tracers' if a.tracers else '') + (' + remap' if a.remap else '')}); state finite: {ok}; wrote {out}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
