#!/usr/bin/env python3
"""Regenerate the data fixtures under tests/golden/ from /root/reference.

Run in the build container only (the GPU box has no /root/reference).  The
fixtures are *data* the reference tree holds, down-selected:

* eta79.npz        - L79 ak/bk table
  [REF examples/notebooks/generate_eta_file_netcdf.ipynb:82-135]
* c12_restart_6tiles.npz - the real FV3 C12 L63 model state of the reference tree, ALL SIX tiles (u, v, W, DZ, T, delp, phis, sphum, liq_wat per tile, stacked on a leading tile
  axis) + ak/bk(64): the one reference-held dataset that pins the cube topology (adjacency, rotations, signs) -- see
  tests/test_restart_six_tiles.py  [REF tests/main/data/c12_restart/fv_core.res.tile[1-6].nc, fv_tracer.res.tile[1-6].nc;
  read by the reference in tests/main/driver/test_restart_fortran.py:21-67]
* reference_thresholds_fv_dynamics.json - the reference's own calibrated savepoint thresholds (absolute / relative per variable)
  of the C_SW-Out, D_SW-Out, Tracer2D1L-In / Out, Remapping-In / Out and FVDynamics-Out savepoints
  [REF tests/savepoint/thresholds/fv_dynamics.yaml:2-360]
"""
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def eta79():
    nb = json.load(open(os.path.join(REF, "examples/notebooks/generate_eta_file_netcdf.ipynb")))
    out = {}
    for cell in nb["cells"]:
        src = "".join(cell["source"])
        for name in ("ak", "bk"):
            m = re.search(name + r"\[:\]\s*=\s*np\.array\(\s*\[(.*?)\]\s*\)", src, re.S)
            if m:
                out[name] = np.array([float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()])
    assert out["ak"].shape == (80,) and out["bk"].shape == (80,), {k: v.shape for k, v in out.items()}
    np.savez(os.path.join(OUT, "eta79.npz"), **out)
    print("eta79.npz", out["ak"][:3], out["bk"][-3:])


def c12_restart_six_tiles():
    from scipy.io import netcdf_file

    d = os.path.join(REF, "tests/main/data/c12_restart")
    out = {}
    core = ("u", "v", "W", "DZ", "T", "delp", "phis")
    tracers = ("sphum", "liq_wat")  # specific humidity (virtual temperature) and the condensate the acoustic path carries as q_con
    for k in core + tracers:
        out[k] = []
    for t in range(1, 7):
        with netcdf_file(os.path.join(d, f"fv_core.res.tile{t}.nc"), "r", mmap=False) as f:
            for k in core:
                out[k].append(np.array(f.variables[k][0], dtype=np.float64))
        with netcdf_file(os.path.join(d, f"fv_tracer.res.tile{t}.nc"), "r", mmap=False) as f:
            for k in tracers:
                out[k].append(np.array(f.variables[k][0], dtype=np.float64))
    out = {k: np.stack(v) for k, v in out.items()}
    with netcdf_file(os.path.join(d, "fv_core.res.nc"), "r", mmap=False) as f:
        out["ak"] = np.array(f.variables["ak"][0], dtype=np.float64)
        out["bk"] = np.array(f.variables["bk"][0], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "c12_restart_6tiles.npz"), **out)
    print("c12_restart_6tiles.npz", {k: v.shape for k, v in out.items()})


def thresholds():
    import yaml

    d = yaml.safe_load(open(os.path.join(REF, "tests/savepoint/thresholds/fv_dynamics.yaml")))["savepoints"]
    out = {}
    for sec in ("C_SW-Out", "D_SW-Out", "Tracer2D1L-In", "Tracer2D1L-Out", "Remapping-In", "Remapping-Out", "FVDynamics-Out"):
        for item in d[sec]:
            for var, v in item.items():
                a, r = v.get("absolute"), v.get("relative")
                if a is None or a != a:  # (nan entries: the reference holds no number)
                    continue
                out[f"{sec}/{var}"] = {"absolute": float(a), "relative": None if r is None or r != r else float(r)}
    json.dump(out, open(os.path.join(OUT, "reference_thresholds_fv_dynamics.json"), "w"), indent=1, sort_keys=True)
    print("reference_thresholds_fv_dynamics.json", len(out), "entries")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference")
    os.makedirs(OUT, exist_ok=True)
    eta79()
    c12_restart_six_tiles()
    thresholds()
