#!/usr/bin/env python3
"""In-kernel phase stamps of the marching kernels (diagnostic build: FV3_LIB_TAG=stamps, built with -DFV3_STAMPS).

    FV3_LIB_TAG=stamps python tools/exp/stamps.py [--config c768] [--out file.md]

Runs one acoustic call (n_split 2) of the bench workload and prints, per marching kernel, the mean shader-clock cycles a wave
spends per step in each phase (csrc/fv3_common.h FV3_STAMP): 0 = loop overhead between steps, 1 = issuing the step's loads
(incl. the waits the register rotation forces: the copies of rows still in flight), 2 = waiting for row r, 3 = phase 1
(M-direction sweep of q, del-n own-lane part), 4 = phase 2 (L-direction sweeps), 5 = phase 3 (second M sweep, epilogue, stores).
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import faulthandler

    faulthandler.enable()
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c768")
    ap.add_argument("--out", default=None)
    ap.add_argument("--kernels", default="1111,1211,62288,62280", help="kernel ids to record, one measured step each (0 = whatever comes first)")
    a = ap.parse_args()
    from pace_amd import lib as L
    from pace_amd.harness import CONFIGS, DycoreHarness

    kw = dict(CONFIGS[a.config])
    kw["k_split"], kw["n_split"] = 1, 2
    print("[stamps] building the harness", flush=True)
    h = DycoreHarness(device="cuda:0", verbose=True, **kw)
    lib = L.load(64)
    print("[stamps] library", L._build.lib_path(64), flush=True)
    lib.fv3_stamps_read.restype = C.c_long
    lib.fv3_stamps_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_long]
    h.step()
    torch.cuda.synchronize()
    print("[stamps] warm-up step done", flush=True)
    lib.fv3_stamps_reset.argtypes = [C.c_ulonglong]
    cap = 16384
    names = {1111: "dsw_scalars AIR interior FD", 1211: "dsw_scalars TRC interior FD", 1121: "dsw_scalars AIR edge FD", 1221: "dsw_scalars TRC edge FD",
             2000 + 288: "tp2d<288> vorticity + winds FD", 2000 + 280: "tp2d<280> interface heights FD", 62288: "tp2d<288, HC 6> vorticity + winds (FA)",
             62280: "tp2d<280, HC 6> interface heights (FA)"}
    lines = [f"config {a.config}, one model step of 2 acoustic sub-steps per kernel, up to {cap} waves recorded each; shader-clock cycles per wave and step", "",
             "| kernel | waves | steps/wave | cycles/step | between | issue loads | wait row | phase 1 | phase 2 | phase 3 |", "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for want in [int(x) for x in a.kernels.split(",")]:
        print("[stamps] reset ->", lib.fv3_stamps_reset(want), flush=True)
        h.step()
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * (cap * 8))()
        n = lib.fv3_stamps_read(buf, cap)
        rec = np.frombuffer(buf, dtype=np.uint64).reshape(cap, 8)[:n].astype(np.float64)
        print(f"[stamps] kernel {want}: {n} records", flush=True)
        for kid in sorted(set(rec[:, 0].astype(int))):
            r = rec[rec[:, 0] == kid]
            steps = r[:, 1].sum()
            ph = r[:, 2:8].sum(axis=0) / max(steps, 1)
            lines.append(f"| {names.get(kid, kid)} | {len(r)} | {steps / len(r):.1f} | {ph.sum():.0f} | " + " | ".join(f"{v:.0f} ({100 * v / ph.sum():.0f} %)" for v in ph) + " |")
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
