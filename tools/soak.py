#!/usr/bin/env python3
"""Long run of the step_dynamics body (acoustic calls + tracer advection + vertical remap) on the baroclinic-wave state:
global air / tracer mass, extrema and finiteness every few steps.  python tools/soak.py --nx 192 --steps 100"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pace_amd.harness import DycoreHarness  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=96)
    ap.add_argument("--nz", type=int, default=79)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--every", type=int, default=10)
    ap.add_argument("--dt", type=float, default=225.0)
    ap.add_argument("--k-split", type=int, default=2)
    ap.add_argument("--n-split", type=int, default=6)
    ap.add_argument("--tracers", type=int, default=2)
    ap.add_argument("--no-remap", action="store_true")
    ap.add_argument("--precision", type=int, default=64)
    a = ap.parse_args()
    h = DycoreHarness(a.nx, nz=a.nz, layout=(1, 1), dt_atmos=a.dt, k_split=a.k_split, n_split=a.n_split, init="baroclinic", n_tracers=a.tracers,
                      remap=not a.no_remap, device="cuda:0", dtype=torch.float64 if a.precision == 64 else torch.float32)
    n, nz = a.nx, a.nz
    area = [torch.as_tensor(g.area[3 : 3 + n, 3 : 3 + n], device="cuda:0") for g in h.grids]

    def diag():
        m = t = 0.0
        for i in range(len(h.grids)):
            dp = h.state.delp.sub(i).view[...][:n, :n, :nz].double()
            m += float((dp.sum(dim=2) * area[i]).sum())
            if h.tracers:
                q = h.tracers["tracer0"].sub(i).view[...][:n, :n, :nz].double()
                t += float(((dp * q).sum(dim=2) * area[i]).sum())
        s = h.sanity()
        temp_lo = temp_hi = float("nan")
        pt = h.state.pt.sub(0).view[...][:n, :n, :nz].double()
        pkz = h.state.pkz.sub(0).view[...][:n, :n, :nz].double()
        if not a.no_remap:
            temp_lo, temp_hi = float((pt * pkz).min()), float((pt * pkz).max())
        return m, t, s, temp_lo, temp_hi

    m0, t0, *_ = diag()
    t_start = time.time()
    for step in range(1, a.steps + 1):
        h.step()
        if step % a.every == 0 or step == a.steps:
            h.synchronize()
            m, t, s, tl, th = diag()
            ok = all(v[2] for v in s.values())
            print(f"step {step:4d} ({step * a.dt / 3600:.2f} h)  air mass {m / m0 - 1:+.2e}  tracer mass {(t / t0 - 1) if t0 else 0:+.2e}  "
                  f"u [{s['u'][0]:.1f}, {s['u'][1]:.1f}]  w [{s['w'][0]:.3f}, {s['w'][1]:.3f}]  T(tile 0) [{tl:.1f}, {th:.1f}]  finite {ok}  "
                  f"({time.time() - t_start:.0f} s)", flush=True)
            if not ok:
                sys.exit(1)


if __name__ == "__main__":
    main()
