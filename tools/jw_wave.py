#!/usr/bin/env python3
"""The perturbed Jablonowski-Williamson (2006) baroclinic wave run forward for several model days with the body of step_dynamics this build has
(acoustic calls + tracer advection + vertical remap; dry, no physics) -- the reference's default analytic case
[REF driver/examples/configs/baroclinic_c12.yaml:1-30; JW2006 = Jablonowski & Williamson, Q. J. R. Meteorol. Soc. 132 (2006) 2943-2975, sections 3 - 5].

What it is for: every parity test of this tree compares the HIP kernels with the tree's OWN restatement over one or a few
sub-steps.  A wrong coefficient shared by both passes all of them; it does not survive nine days of a flow whose published evolution is known: the wave
stays linear to day ~4 (surface-pressure deviations of a few hPa), deepens explosively between days 6 and 9 and breaks around day 9 (JW2006 figs. 4 - 6:
minimum surface pressure ~ 990 hPa at day 6... ~ 940 - 955 hPa at day 9 in the 1-degree reference solutions, maxima ~ 1020 hPa), and solutions of different
resolutions converge on one another until the wave breaks (their fig. 10: l2 differences of the surface pressure between resolutions stay under the
uncertainty of the reference solutions until day ~9).  The run prints / stores per model day: min / max surface pressure and where the minimum is, the
largest |w|, the global air-mass drift, finiteness -- and for a pair of resolutions (--compare) the area-weighted l2 difference of the surface pressure of the
coarser run against the finer one averaged onto the coarser grid (cell means: an integer refinement ratio makes that exact).

    python tools/jw_wave.py --nx 48 96 192 --days 9 --out gpurun_out/jw_wave.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pace_amd.harness import DycoreHarness  # noqa: E402

# (dt_atmos, k_split, n_split) per resolution: the acoustic step scales with the grid spacing (150 s at C48; the headline C768 run uses 18.75 s)
STEPPING = {24: (3600.0, 2, 6), 48: (1800.0, 2, 6), 96: (900.0, 2, 6), 192: (450.0, 2, 6), 384: (225.0, 2, 6)}


def surface_pressure(h):
    """[tile][nx, ny] float64 numpy, Pa: ptop + the column sum of delp"""
    n, nz = h.part.nx, h.cfg.npz
    ptop = float(h.sf.ptop) if hasattr(h.sf, "ptop") else None
    out = []
    for i in range(len(h.grids)):
        dp = h.state.delp.sub(i).view[...][:n, :n, :nz].double()
        pe0 = h.state.pe.sub(i).view[...][:n, :n, 0].double()  # the model top pressure as the state holds it
        out.append((dp.sum(dim=2) + (ptop if ptop is not None else pe0)).cpu().numpy())
    return out


def run(nx, days, device, every_h=24.0, nz=79, tracers=1):
    dt, ks, ns = STEPPING[nx]
    h = DycoreHarness(nx, nz=nz, layout=(1, 1), dt_atmos=dt, k_split=ks, n_split=ns, init="baroclinic", n_tracers=tracers, remap=True, device=device)
    n = nx
    area = [np.asarray(g.area[3 : 3 + n, 3 : 3 + n], dtype=np.float64) for g in h.grids]
    lon = [np.asarray(g.lon_agrid[3 : 3 + n, 3 : 3 + n]) if hasattr(g, "lon_agrid") else None for g in h.grids]
    lat = [np.asarray(g.lat_agrid[3 : 3 + n, 3 : 3 + n]) if hasattr(g, "lat_agrid") else None for g in h.grids]
    asum = sum(a.sum() for a in area)
    ps0 = surface_pressure(h)
    m0 = sum((p * a).sum() for p, a in zip(ps0, area))
    rec, snaps = [], {}
    steps_per_out = int(round(every_h * 3600.0 / dt))
    n_steps = int(round(days * 86400.0 / dt))
    t0 = time.time()
    for step in range(1, n_steps + 1):
        h.step()
        if step % steps_per_out == 0 or step == n_steps:
            h.synchronize()
            ps = surface_pressure(h)
            mins = [float(p.min()) for p in ps]
            it = int(np.argmin(mins))
            ij = np.unravel_index(int(np.argmin(ps[it])), ps[it].shape)
            s = h.sanity()
            day = step * dt / 86400.0
            r = {"day": day, "ps_min_hPa": min(mins) / 100.0, "ps_max_hPa": max(float(p.max()) for p in ps) / 100.0,
                 "ps_min_tile": it, "ps_min_lon_deg": float(np.degrees(lon[it][ij])) if lon[it] is not None else None,
                 "ps_min_lat_deg": float(np.degrees(lat[it][ij])) if lat[it] is not None else None,
                 "w_abs_max": max(abs(s["w"][0]), abs(s["w"][1])), "u_min": s["u"][0], "u_max": s["u"][1],
                 "air_mass_drift": float(sum((p * a).sum() for p, a in zip(ps, area)) / m0 - 1.0), "finite": all(v[2] for v in s.values()),
                 "ps_mean_hPa": float(sum((p * a).sum() for p, a in zip(ps, area)) / asum / 100.0), "wall_s": time.time() - t0}
            rec.append(r)
            snaps[round(day, 3)] = ps
            print(f"C{nx} day {day:5.2f}  ps [{r['ps_min_hPa']:8.2f}, {r['ps_max_hPa']:8.2f}] hPa at tile {it} lon {r['ps_min_lon_deg']} lat {r['ps_min_lat_deg']}  "
                  f"|w| {r['w_abs_max']:.3f}  u [{r['u_min']:.1f}, {r['u_max']:.1f}]  mass {r['air_mass_drift']:+.1e}  finite {r['finite']}  ({r['wall_s']:.0f} s)", flush=True)
            if not r["finite"]:
                break
    h.close()
    return {"nx": nx, "nz": nz, "dt_atmos": dt, "k_split": ks, "n_split": ns, "acoustic_dt": dt / ks / ns, "tracers": tracers, "per_day": rec}, snaps, area


def coarsen(ps_f, ratio):
    n = ps_f.shape[0] // ratio
    return ps_f.reshape(n, ratio, n, ratio)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, nargs="+", default=[48, 96])
    ap.add_argument("--days", type=float, default=9.0)
    ap.add_argument("--nz", type=int, default=79)
    ap.add_argument("--out", default=None)
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args()
    res, snaps, areas = {}, {}, {}
    for nx in a.nx:
        res[nx], snaps[nx], areas[nx] = run(nx, a.days, a.device, nz=a.nz)
        if a.out:  # (partial results survive a time limit on the finest run)
            os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
            json.dump({"runs": [res[n_] for n_ in res]}, open(a.out + ".partial", "w"), indent=1)
    out = {"case": "JW2006 baroclinic wave, perturbed, dry, acoustic dynamics + tracer advection + vertical remap of this build (no physics)", "runs": [res[nx] for nx in a.nx]}
    # self-convergence: the coarser run against the finer one averaged onto the coarser cells (area-weighted cell means)
    conv = []
    for c_, f_ in zip(a.nx[:-1], a.nx[1:]):
        if f_ % c_:
            continue
        ratio = f_ // c_
        for day in sorted(set(snaps[c_]) & set(snaps[f_])):
            num = den = 0.0
            for t in range(6):
                af = areas[f_][t]
                pf = (coarsen(snaps[f_][day][t] * af, ratio).sum(axis=(1, 3))) / coarsen(af, ratio).sum(axis=(1, 3))
                d = snaps[c_][day][t] - pf
                num += float((d * d * areas[c_][t]).sum())
                den += float(areas[c_][t].sum())
            conv.append({"coarse": c_, "fine": f_, "day": day, "l2_ps_diff_hPa": (num / den) ** 0.5 / 100.0})
            print(f"C{c_} vs C{f_} day {day:5.2f}: l2(ps) {conv[-1]['l2_ps_diff_hPa']:.4f} hPa", flush=True)
    out["self_convergence"] = conv
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
