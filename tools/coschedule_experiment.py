#!/usr/bin/env python3
"""Would update_dz_d beside d_sw pay?  update_dz_d needs only d_sw's first kernel (fxadv: the Courant numbers / area fluxes) and the
interface heights, so the sequencer could run it on a second stream beside d_sw's transports -- if two latency-bound marching
operators share the GPU better than they queue.  This experiment answers that before anyone untangles the scratch arrays the two
operators share: two contexts (each half of the C768 problem: 12 sub-domains), d_sw on one, update_dz_d on the other, timed alone and
together on two streams.
    python tools/coschedule_experiment.py [--reps 4]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--nx", type=int, default=768)
    a = ap.parse_args()
    from pace_amd.harness import DycoreHarness

    hs = [DycoreHarness(a.nx, nz=79, layout=(2, 2), dt_atmos=225.0, k_split=2, n_split=6, world_size=2, proc=0, backend="hip:gfx950", device="cuda:0", loopback=True)
          for _ in range(2)]
    for h in hs:
        h.dyn(h.state, 112.5, n_map=1)  # (fills every work array with a sub-step's values)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for h, s in zip(hs, streams):
        h.sf.stream = s.cuda_stream
    dt = 112.5 / 6

    def d_sw(h):
        d, st = h.dyn, h.state
        d.dgrid_shallow_water_lagrangian_dynamics(d._vt_scratch, st.delp, st.pt, st.u, st.v, st.w, st.uc, st.vc, st.ua, st.va, d._divgd, st.mfxd, st.mfyd, st.cxd, st.cyd,
                                                  d._crx, d._cry, d._xfx, d._yfx, st.q_con, d._zh, d._heat_source, st.diss_estd, dt)

    def dz_d(h):
        d = h.dyn
        d.update_height_on_d_grid(d._zs, d._zh, d._crx, d._cry, d._xfx, d._yfx, d._wsd, dt)

    def timed(fns):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in streams:
            s.wait_event(e0)
        for _ in range(a.reps):
            for f, h in fns:
                f(h)
        cur = torch.cuda.current_stream()
        for s in streams:
            cur.wait_stream(s)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    timed([(d_sw, hs[0]), (dz_d, hs[1])])  # warm-up
    t_d = timed([(d_sw, hs[0])])
    t_u = timed([(dz_d, hs[1])])
    t_both = timed([(d_sw, hs[0]), (dz_d, hs[1])])
    print(f"d_sw alone {t_d:.2f} ms, update_dz_d alone {t_u:.2f} ms, sum {t_d + t_u:.2f} ms; together on two streams {t_both:.2f} ms "
          f"({100 * (t_d + t_u - t_both) / (t_d + t_u):.1f} % of the sum saved; {12} sub-domains of 384^2 per context)")


if __name__ == "__main__":
    main()
