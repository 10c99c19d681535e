#!/usr/bin/env python3
"""Does running an issue-bound marching kernel (fv_tp_2d) concurrently with bandwidth-bound stage
kernels (c_sw) on two HIP streams beat running them back to back?  C768-sized fields, one GPU."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pace_amd.harness import CONFIGS, DycoreHarness  # noqa: E402


def main():
    kw = dict(CONFIGS["c768"])
    h = DycoreHarness(world_size=1, proc=0, device="cuda:0", **kw)
    sf, st, dyn = h.sf, h.state, h.dyn
    h.step()  # fills crx/cry/xfx/yfx etc.
    torch.cuda.synchronize()
    qf = sf.quantity_factory
    fx, fy = qf.zeros(("x", "y", "z")), qf.zeros(("x", "y", "z"))
    cs = dyn.cgrid_shallow_water_lagrangian_dynamics
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def tp(stream):
        sf.lib.fv3_fv_tp_2d(sf.ctx, st.pt.fref, dyn._crx.fref, dyn._cry.fref, dyn._xfx.fref, dyn._yfx.fref, fx.fref, fy.fref, None, None, None, 6, -1, 0.0, stream.cuda_stream)

    def csw(stream):
        sf.lib.fv3_c_sw(sf.ctx, st.delp.fref, st.pt.fref, st.u.fref, st.v.fref, st.w.fref, st.uc.fref, st.vc.fref, st.ua.fref, st.va.fref, dyn._ut.fref, dyn._vt.fref,
                        dyn._divgd.fref, st.omga.fref, cs.delpc.fref, cs.ptc.fref, 9.375, stream.cuda_stream)

    def timed(fn, n=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    t_tp = timed(lambda: [tp(s1) for _ in range(4)])
    t_cs = timed(lambda: csw(s1))
    t_seq = timed(lambda: ([tp(s1) for _ in range(4)], csw(s1)))
    t_par = timed(lambda: ([tp(s1) for _ in range(4)], csw(s2)))
    print(f"4 x fv_tp_2d {t_tp:.2f} ms, c_sw {t_cs:.2f} ms, back to back {t_seq:.2f} ms, two streams {t_par:.2f} ms")


if __name__ == "__main__":
    main()
