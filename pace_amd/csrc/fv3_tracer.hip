// fv3_tracer.hip -- tracer_2d_1l: sub-cycled 2-D advection of the tracers with the mass fluxes / Courant numbers d_sw
// accumulated over the acoustic sub-steps (the "next" row after the acoustic path).
// CPU twin: oracle/fv3_oracle/tracer_2d_1l.py.  Reference operator: pyFV3 TracerAdvection
// [REF examples/notebooks/functions.py:916-951, 1037-1044; tests/savepoint/thresholds/fv_dynamics.yaml:328-360].
//
// The transports are the two-tracer march of fv3_tp4.hip (role TRC: tracers ride on stored mass fluxes, the old air
// mass is the epilogue multiplier, the new one the divisor): tracers go through it in pairs, so the Courant numbers,
// area fluxes, mass fluxes, areas and both air masses are read once per pair.  Around it: three pointwise kernels (area
// fluxes + 1 / n_split scaling, the new air mass of a sub-cycle, the copy of the out-of-place results) and the
// Courant-number bound, reduced per column on the device and over the columns on the host.
#include "fv3_ops.h"

extern "C" int fv3_tracer_2d_1l_cmax(fv3_ctx *c, const fv3_field *cxd_, const fv3_field *cyd_, double *cmax, void *stream) {
  if (!c || !cmax) return FV3_ERR_ARG;
  FV3_FIELD(cx, cxd_) FV3_FIELD(cy, cyd_)
  const Geo g = c->g;
  if (!g.sin_sg5) return fv3_fail(c, FV3_ERR_ARG, "tracer_2d_1l: griddata.sin_sg5 was not given at context creation");
  fv3_stream_t s = (fv3_stream_t)stream;
  Real *col = c->scratch[SC_A];  // plane 0 of a scratch field: per-column maxima
  launch2(c, s, Box{1, g.nx, 1, g.ny, 0, 0}, [=] FV3_HD(int t, int i, int j) {
    const unsigned p = IX(i, j);
    const Real s5 = (g.sin_sg5 + t * g.st2)[p];
    Real m = (Real)0;
    for (int k = 0; k < g.nz; ++k) {
      const long b = t * g.st + k * g.sk;
      const Real v = fv3_max(fabs((cx + b)[p]), fabs((cy + b)[p])) + (Real)1 - s5;
      m = fv3_max(m, v);
    }
    (col + t * g.st)[p] = m;
  });
  // reduce the columns on the host (one plane per sub-domain)
  std::vector<Real> h((size_t)g.sk);
  double mx = 0.0;
  for (int t = 0; t < g.nsub; ++t) {
#ifdef FV3_HOST_EMU
    memcpy(h.data(), col + t * g.st, sizeof(Real) * g.sk);
#else
    if (hipMemcpyAsync(h.data(), col + t * g.st, sizeof(Real) * g.sk, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      return fv3_fail(c, FV3_ERR_HIP, "tracer_2d_1l_cmax: device-to-host copy failed");
#endif
    for (int j = 1; j <= g.ny; ++j)
      for (int i = 1; i <= g.nx; ++i) mx = std::max(mx, (double)h[(size_t)(j + g.o) * g.sj32 + (i + g.o)]);
  }
  *cmax = mx;
  return fv3_post(c, s, "tracer_2d_1l_cmax");
}

extern "C" int fv3_tracer_2d_1l(fv3_ctx *c, int n_tracers, const fv3_field *const *tracers, const fv3_field *dp1_, const fv3_field *mfxd_, const fv3_field *mfyd_,
                                const fv3_field *cxd_, const fv3_field *cyd_, int n_split, int hord, fv3_halo_plan *tracer_halo, void *stream) {
  if (!c || n_tracers < 0 || (n_tracers && !tracers)) return FV3_ERR_ARG;
  FV3_FIELD(dp1, dp1_) FV3_FIELD(mfx, mfxd_) FV3_FIELD(mfy, mfyd_) FV3_FIELD(cx, cxd_) FV3_FIELD(cy, cyd_)
  if (hord != 5 && hord != 6 && hord != 8) return fv3_fail(c, FV3_ERR_UNSUPPORTED, "tracer_2d_1l: hord must be 5, 6 or 8");
  if (n_split < 1) return fv3_fail(c, FV3_ERR_ARG, "tracer_2d_1l: n_split must be >= 1");
  if (n_split > 1 && !tracer_halo) return fv3_fail(c, FV3_ERR_ARG, "tracer_2d_1l: n_split > 1 needs the tracers' halo plan");
  std::vector<Real *> q(n_tracers);
  for (int n = 0; n < n_tracers; ++n) {
    q[n] = fv3_chk(c, tracers[n], "tracer");
    if (!q[n]) return FV3_ERR_ARG;
  }
  const Geo g = c->g;
  fv3_stream_t s = (fv3_stream_t)stream;
  const int nz1 = g.nz - 1;
  const int isd = 1 - g.nh, ied = g.nx + g.nh, jsd = 1 - g.nh, jed = g.ny + g.nh;
  Real *xfx = c->scratch[SC_A], *yfx = c->scratch[SC_B], *dp2 = c->scratch[SC_C], *oa = c->scratch[SC_D], *ob = c->scratch[SC_E];
  const Real frac = (Real)1 / (Real)n_split;
  const bool scale = n_split > 1;
  // area fluxes from the accumulated Courant numbers; 1 / n_split scaling of cx, cy, xfx, yfx, mfx, mfy
  launch3(c, s, Box{isd, ied, jsd, jed, 0, nz1}, [=] FV3_HD(int t, int k, int i, int j) {
    const long b = t * g.st + k * g.sk, m2 = t * g.st2;
    const unsigned p = IX(i, j);
    if (i >= 1 && i <= g.nx + 1) {
      const unsigned pm = IX(i - 1, j);
      const Real cv = (cx + b)[p];
      Real x = cv > (Real)0 ? cv * (g.dxa + m2)[pm] * (g.dy + m2)[p] * (g.sin_sg3 + m2)[pm] : cv * (g.dxa + m2)[p] * (g.dy + m2)[p] * (g.sin_sg1 + m2)[p];
      if (scale) {
        x = x * frac;
        (cx + b)[p] = cv * frac;
        if (j >= 1 && j <= g.ny) (mfx + b)[p] = (mfx + b)[p] * frac;
      }
      (xfx + b)[p] = x;
    }
    if (j >= 1 && j <= g.ny + 1) {
      const unsigned pm = IX(i, j - 1);
      const Real cv = (cy + b)[p];
      Real y = cv > (Real)0 ? cv * (g.dya + m2)[pm] * (g.dx + m2)[p] * (g.sin_sg4 + m2)[pm] : cv * (g.dya + m2)[p] * (g.dx + m2)[p] * (g.sin_sg2 + m2)[p];
      if (scale) {
        y = y * frac;
        (cy + b)[p] = cv * frac;
        if (i >= 1 && i <= g.nx) (mfy + b)[p] = (mfy + b)[p] * frac;
      }
      (yfx + b)[p] = y;
    }
  });
  Deln off;
  memset(&off, 0, sizeof(off));
  for (int it = 0; it < n_split; ++it) {
    launch3<4>(c, s, Box{1, g.nx, 1, g.ny, 0, nz1}, [=] FV3_HD(int t, int k, int i, int j) {
      const long b = t * g.st + k * g.sk;
      const unsigned p = IX(i, j);
      (dp2 + b)[p] = (dp1 + b)[p] + ((mfx + b)[p] - (mfx + b)[IX(i + 1, j)] + (mfy + b)[p] - (mfy + b)[IX(i, j + 1)]) * g.rarea[t * g.st2 + p];
    });
    for (int n = 0; n < n_tracers; n += 2) {
      Real *qa = q[n], *qb = n + 1 < n_tracers ? q[n + 1] : q[n];
      DswScalars a{};
      a.delp = dp1;
      a.o_delp = dp2;
      a.q_con = qa;
      a.pt = qb;
      a.o_q_con = oa;
      a.o_pt = ob;
      a.crx = cx;
      a.cry = cy;
      a.xfx = xfx;
      a.yfx = yfx;
      a.fx = mfx;
      a.fy = mfy;
      a.hord_dp = a.hord_vt = a.hord_tm = hord;
      a.dn_vt = off;
      a.dn_t = off;
      a.dt = (Real)0;
      tracer_pair_stream(c, s, a);
      const bool two = n + 1 < n_tracers;
      launch3<4>(c, s, Box{1, g.nx, 1, g.ny, 0, nz1}, [=] FV3_HD(int t, int k, int i, int j) {
        const long p = t * g.st + k * g.sk + IX(i, j);
        qa[p] = oa[p];
        if (two) qb[p] = ob[p];
      });
    }
    if (it < n_split - 1) {
      launch3<4>(c, s, Box{1, g.nx, 1, g.ny, 0, nz1}, [=] FV3_HD(int t, int k, int i, int j) {
        const long p = t * g.st + k * g.sk + IX(i, j);
        dp1[p] = dp2[p];
      });
      int st = fv3_halo_plan_start(c, tracer_halo, stream);
      if (st == FV3_OK) st = fv3_halo_plan_wait(c, tracer_halo, stream);
      if (st != FV3_OK) return st;
    }
  }
  return fv3_post(c, s, "tracer_2d_1l");
}
