// fv3_tp2d.hip -- Lin-Rood 2-D flux-form transport (fv_tp_2d) and the del-n damping fluxes.
// CPU twin: oracle/fv3_oracle/fvtp2d.py.  [SURVEY A.4, A.4.3; reference operator
// FiniteVolumeTransport, REF examples/notebooks/functions.py:935-951]
//
// Baseline structure (round 1): 3 launches for the transport proper
//   (1) inner fluxes fy2 = yppm(q), fx2 = xppm(q)           -- q read through the corner remap
//   (2) q_i, q_j (flux-updated in the cross direction)
//   (3) outer fluxes, averaged with the inner ones and scaled by the area / mass flux
// plus the del-n chain when damping is on.  HBM traffic per level is dominated by the four
// intermediates (fy2, fx2, q_i, q_j); fusing them through LDS is the next step.
#include "fv3_ops.h"
#include "fv3_ppm.h"

void del6_vt_flux(fv3_ctx *c, fv3_stream_t s, const Real *q, Real *d2, Real *fx2, Real *fy2, const Deln &dn, bool q_raw, int k0, int k1) {
  const Geo g = c->g;
  const int nm = dn.nord_max;
  const Deln d = dn;
  // d2 = damp * q on (is-1-nord .. ie+1+nord)^2
  launch3(c, s, Box{-nm, g.nx + 1 + nm, -nm, g.ny + 1 + nm, k0, k1}, [=] FV3_HD(int t, int k, int i, int j) {
    if (!deln_on(d, k)) return;
    const int n = deln_nord(d, k);
    if (i < -n || i > g.nx + 1 + n || j < -n || j > g.ny + 1 + n) return;
    const long p = t * g.st + k * g.sk + IX(i, j);
    d2[p] = q_raw ? q[p] : deln_damp(d, k) * q[p];
  });
  // first fluxes (copy_corners only when nord > 0)
  launch3(c, s, Box{1 - nm, g.nx + nm + 1, 1 - nm, g.ny + nm + 1, k0, k1}, [=] FV3_HD(int t, int k, int i, int j) {
    if (!deln_on(d, k)) return;
    const int n = deln_nord(d, k);
    const int fl = g.flags[t];
    const long b = t * g.st + k * g.sk;
    const long m2 = t * g.st2;
    const Real *dd = d2 + b;
    if (i >= 1 - n && i <= g.nx + n + 1 && j >= 1 - n && j <= g.ny + n) {
      const Real a = n > 0 ? cc<1>(dd, g, fl, i - 1, j) : dd[IX(i - 1, j)];
      const Real e = n > 0 ? cc<1>(dd, g, fl, i, j) : dd[IX(i, j)];
      fx2[b + IX(i, j)] = g.del6_v[m2 + IX(i, j)] * (a - e);
    }
    if (i >= 1 - n && i <= g.nx + n && j >= 1 - n && j <= g.ny + n + 1) {
      const Real a = n > 0 ? cc<2>(dd, g, fl, i, j - 1) : dd[IX(i, j - 1)];
      const Real e = n > 0 ? cc<2>(dd, g, fl, i, j) : dd[IX(i, j)];
      fy2[b + IX(i, j)] = g.del6_u[m2 + IX(i, j)] * (a - e);
    }
  });
  for (int n = 1; n <= nm; ++n) {
    launch3(c, s, Box{-(nm - n), g.nx + 1 + (nm - n), -(nm - n), g.ny + 1 + (nm - n), k0, k1}, [=] FV3_HD(int t, int k, int i, int j) {
      if (!deln_on(d, k)) return;
      const int nord = deln_nord(d, k);
      if (n > nord) return;
      const int nt = nord - n;
      if (i < -nt || i > g.nx + 1 + nt || j < -nt || j > g.ny + 1 + nt) return;
      const long b = t * g.st + k * g.sk;
      const long p = IX(i, j);
      d2[b + p] = (fx2[b + p] - fx2[b + IX(i + 1, j)] + fy2[b + p] - fy2[b + IX(i, j + 1)]) * g.rarea[t * g.st2 + p];
    });
    launch3(c, s, Box{1 - (nm - n), g.nx + (nm - n) + 1, 1 - (nm - n), g.ny + (nm - n) + 1, k0, k1}, [=] FV3_HD(int t, int k, int i, int j) {
      if (!deln_on(d, k)) return;
      const int nord = deln_nord(d, k);
      if (n > nord) return;
      const int nt = nord - n;
      const int fl = g.flags[t];
      const long b = t * g.st + k * g.sk;
      const long m2 = t * g.st2;
      const Real *dd = d2 + b;
      if (i >= 1 - nt && i <= g.nx + nt + 1 && j >= 1 - nt && j <= g.ny + nt)
        fx2[b + IX(i, j)] = g.del6_v[m2 + IX(i, j)] * (cc<1>(dd, g, fl, i, j) - cc<1>(dd, g, fl, i - 1, j));
      if (i >= 1 - nt && i <= g.nx + nt && j >= 1 - nt && j <= g.ny + nt + 1)
        fy2[b + IX(i, j)] = g.del6_u[m2 + IX(i, j)] * (cc<2>(dd, g, fl, i, j) - cc<2>(dd, g, fl, i, j - 1));
    });
  }
}

void tp2d(fv3_ctx *c, fv3_stream_t s, const Real *q, const Real *crx, const Real *cry, const Real *xfx, const Real *yfx, Real *fx, Real *fy,
          const Real *mfx, const Real *mfy, const Real *mass, int hord, const Deln *dn, int k0, int k1) {
  const Geo g = c->g;
  Real *fy2 = c->scratch[SC_TP_FY2], *fx2 = c->scratch[SC_TP_FX2], *q_i = c->scratch[SC_TP_QI], *q_j = c->scratch[SC_TP_QJ];
  const int isd = 1 - g.nh, ied = g.nx + g.nh, jsd = 1 - g.nh, jed = g.ny + g.nh;
  // (1) inner fluxes
  launch3(c, s, Box{isd, ied, jsd, jed, k0, k1}, [=] FV3_HD(int t, int k, int i, int j) {
    const int fl = g.flags[t];
    const long b = t * g.st + k * g.sk;
    const long m2 = t * g.st2;
    const Real *qq = q + b;
    if (j >= 1 && j <= g.ny + 1) {  // fy2 on i = isd..ied
      auto Q = [&](int s_) { return cc<2>(qq, g, fl, i, s_); };
      auto M = [&](int s_) { return g.dya[m2 + IX(i, s_)]; };
      fy2[b + IX(i, j)] = ppm_flux(Q, M, cry[b + IX(i, j)], j, (fl & FV3_S) != 0, (fl & FV3_N) != 0, g.npy, hord);
    }
    if (i >= 1 && i <= g.nx + 1) {  // fx2 on j = jsd..jed
      auto Q = [&](int s_) { return cc<1>(qq, g, fl, s_, j); };
      auto M = [&](int s_) { return g.dxa[m2 + IX(s_, j)]; };
      fx2[b + IX(i, j)] = ppm_flux(Q, M, crx[b + IX(i, j)], i, (fl & FV3_W) != 0, (fl & FV3_E) != 0, g.npx, hord);
    }
  });
  // (2) cross-direction updates
  launch3(c, s, Box{isd, ied, jsd, jed, k0, k1}, [=] FV3_HD(int t, int k, int i, int j) {
    const long b = t * g.st + k * g.sk;
    const long m2 = t * g.st2;
    const long p = IX(i, j);
    const Real qa = q[b + p] * g.area[m2 + p];
    if (j >= 1 && j <= g.ny) {
      const long pn = IX(i, j + 1);
      const Real y0 = yfx[b + p], y1 = yfx[b + pn];
      const Real ra_y = g.area[m2 + p] + y0 - y1;
      q_i[b + p] = (qa + y0 * fy2[b + p] - y1 * fy2[b + pn]) / ra_y;
    }
    if (i >= 1 && i <= g.nx) {
      const long pe = IX(i + 1, j);
      const Real x0 = xfx[b + p], x1 = xfx[b + pe];
      const Real ra_x = g.area[m2 + p] + x0 - x1;
      q_j[b + p] = (qa + x0 * fx2[b + p] - x1 * fx2[b + pe]) / ra_x;
    }
  });
  // del-n damping fluxes (computed before the outer stage so they can be added in it)
  Real *dfx = c->scratch[SC_DN_FX], *dfy = c->scratch[SC_DN_FY];
  const bool damped = dn != nullptr;
  Deln d;
  memset(&d, 0, sizeof(d));
  if (damped) {
    d = *dn;
    del6_vt_flux(c, s, q, c->scratch[SC_DN_D2], dfx, dfy, d, mass != nullptr, k0, k1);
  }
  // (3) outer fluxes
  launch3(c, s, Box{1, g.nx + 1, 1, g.ny + 1, k0, k1}, [=] FV3_HD(int t, int k, int i, int j) {
    const int fl = g.flags[t];
    const long b = t * g.st + k * g.sk;
    const long m2 = t * g.st2;
    const long p = IX(i, j);
    const bool on = damped && deln_on(d, k);
    if (j <= g.ny) {
      const Real *qq = q_i + b;
      auto Q = [&](int s_) { return qq[IX(s_, j)]; };
      auto M = [&](int s_) { return g.dxa[m2 + IX(s_, j)]; };
      const Real f = ppm_flux(Q, M, crx[b + p], i, (fl & FV3_W) != 0, (fl & FV3_E) != 0, g.npx, hord);
      Real v = (Real)0.5 * (f + fx2[b + p]) * (mfx ? mfx[b + p] : xfx[b + p]);
      if (on) {
        if (mass)
          v = v + (Real)0.5 * deln_damp(d, k) * (mass[b + IX(i - 1, j)] + mass[b + p]) * dfx[b + p];
        else
          v = v + dfx[b + p];
      }
      fx[b + p] = v;
    }
    if (i <= g.nx) {
      const Real *qq = q_j + b;
      auto Q = [&](int s_) { return qq[IX(i, s_)]; };
      auto M = [&](int s_) { return g.dya[m2 + IX(i, s_)]; };
      const Real f = ppm_flux(Q, M, cry[b + p], j, (fl & FV3_S) != 0, (fl & FV3_N) != 0, g.npy, hord);
      Real v = (Real)0.5 * (f + fy2[b + p]) * (mfy ? mfy[b + p] : yfx[b + p]);
      if (on) {
        if (mass)
          v = v + (Real)0.5 * deln_damp(d, k) * (mass[b + IX(i, j - 1)] + mass[b + p]) * dfy[b + p];
        else
          v = v + dfy[b + p];
      }
      fy[b + p] = v;
    }
  });
}

extern "C" int fv3_fv_tp_2d(fv3_ctx *c, const fv3_field *q_, const fv3_field *crx_, const fv3_field *cry_, const fv3_field *xfx_, const fv3_field *yfx_,
                            const fv3_field *fx_, const fv3_field *fy_, const fv3_field *mfx_, const fv3_field *mfy_, const fv3_field *mass_, int hord,
                            int nord, double damp_c, void *stream) {
  if (!c) return FV3_ERR_ARG;
  FV3_FIELD(q, q_) FV3_FIELD(crx, crx_) FV3_FIELD(cry, cry_) FV3_FIELD(xfx, xfx_) FV3_FIELD(yfx, yfx_) FV3_FIELD(fx, fx_) FV3_FIELD(fy, fy_)
  Real *mfx = nullptr, *mfy = nullptr, *mass = nullptr;
  if (mfx_ || mfy_) {
    if (!(mfx_ && mfy_)) return fv3_fail(c, FV3_ERR_ARG, "mfx and mfy must be given together");
    mfx = fv3_chk(c, mfx_, "mfx");
    mfy = fv3_chk(c, mfy_, "mfy");
    if (!mfx || !mfy) return FV3_ERR_ARG;
  }
  if (mass_) {
    mass = fv3_chk(c, mass_, "mass");
    if (!mass) return FV3_ERR_ARG;
  }
  if (hord != 5 && hord != 6) return fv3_fail(c, FV3_ERR_UNSUPPORTED, "hord must be 5 or 6");
  if (nord > 2) return fv3_fail(c, FV3_ERR_UNSUPPORTED, "fv_tp_2d damping order must be <= 2 (halo of 3)");
  Deln d;
  memset(&d, 0, sizeof(d));
  const bool damped = nord >= 0 && damp_c > 1.0e-4 && (mfx == nullptr || mass != nullptr);
  if (damped) {
    d.nord_u = nord;
    d.damp_u = (Real)std::pow(damp_c * (double)c->g.da_min, (double)(nord + 1));
    d.on_u = true;
    d.nord_max = nord;
  }
  // the reference leaves q's cube-corner halo overwritten; here q is read-only (corner reads are remapped)
  tp2d(c, (fv3_stream_t)stream, q, crx, cry, xfx, yfx, fx, fy, mfx, mfy, mass, hord, damped ? &d : nullptr, 0, c->g.nz - 1);
  return fv3_post(c, (fv3_stream_t)stream, "fv_tp_2d");
}
