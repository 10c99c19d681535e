// fv3_remap.hip -- Lagrangian-to-Eulerian vertical remapping (the step after the acoustic loop and the tracer
// advection in DynamicalCore.step_dynamics).  CPU twin: oracle/remap_oracle.c.  Reference operator: pyFV3 LagrangianToEulerian
// [REF driver/pace/driver/driver.py:494-504, 639-644; savepoint Remapping, tests/savepoint/thresholds/fv_dynamics.yaml:227-326;
// kord_tm -9, kord_mt / kord_tr / kord_wz 9, consv_te 0: driver/examples/configs/baroclinic_c12.yaml:45,65-68].
// Restated from fv_mapz.F90 (map_scalar / map1_ppm with cs_profile, kord 9); configuration of the reference configs:
// non-hydrostatic, T_v remapped in log(p), moist-cappa pkz with the cappa field given, no energy fixer, no saturation
// adjustment, no fillz, omga untouched.
//
// A thread owns a column (i fastest: every level access of a wave is one coalesced row).  What a column needs beyond
// O(1) registers are the edge values of the parabolas (a tridiagonal solve in k): two scratch fields (the elimination factors
// and the edge values); the limited parabola of a source layer is a function of the edge values and five neighbouring means
// only, so it is rebuilt where the conservative integration needs it instead of being stored (the Fortran keeps a4(4, km)
// per column).  The Lagrangian interface pressures are the pe / peln fields the last acoustic sub-step left (riem_solver3
// with last_call, edge_pe for the ring the D-grid winds average over); the Eulerian ones are ak + bk * ps on the fly.
#include "fv3_ops.h"

namespace {

#define RM_R3 ((Real)(1.0 / 3.0))
#define RM_R23 ((Real)(2.0 / 3.0))
#define RM_R12 ((Real)(1.0 / 12.0))

struct Prof {
  Real a1, a2, a3, a4;  // mean, upper edge, lower edge, curvature
};

FV3_HD inline Real rmin3(Real a, Real b, Real c) { return fv3_min(a, fv3_min(b, c)); }
FV3_HD inline Real rmax3(Real a, Real b, Real c) { return fv3_max(a, fv3_max(b, c)); }

FV3_HD inline void cs_limiters(bool extm, Prof &p, int iv) {
  if (iv == 0) {
    if (p.a1 <= (Real)0) {
      p.a2 = p.a1, p.a3 = p.a1, p.a4 = (Real)0;
    } else if (fabs(p.a3 - p.a2) < -p.a4) {
      if (p.a1 + (Real)0.25 * (p.a3 - p.a2) * (p.a3 - p.a2) / p.a4 + p.a4 * RM_R12 < (Real)0) {
        if (p.a1 < p.a3 && p.a1 < p.a2) {
          p.a3 = p.a1, p.a2 = p.a1, p.a4 = (Real)0;
        } else if (p.a3 > p.a2) {
          p.a4 = (Real)3.0 * (p.a2 - p.a1);
          p.a3 = p.a2 - p.a4;
        } else {
          p.a4 = (Real)3.0 * (p.a3 - p.a1);
          p.a2 = p.a3 - p.a4;
        }
      }
    }
  } else {
    const bool flat = iv == 1 ? ((p.a1 - p.a2) * (p.a1 - p.a3) >= (Real)0) : extm;
    if (flat) {
      p.a2 = p.a1, p.a3 = p.a1, p.a4 = (Real)0;
    } else {
      const Real da1 = p.a3 - p.a2, da2 = da1 * da1, a6da = p.a4 * da1;
      if (a6da < -da2) {
        p.a4 = (Real)3.0 * (p.a2 - p.a1);
        p.a3 = p.a2 - p.a4;
      } else if (a6da > da2) {
        p.a4 = (Real)3.0 * (p.a3 - p.a1);
        p.a2 = p.a3 - p.a4;
      }
    }
  }
}

// One column.  PE1(l): source interface coordinate (l = 0 .. km), PE2(k): target one; Q1(k): source layer mean;
// GAM / QE: per-column scratch accessors (km + 1 levels); OUT(k, value): receives the remapped means.
template <class PE1, class PE2, class Q1F, class GAMF, class QEF, class OUTF>
FV3_HD inline void remap_col(int km, PE1 pe1, PE2 pe2, Q1F Q1, GAMF GAM, QEF QE, OUTF OUT, int iv, Real qs, bool use_qmin, Real qmin) {
  auto DP = [&](int k) { return pe1(k + 1) - pe1(k); };
  // ---- edge values: tridiagonal solve (cs_profile)
  if (iv == -2) {
    GAM(1) = (Real)0.5;
    Real qp = (Real)1.5 * Q1(0);
    QE(0) = qp;
    for (int k = 1; k < km - 1; ++k) {
      const Real grat = DP(k - 1) / DP(k);
      const Real bet = (Real)2.0 + grat + grat - GAM(k);
      qp = ((Real)3.0 * (Q1(k - 1) + Q1(k)) - qp) / bet;
      QE(k) = qp;
      GAM(k + 1) = grat / bet;
    }
    const Real grat = DP(km - 2) / DP(km - 1);
    qp = ((Real)3.0 * (Q1(km - 2) + Q1(km - 1)) - grat * qs - qp) / ((Real)2.0 + grat + grat - GAM(km - 1));
    QE(km - 1) = qp;
    QE(km) = qs;
    Real qn = qs;
    qn = qp;  // q(km-1) needs no back substitution of its own: it was solved against the boundary value
    for (int k = km - 2; k >= 0; --k) {
      qn = QE(k) - GAM(k + 1) * qn;
      QE(k) = qn;
    }
  } else {
    const Real grat = DP(1) / DP(0);
    Real bet = grat * (grat + (Real)0.5);
    Real qp = ((grat + grat) * (grat + (Real)1.0) * Q1(0) + Q1(1)) / bet;
    QE(0) = qp;
    Real gp = ((Real)1.0 + grat * (grat + (Real)1.5)) / bet;
    GAM(0) = gp;
    Real d4 = (Real)0;
    for (int k = 1; k < km; ++k) {
      d4 = DP(k - 1) / DP(k);
      bet = (Real)2.0 + d4 + d4 - gp;
      qp = ((Real)3.0 * (Q1(k - 1) + d4 * Q1(k)) - qp) / bet;
      QE(k) = qp;
      gp = d4 / bet;
      GAM(k) = gp;
    }
    const Real a_bot = (Real)1.0 + d4 * (d4 + (Real)1.5);
    Real qn = ((Real)2.0 * d4 * (d4 + (Real)1.0) * Q1(km - 1) + Q1(km - 2) - a_bot * qp) / (d4 * (d4 + (Real)0.5) - a_bot * gp);
    QE(km) = qn;
    for (int k = km - 1; k >= 0; --k) {
      qn = QE(k) - GAM(k) * qn;
      QE(k) = qn;
    }
  }
  // ---- large-scale constraints on the edge values
  auto GD = [&](int k) { return Q1(k) - Q1(k - 1); };  // difference of the means across interface k (1 .. km-1)
  {
    Real q = QE(1);
    q = fv3_min(q, fv3_max(Q1(0), Q1(1)));
    q = fv3_max(q, fv3_min(Q1(0), Q1(1)));
    QE(1) = q;
  }
  for (int k = 2; k < km - 1; ++k) {
    Real q = QE(k);
    const Real lo = fv3_min(Q1(k - 1), Q1(k)), hi = fv3_max(Q1(k - 1), Q1(k));
    const Real gm = GD(k - 1), gp = GD(k + 1);
    if (gm * gp > (Real)0) {
      q = fv3_min(q, hi);
      q = fv3_max(q, lo);
    } else if (gm > (Real)0) {
      q = fv3_max(q, lo);
    } else {
      q = fv3_min(q, hi);
      if (iv == 0) q = fv3_max((Real)0, q);
    }
    QE(k) = q;
  }
  {
    Real q = QE(km - 1);
    q = fv3_min(q, fv3_max(Q1(km - 2), Q1(km - 1)));
    q = fv3_max(q, fv3_min(Q1(km - 2), Q1(km - 1)));
    QE(km - 1) = q;
  }
  // ---- the limited parabola of source layer l, rebuilt on demand
  auto EXTM = [&](int m) { return GD(m) * GD(m + 1) < (Real)0; };  // 1 <= m <= km - 2
  auto profile = [&](int l) {
    Prof p;
    p.a1 = Q1(l);
    p.a2 = QE(l);
    p.a3 = QE(l + 1);
    if (l == 0) {
      if (iv == 0)
        p.a2 = fv3_max((Real)0, p.a2);
      else if (iv == -1 && p.a2 * p.a1 <= (Real)0)
        p.a2 = (Real)0;
      p.a4 = (Real)3.0 * ((Real)2.0 * p.a1 - (p.a2 + p.a3));
      cs_limiters(false, p, 1);
    } else if (l == km - 1) {
      if (iv == 0)
        p.a3 = fv3_max((Real)0, p.a3);
      else if (iv == -1 && p.a3 * p.a1 <= (Real)0)
        p.a3 = (Real)0;
      p.a4 = (Real)3.0 * ((Real)2.0 * p.a1 - (p.a2 + p.a3));
      cs_limiters(false, p, 1);
    } else if (l == 1 || l == km - 2) {
      p.a4 = (Real)3.0 * ((Real)2.0 * p.a1 - (p.a2 + p.a3));
      cs_limiters(EXTM(l), p, 2);
    } else {
      const bool e = EXTM(l);
      if ((e && EXTM(l - 1)) || (e && EXTM(l + 1)) || (use_qmin && e && p.a1 < qmin)) {
        p.a2 = p.a1, p.a3 = p.a1, p.a4 = (Real)0;
      } else {
        p.a4 = (Real)6.0 * p.a1 - (Real)3.0 * (p.a2 + p.a3);
        if (fabs(p.a4) > fabs(p.a2 - p.a3)) {
          const Real pmp_1 = p.a1 - (Real)2.0 * GD(l + 1), lac_1 = pmp_1 + (Real)1.5 * GD(l + 2);
          p.a2 = fv3_min(fv3_max(p.a2, rmin3(p.a1, pmp_1, lac_1)), rmax3(p.a1, pmp_1, lac_1));
          const Real pmp_2 = p.a1 + (Real)2.0 * GD(l), lac_2 = pmp_2 - (Real)1.5 * GD(l - 1);
          p.a3 = fv3_min(fv3_max(p.a3, rmin3(p.a1, pmp_2, lac_2)), rmax3(p.a1, pmp_2, lac_2));
          p.a4 = (Real)6.0 * p.a1 - (Real)3.0 * (p.a2 + p.a3);
        }
      }
      if (iv == 0) cs_limiters(e, p, 0);
    }
    return p;
  };
  // ---- conservative integration over the target layers
  int k0 = 0;
  Real t_lo = pe2(0);
  for (int k = 0; k < km; ++k) {
    const Real t_hi = pe2(k + 1);
    Real val = Q1(k);
    for (int l = k0; l < km; ++l) {
      const Real s_lo = pe1(l), s_hi = pe1(l + 1);
      if (t_lo >= s_lo && t_lo <= s_hi) {
        const Real dl = s_hi - s_lo;
        const Real pl = (t_lo - s_lo) / dl;
        const Prof a = profile(l);
        if (t_hi <= s_hi) {
          const Real pr = (t_hi - s_lo) / dl;
          val = a.a2 + (Real)0.5 * (a.a4 + a.a3 - a.a2) * (pr + pl) - a.a4 * RM_R3 * (pr * (pr + pl) + pl * pl);
          k0 = l;
        } else {
          Real qsum = (s_hi - t_lo) * (a.a2 + (Real)0.5 * (a.a4 + a.a3 - a.a2) * ((Real)1.0 + pl) - a.a4 * (RM_R3 * ((Real)1.0 + pl * ((Real)1.0 + pl))));
          for (int m = l + 1; m < km; ++m) {
            const Real m_lo = pe1(m), m_hi = pe1(m + 1);
            if (t_hi > m_hi) {
              qsum = qsum + (m_hi - m_lo) * Q1(m);
            } else {
              const Real dp = t_hi - m_lo, esl = dp / (m_hi - m_lo);
              const Prof b = profile(m);
              qsum = qsum + dp * (b.a2 + (Real)0.5 * esl * (b.a3 - b.a2 + b.a4 * ((Real)1.0 - RM_R23 * esl)));
              k0 = m;
              break;
            }
          }
          val = qsum / (t_hi - t_lo);
        }
        break;
      }
    }
    OUT(k, val);
    t_lo = t_hi;
  }
}

}  // namespace

// The scratch accessors below index [level][column] fields: K(arr, k).
#define RK(arr, k) ((arr) + tb + (long)(k)*g.sk)[pix]

extern "C" int fv3_remap(fv3_ctx *c, int n_tracers, const fv3_field *const *tracers, const fv3_field *pt_, const fv3_field *delp_, const fv3_field *delz_,
                         const fv3_field *peln_, const fv3_field *pe_, const fv3_field *pk_, const fv3_field *pkz_, const fv3_field *u_, const fv3_field *v_,
                         const fv3_field *w_, const fv3_field *cappa_, const fv3_field *ps_, const fv3_field *wsd_, void *stream) {
  if (!c || n_tracers < 0 || (n_tracers && !tracers)) return FV3_ERR_ARG;
  FV3_FIELD(pt, pt_) FV3_FIELD(delp, delp_) FV3_FIELD(delz, delz_) FV3_FIELD(peln, peln_) FV3_FIELD(pe, pe_) FV3_FIELD(pk, pk_) FV3_FIELD(pkz, pkz_)
  FV3_FIELD(u, u_) FV3_FIELD(v, v_) FV3_FIELD(w, w_) FV3_FIELD(cappa, cappa_) FV3_FIELD2D(ps, ps_) FV3_FIELD2D(wsd, wsd_)
  const Geo g = c->g;
  if (g.nz < 5) return fv3_fail(c, FV3_ERR_UNSUPPORTED, "remap: needs at least 5 levels (two monotone layers at either end + an interior)");
  std::vector<Real *> q(n_tracers);
  for (int n = 0; n < n_tracers; ++n) {
    q[n] = fv3_chk(c, tracers[n], "tracer");
    if (!q[n]) return FV3_ERR_ARG;
  }
  fv3_stream_t s = (fv3_stream_t)stream;
  const int km = g.nz;
  Real *GAM = c->scratch[SC_A], *QE = c->scratch[SC_B], *Q2 = c->scratch[SC_C], *TV = c->scratch[SC_D];
  std::vector<Real> akh(c->ak.begin(), c->ak.end()), bkh(c->bk.begin(), c->bk.end());
  // ak / bk as device tables (uploaded once per context)
  if (!c->ak_dev) {
    c->ak_dev = (Real *)fv3_dev_alloc(c, sizeof(Real) * (km + 1));
    c->bk_dev = (Real *)fv3_dev_alloc(c, sizeof(Real) * (km + 1));
    if (!c->ak_dev || !c->bk_dev) return fv3_fail(c, FV3_ERR_NOMEM, "remap: table allocation failed");
    fv3_h2d(c->ak_dev, akh.data(), sizeof(Real) * (km + 1));
    fv3_h2d(c->bk_dev, bkh.data(), sizeof(Real) * (km + 1));
  }
  const Real *ak = c->ak_dev, *bk = c->bk_dev;
  const Real ptop = (Real)c->ptop, akap = (Real)(c->cst.rdgas / c->cst.cp_air), rrg = (Real)(-c->cst.rdgas / c->cst.grav), t_min = (Real)184.0;
  const Box cells{1, g.nx, 1, g.ny, 0, 0};
  // ---- T_v of the Lagrangian layers (kord_tm < 0: the temperature is what is remapped)
  launch3<4>(c, s, Box{1, g.nx, 1, g.ny, 0, km - 1}, [=] FV3_HD(int t, int k, int i, int j) {
    const long p = t * g.st + k * g.sk + IX(i, j);
    const Real ptv = pt[p], cp = cappa[p];
    TV[p] = ptv * exp(cp / ((Real)1.0 - cp) * log(rrg * delp[p] / delz[p] * ptv));
  });
  // ---- one column kernel per remapped field.  mode 0: pressure coordinate, 1: log-pressure
  auto scalar = [&](Real *field, const Real *src, int mode, int iv, const Real *qs2d, bool use_qmin, Real qmin, int post) {
    launch2(c, s, cells, [=] FV3_HD(int t, int i, int j) {
      const long tb = t * g.st;
      const unsigned pix = IX(i, j);
      const Real psv = RK(pe, km);
      const Real *coord = mode == 1 ? peln : pe;
      auto PE1 = [&](int l) -> Real { return RK(coord, l); };
      auto PE2 = [&](int k) -> Real {
        if (mode == 1) return k == km ? RK(peln, km) : (k == 0 ? log(ptop) : log(ak[k] + bk[k] * psv));
        return k == 0 ? ptop : (k == km ? psv : ak[k] + bk[k] * psv);
      };
      auto Q1 = [&](int k) -> Real { return post == 2 ? -RK(src, k) / RK(delp, k) : RK(src, k); };
      auto GAMf = [&](int k) -> Real & { return RK(GAM, k); };
      auto QEf = [&](int k) -> Real & { return RK(QE, k); };
      auto OUT = [&](int k, Real v_) { RK(Q2, k) = v_; };
      remap_col(km, PE1, PE2, Q1, GAMf, QEf, OUT, iv, qs2d ? qs2d[t * g.st2 + pix] : (Real)0, use_qmin, qmin);
      // the column is done: its own thread moves the result home
      for (int k = 0; k < km; ++k) {
        Real v_ = RK(Q2, k);
        if (post == 2) {  // layer thickness from the remapped -delz / delp and the Eulerian dp
          const Real p0 = k == 0 ? ptop : ak[k] + bk[k] * psv, p1 = k + 1 == km ? psv : ak[k + 1] + bk[k + 1] * psv;
          v_ = -v_ * (p1 - p0);
        }
        RK(field, k) = v_;
      }
    });
  };
  scalar(TV, TV, 1, 1, nullptr, true, t_min, 0);
  for (int n = 0; n < n_tracers; ++n) scalar(q[n], q[n], 0, 0, nullptr, false, (Real)0, 0);
  scalar(w, w, 0, -2, wsd, false, (Real)0, 0);
  scalar(delz, delz, 0, 1, nullptr, false, (Real)0, 2);
  // ---- D-grid winds: interfaces averaged to the wind points (pe's halo ring comes from edge_pe)
  auto wind = [&](Real *field, bool is_u) {
    launch2(c, s, is_u ? Box{1, g.nx, 1, g.ny + 1, 0, 0} : Box{1, g.nx + 1, 1, g.ny, 0, 0}, [=] FV3_HD(int t, int i, int j) {
      const long tb = t * g.st;
      const unsigned pix = IX(i, j), pnb = is_u ? IX(i, j - 1) : IX(i - 1, j);
      auto PEH = [&](int l) -> Real { return (Real)0.5 * ((pe + tb + (long)l * g.sk)[pnb] + RK(pe, l)); };
      const Real psm = PEH(km);
      auto PE1 = [&](int l) -> Real { return l == 0 ? ptop : PEH(l); };
      auto PE2 = [&](int k) -> Real { return k == 0 ? ptop : (k == km ? psm : ak[k] + bk[k] * psm); };
      auto Q1 = [&](int k) -> Real { return RK(field, k); };
      auto GAMf = [&](int k) -> Real & { return RK(GAM, k); };
      auto QEf = [&](int k) -> Real & { return RK(QE, k); };
      auto OUT = [&](int k, Real v_) { RK(Q2, k) = v_; };
      remap_col(km, PE1, PE2, Q1, GAMf, QEf, OUT, -1, (Real)0, false, (Real)0);
      for (int k = 0; k < km; ++k) RK(field, k) = RK(Q2, k);
    });
  };
  wind(u, true);
  wind(v, false);
  // ---- Eulerian pressures, pkz from the remapped T_v, pt back to the loop's form, the new layer thickness
  launch2(c, s, cells, [=] FV3_HD(int t, int i, int j) {
    const long tb = t * g.st;
    const unsigned pix = IX(i, j);
    const Real psv = RK(pe, km);
    ps[t * g.st2 + pix] = psv;
    Real p0 = ptop;
    RK(peln, 0) = log(ptop);
    RK(pk, 0) = exp(akap * log(ptop));
    for (int k = 0; k < km; ++k) {
      const Real p1 = k + 1 == km ? psv : ak[k + 1] + bk[k + 1] * psv;
      const Real dp2 = p1 - p0;
      const Real cp = RK(cappa, k), tv = RK(TV, k);
      const Real pz = exp(cp * log(rrg * dp2 / RK(delz, k) * tv));
      RK(pkz, k) = pz;
      RK(pt, k) = tv / pz;
      RK(delp, k) = dp2;
      if (k + 1 < km) {
        const Real pn = log(p1);
        RK(pe, k + 1) = p1;
        RK(peln, k + 1) = pn;
        RK(pk, k + 1) = exp(akap * pn);
      } else {
        RK(pk, km) = exp(akap * RK(peln, km));
      }
      p0 = p1;
    }
    RK(pe, 0) = ptop;
  });
  return fv3_post(c, s, "remap");
}
