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
//
// Three sweeps, each streaming (every accessor is called once per level and sweep, with the neighbouring levels kept in a
// sliding register window and the next level's loads issued a step ahead -- the first form of this routine called the
// accessors where the formulas name them, ~40 dependent loads per level, and ran at 0.3 TB/s):
//   1 (down the column) forward elimination of the edge-value system            reads Q1, PE1        writes QE, GAM
//   2 (up)              back substitution + the large-scale constraints         reads QE, GAM, Q1    writes QE
//   3 (down)            the limited parabola of each source layer as the conservative integration reaches it
//                                                                               reads Q1, QE, PE1    writes OUT
// The arithmetic is expression for expression the one of oracle/remap_oracle.c.
template <class PE1, class PE2, class Q1F, class GAMF, class QEF, class OUTF>
FV3_HD inline void remap_col(int km, PE1 pe1, PE2 pe2, Q1F Q1, GAMF GAM, QEF QE, OUTF OUT, int iv, Real qs, bool use_qmin, Real qmin) {
  // ---- sweep 1: forward elimination (cs_profile).  q_a / q_b = Q1(k-1) / Q1(k), dp_a / dp_b = the layer thicknesses
  Real qp, gp = (Real)0, d4 = (Real)0, q_last2, q_last1;  // (q_last2 / q_last1 = Q1(km-2) / Q1(km-1) for the closure)
  {
    Real pe_a = pe1(0), pe_b = pe1(1), pe_c = pe1(2);
    Real q_a = Q1(0), q_b = Q1(1);
    Real dp_a = pe_b - pe_a, dp_b = pe_c - pe_b;
    if (iv == -2) {
      Real gam_k = (Real)0.5;  // GAM(1)
      GAM(1) = gam_k;
      qp = (Real)1.5 * q_a;
      QE(0) = qp;
      for (int k = 1; k < km - 1; ++k) {
        // prefetch of the next level
        const Real pe_n = pe1(k + 2), q_n = Q1(k + 1);
        const Real grat = dp_a / dp_b;
        const Real bet = (Real)2.0 + grat + grat - gam_k;
        qp = ((Real)3.0 * (q_a + q_b) - qp) / bet;
        QE(k) = qp;
        gam_k = grat / bet;
        GAM(k + 1) = gam_k;
        dp_a = dp_b;
        dp_b = pe_n - pe_c;
        pe_c = pe_n;
        q_a = q_b;
        q_b = q_n;
      }
      const Real grat = dp_a / dp_b;
      qp = ((Real)3.0 * (q_a + q_b) - grat * qs - qp) / ((Real)2.0 + grat + grat - gam_k);
      q_last2 = q_a;
      q_last1 = q_b;
    } else {
      const Real grat = dp_b / dp_a;
      Real bet = grat * (grat + (Real)0.5);
      qp = ((grat + grat) * (grat + (Real)1.0) * q_a + q_b) / bet;
      QE(0) = qp;
      gp = ((Real)1.0 + grat * (grat + (Real)1.5)) / bet;
      GAM(0) = gp;
      for (int k = 1; k < km; ++k) {
        const int kn = k + 2 <= km ? k + 2 : km, qn_ = k + 1 <= km - 1 ? k + 1 : km - 1;
        const Real pe_n = pe1(kn), q_n = Q1(qn_);
        d4 = dp_a / dp_b;
        bet = (Real)2.0 + d4 + d4 - gp;
        qp = ((Real)3.0 * (q_a + d4 * q_b) - qp) / bet;
        QE(k) = qp;
        gp = d4 / bet;
        GAM(k) = gp;
        if (k + 1 < km) {
          dp_a = dp_b;
          dp_b = pe_n - pe_c;
          pe_c = pe_n;
          q_a = q_b;
          q_b = q_n;
        }
      }
      q_last2 = q_a;
      q_last1 = q_b;
    }
  }
  // ---- sweep 2: back substitution (the recurrence runs on the unconstrained values), the large-scale constraints on the way
  {
    // constrained edge value of interface k (1 .. km-1) from the four means around it: qm2 = Q1(k-2) .. qp1 = Q1(k+1)
    auto constrain = [&](int k, Real q, Real qm2, Real qm1, Real q0, Real qp1) {
      if (k == 1 || k == km - 1) {
        q = fv3_min(q, fv3_max(qm1, q0));
        q = fv3_max(q, fv3_min(qm1, q0));
        return q;
      }
      const Real lo = fv3_min(qm1, q0), hi = fv3_max(qm1, q0);
      const Real gm = qm1 - qm2, gpl = qp1 - q0;
      if (gm * gpl > (Real)0) {
        q = fv3_min(q, hi);
        q = fv3_max(q, lo);
      } else if (gm > (Real)0) {
        q = fv3_max(q, lo);
      } else {
        q = fv3_min(q, hi);
        if (iv == 0) q = fv3_max((Real)0, q);
      }
      return q;
    };
    Real qn;
    int kstart;
    if (iv == -2) {
      QE(km) = qs;
      QE(km - 1) = constrain(km - 1, qp, (Real)0, q_last2, q_last1, (Real)0);
      qn = qp;
      kstart = km - 2;
    } else {
      const Real a_bot = (Real)1.0 + d4 * (d4 + (Real)1.5);
      qn = ((Real)2.0 * d4 * (d4 + (Real)1.0) * q_last1 + q_last2 - a_bot * qp) / (d4 * (d4 + (Real)0.5) - a_bot * gp);
      QE(km) = qn;
      kstart = km - 1;
    }
    // window of the means around interface k: w_m2 = Q1(k-2), w_m1 = Q1(k-1), w_0 = Q1(k), w_p1 = Q1(k+1)
    auto Qc = [&](int k) { return Q1(k < 0 ? 0 : k > km - 1 ? km - 1 : k); };
    Real w_p1 = Qc(kstart + 1), w_0 = Qc(kstart), w_m1 = Qc(kstart - 1), w_m2 = Qc(kstart - 2);
    Real e_k = QE(kstart), g_k = iv == -2 ? GAM(kstart + 1) : GAM(kstart);
    for (int k = kstart; k >= 0; --k) {
      // the level below in the sweep, a step ahead
      const int kb = k - 1 >= 0 ? k - 1 : 0;
      const Real e_n = QE(kb), g_n = iv == -2 ? GAM(kb + 1) : GAM(kb), w_n = Qc(k - 3);
      qn = e_k - g_k * qn;
      QE(k) = (k >= 1 && k <= km - 1) ? constrain(k, qn, w_m2, w_m1, w_0, w_p1) : qn;
      e_k = e_n;
      g_k = g_n;
      w_p1 = w_0;
      w_0 = w_m1;
      w_m1 = w_m2;
      w_m2 = w_n;
    }
  }
  // ---- sweep 3: conservative integration over the target layers; a cursor walks the source layers, holding the five means
  //      around the layer (qw[0 .. 4] = Q1(c-2 .. c+2)), its two constrained edge values and its interfaces
  {
    int c = 0;
    Real qw0 = (Real)0, qw1 = (Real)0, qw2 = Q1(0), qw3 = Q1(1), qw4 = Q1(2);
    Real ec = QE(0), ec1 = QE(1);
    Real s_lo = pe1(0), s_hi = pe1(1);
    // one level ahead: Q1(c+3), QE(c+2), pe1(c+2)
    Real q_n = Q1(3 <= km - 1 ? 3 : km - 1), e_n = QE(2), p_n = pe1(2);
    bool have = false;
    Prof cur{(Real)0, (Real)0, (Real)0, (Real)0};
    auto advance = [&]() {
      ++c;
      s_lo = s_hi;
      s_hi = p_n;
      qw0 = qw1;
      qw1 = qw2;
      qw2 = qw3;
      qw3 = qw4;
      qw4 = q_n;
      ec = ec1;
      ec1 = e_n;
      have = false;
      const int qi = c + 3 <= km - 1 ? c + 3 : km - 1, ei = c + 2 <= km ? c + 2 : km;
      q_n = Q1(qi);
      e_n = QE(ei);
      p_n = pe1(ei);
    };
    // the limited parabola of the cursor's layer (cs_profile, kord 9)
    auto profile = [&]() {
      if (have) return;
      have = true;
      const int l = c;
      Prof p;
      p.a1 = qw2;
      p.a2 = ec;
      p.a3 = ec1;
      const Real g0 = qw1 - qw0, g1 = qw2 - qw1, g2 = qw3 - qw2, g3 = qw4 - qw3;  // GD(l-1), GD(l), GD(l+1), GD(l+2)
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
        cs_limiters(g1 * g2 < (Real)0, p, 2);
      } else {
        const bool e = g1 * g2 < (Real)0;
        if ((e && g0 * g1 < (Real)0) || (e && g2 * g3 < (Real)0) || (use_qmin && e && p.a1 < qmin)) {
          p.a2 = p.a1, p.a3 = p.a1, p.a4 = (Real)0;
        } else {
          p.a4 = (Real)6.0 * p.a1 - (Real)3.0 * (p.a2 + p.a3);
          if (fabs(p.a4) > fabs(p.a2 - p.a3)) {
            const Real pmp_1 = p.a1 - (Real)2.0 * g2, lac_1 = pmp_1 + (Real)1.5 * g3;
            p.a2 = fv3_min(fv3_max(p.a2, rmin3(p.a1, pmp_1, lac_1)), rmax3(p.a1, pmp_1, lac_1));
            const Real pmp_2 = p.a1 + (Real)2.0 * g1, lac_2 = pmp_2 - (Real)1.5 * g0;
            p.a3 = fv3_min(fv3_max(p.a3, rmin3(p.a1, pmp_2, lac_2)), rmax3(p.a1, pmp_2, lac_2));
            p.a4 = (Real)6.0 * p.a1 - (Real)3.0 * (p.a2 + p.a3);
          }
        }
        if (iv == 0) cs_limiters(e, p, 0);
      }
      cur = p;
    };
    Real t_lo = pe2(0);
    for (int k = 0; k < km; ++k) {
      const Real t_hi = pe2(k + 1);
      // the source layer that holds the upper interface of the target layer
      while (!(t_lo >= s_lo && t_lo <= s_hi) && c < km - 1) advance();
      Real val;
      if (!(t_lo >= s_lo && t_lo <= s_hi)) {
        val = Q1(k);  // (not reached for interface sets that share their end points)
      } else {
        const Real dl = s_hi - s_lo;
        const Real pl = (t_lo - s_lo) / dl;
        profile();
        if (t_hi <= s_hi) {
          const Real pr = (t_hi - s_lo) / dl;
          val = cur.a2 + (Real)0.5 * (cur.a4 + cur.a3 - cur.a2) * (pr + pl) - cur.a4 * RM_R3 * (pr * (pr + pl) + pl * pl);
        } else {
          Real qsum = (s_hi - t_lo) * (cur.a2 + (Real)0.5 * (cur.a4 + cur.a3 - cur.a2) * ((Real)1.0 + pl) - cur.a4 * (RM_R3 * ((Real)1.0 + pl * ((Real)1.0 + pl))));
          while (c < km - 1) {
            advance();
            if (t_hi > s_hi) {
              qsum = qsum + (s_hi - s_lo) * qw2;
            } else {
              const Real dp = t_hi - s_lo, esl = dp / (s_hi - s_lo);
              profile();
              qsum = qsum + dp * (cur.a2 + (Real)0.5 * esl * (cur.a3 - cur.a2 + cur.a4 * ((Real)1.0 - RM_R23 * esl)));
              break;
            }
          }
          val = qsum / (t_hi - t_lo);
        }
      }
      OUT(k, val);
      t_lo = t_hi;
    }
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
