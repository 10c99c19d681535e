// fv3_ppm.h -- 1-D PPM building blocks (hord / iord 5-6) shared by xppm, yppm, xtp_u, ytp_v.
// [SURVEY A.4.2; oracle/fv3_oracle/ppm.py is the line-by-line CPU twin]
//
// Everything is expressed along a generic sweep index s: q(s) reads the advected quantity,
// m(s) the cell metric used by the tile-edge formulas (dxa / dya for scalars, dx / dy for the
// D-grid winds).  lo / hi say whether this rank touches the tile edge at the low / high end of
// the sweep axis, np_ = n + 1 in local numbering.  Expressions keep the oracle's operation
// order so that (with -ffp-contract=off) results agree to the last bit away from libm calls.
#pragma once
#include "fv3_common.h"

#define PPM_P1 ((Real)(7.0 / 12.0))
#define PPM_P2 ((Real)(-1.0 / 12.0))
#define PPM_C1 ((Real)(-2.0 / 14.0))
#define PPM_C2 ((Real)(11.0 / 14.0))
#define PPM_C3 ((Real)(5.0 / 14.0))

FV3_HD inline Real ppm_edge_mean(Real q_m1, Real q_0, Real q_p1, Real q_p2, Real d_m1, Real d_0, Real d_p1, Real d_p2) {
  return (Real)0.5 * ((((Real)2.0 * d_0 + d_m1) * q_0 - d_0 * q_m1) / (d_m1 + d_0) + (((Real)2.0 * d_p1 + d_p2) * q_p1 - d_p1 * q_p2) / (d_p1 + d_p2));
}

// edge value at the low face of cell s
template <class Q, class M>
FV3_HD inline Real ppm_al(Q q, M m, int s, bool lo, bool hi, int np_) {
  if (lo) {
    if (s == 0) return PPM_C1 * q(-2) + PPM_C2 * q(-1) + PPM_C3 * q(0);
    if (s == 1) return ppm_edge_mean(q(-1), q(0), q(1), q(2), m(-1), m(0), m(1), m(2));
    if (s == 2) return PPM_C3 * q(1) + PPM_C2 * q(2) + PPM_C1 * q(3);
  }
  if (hi) {
    if (s == np_ - 1) return PPM_C1 * q(np_ - 3) + PPM_C2 * q(np_ - 2) + PPM_C3 * q(np_ - 1);
    if (s == np_) return ppm_edge_mean(q(np_ - 2), q(np_ - 1), q(np_), q(np_ + 1), m(np_ - 2), m(np_ - 1), m(np_), m(np_ + 1));
    if (s == np_ + 1) return PPM_C3 * q(np_) + PPM_C2 * q(np_ + 1) + PPM_C1 * q(np_ + 2);
  }
  return PPM_P1 * (q(s - 1) + q(s)) + PPM_P2 * (q(s - 2) + q(s + 1));
}

FV3_HD inline bool ppm_smt5(Real bl, Real br, int mord) {
  const Real b0 = bl + br;
  return mord == 5 ? (bl * br) < (Real)0 : ((Real)3.0 * fabs(b0)) < fabs(bl - br);
}

// flux-form value crossing face s (between cells s-1 and s) with Courant number c.
// zero_m / zero_0: force bl = br = 0 in cell s-1 / s (xtp_u / ytp_v at the tile's own corners).
// cfl_m / cfl_0: Courant-number scale for c > 0 / c <= 0 (1 for scalars, rdx / rdy for winds).
template <class Q, class M>
FV3_HD inline Real ppm_flux(Q q, M m, Real c, int s, bool lo, bool hi, int np_, int mord, bool zero_m = false, bool zero_0 = false,
                            Real cfl_m = (Real)1, Real cfl_0 = (Real)1) {
  const Real al_m = ppm_al(q, m, s - 1, lo, hi, np_);
  const Real al_0 = ppm_al(q, m, s, lo, hi, np_);
  const Real al_p = ppm_al(q, m, s + 1, lo, hi, np_);
  const Real qm = q(s - 1), q0 = q(s);
  Real bl_m = al_m - qm, br_m = al_0 - qm;
  Real bl_0 = al_0 - q0, br_0 = al_p - q0;
  if (zero_m) bl_m = br_m = (Real)0;
  if (zero_0) bl_0 = br_0 = (Real)0;
  const Real b0_m = bl_m + br_m, b0_0 = bl_0 + br_0;
  const bool sm = ppm_smt5(bl_m, br_m, mord), s0 = ppm_smt5(bl_0, br_0, mord);
  Real fx1, flux;
  if (c > (Real)0) {
    const Real cfl = c * cfl_m;
    fx1 = ((Real)1 - cfl) * (br_m - cfl * b0_m);
    flux = qm;
  } else {
    const Real cfl = c * cfl_0;
    fx1 = ((Real)1 + cfl) * (bl_0 + cfl * b0_0);
    flux = q0;
  }
  if (sm || s0) flux = flux + fx1;
  return flux;
}

// ---- streaming (marching) form: a lane walks the sweep axis and keeps what consecutive faces share
// edge value at the low face of cell s from q(s-2), q(s-1), q(s), q(s+1) = a, b, c_, d (same
// expressions as ppm_al; s is uniform over the wave so the edge tests are scalar branches)
template <class M>
FV3_HD inline Real ppm_al_win(Real a, Real b, Real c_, Real d, M m, int s, bool lo, bool hi, int np_) {
  if (lo) {
    if (s == 0) return PPM_C1 * a + PPM_C2 * b + PPM_C3 * c_;
    if (s == 1) return ppm_edge_mean(a, b, c_, d, m(-1), m(0), m(1), m(2));
    if (s == 2) return PPM_C3 * b + PPM_C2 * c_ + PPM_C1 * d;
  }
  if (hi) {
    if (s == np_ - 1) return PPM_C1 * a + PPM_C2 * b + PPM_C3 * c_;
    if (s == np_) return ppm_edge_mean(a, b, c_, d, m(np_ - 2), m(np_ - 1), m(np_), m(np_ + 1));
    if (s == np_ + 1) return PPM_C3 * b + PPM_C2 * c_ + PPM_C1 * d;
  }
  return PPM_P1 * (b + c_) + PPM_P2 * (a + d);
}

struct PpmCell {
  Real bl, br, q;
  bool sm;
};

FV3_HD inline PpmCell ppm_cell(Real al_lo, Real al_hi, Real q, int mord) {
  PpmCell c;
  c.bl = al_lo - q;
  c.br = al_hi - q;
  c.q = q;
  c.sm = ppm_smt5(c.bl, c.br, mord);
  return c;
}

// flux-form value through the face between cell m (below) and cell o (above); = ppm_flux with unit cfl scales
FV3_HD inline Real ppm_face(const PpmCell &m, const PpmCell &o, Real c) {
  Real fx1, flux;
  if (c > (Real)0) {
    fx1 = ((Real)1 - c) * (m.br - c * (m.bl + m.br));
    flux = m.q;
  } else {
    fx1 = ((Real)1 + c) * (o.bl + c * (o.bl + o.br));
    flux = o.q;
  }
  if (m.sm || o.sm) flux = flux + fx1;
  return flux;
}

// face value from the six cells q(s-3..s+2) = a..f when no tile-edge formula is within reach
// (the expressions ppm_flux reduces to there)
FV3_HD inline Real ppm_flux_int(Real a, Real b, Real c_, Real d, Real e, Real f, Real cr, int mord) {
  const Real al_m = PPM_P1 * (b + c_) + PPM_P2 * (a + d);
  const Real al_0 = PPM_P1 * (c_ + d) + PPM_P2 * (b + e);
  const Real al_p = PPM_P1 * (d + e) + PPM_P2 * (c_ + f);
  const PpmCell m = ppm_cell(al_m, al_0, c_, mord), o = ppm_cell(al_0, al_p, d, mord);
  return ppm_face(m, o, cr);
}

// variants with Courant-number scales (xtp_u / ytp_v: c is a displacement, cfl = c * rdx of the upwind cell)
FV3_HD inline Real ppm_face_cfl(const PpmCell &m, const PpmCell &o, Real c, Real cfl_m, Real cfl_0) {
  Real fx1, flux;
  if (c > (Real)0) {
    const Real cfl = c * cfl_m;
    fx1 = ((Real)1 - cfl) * (m.br - cfl * (m.bl + m.br));
    flux = m.q;
  } else {
    const Real cfl = c * cfl_0;
    fx1 = ((Real)1 + cfl) * (o.bl + cfl * (o.bl + o.br));
    flux = o.q;
  }
  if (m.sm || o.sm) flux = flux + fx1;
  return flux;
}

FV3_HD inline Real ppm_flux_int_cfl(Real a, Real b, Real c_, Real d, Real e, Real f, Real cr, int mord, Real cfl_m, Real cfl_0) {
  const Real al_m = PPM_P1 * (b + c_) + PPM_P2 * (a + d);
  const Real al_0 = PPM_P1 * (c_ + d) + PPM_P2 * (b + e);
  const Real al_p = PPM_P1 * (d + e) + PPM_P2 * (c_ + f);
  const PpmCell m = ppm_cell(al_m, al_0, c_, mord), o = ppm_cell(al_0, al_p, d, mord);
  return ppm_face_cfl(m, o, cr, cfl_m, cfl_0);
}
