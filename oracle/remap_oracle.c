/* ORACLE (test infrastructure, never the shipped path) -- Lagrangian-to-Eulerian vertical remapping.
 *
 * CPU restatement, in plain C, of the reference operator pyFV3 `LagrangianToEulerian` (called from
 * DynamicalCore.step_dynamics after the acoustic loop and the tracer advection [REF driver/pace/driver/driver.py:494-504,
 * 639-644]; savepoint `Remapping-In/Out` with the variables cappa, delp, delz, dp1, omga, pe, peln, phis, pk, pkz, ps, pt,
 * te_2d, u, ua, v, va, w, wsd [REF tests/savepoint/thresholds/fv_dynamics.yaml:227-326]; options kord_tm -9, kord_mt 9,
 * kord_tr 9, kord_wz 9, consv_te 0 [REF driver/examples/configs/baroclinic_c12.yaml:45,65-68]).
 *
 * PARITY UNPINNED: pyFV3 is an un-vendored submodule (see oracle/fv3_oracle/util.py); this follows the published algorithm,
 * GFDL_atmos_cubed_sphere fv_mapz.F90 (Lagrangian_to_Eulerian, map_scalar, map1_ppm, mapn_tracer, cs_profile /
 * scalar_profile with kord 9, cs_limiters; Lin 2004 sec. 4), for the configuration of the reference configs: non-hydrostatic,
 * remap of T_v in log(p) (kord_tm < 0), moist-cappa form of pkz with the cappa field given, no energy fixer (consv_te 0),
 * no saturation adjustment, no fillz, omga left untouched.
 *
 * Arrays are the oracle's numpy layout: [i][j][k], k fastest, (nx + 2 nh + 1) x (ny + 2 nh + 1) x (nz + 1); compute cells
 * i = nh .. nh + nx - 1.  Build: gcc -O2 -shared -fPIC (oracle/build_remap.py), loaded with ctypes (fv3_oracle/remap.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define R3 (1.0 / 3.0)
#define R23 (2.0 / 3.0)
#define R12 (1.0 / 12.0)
#define KMAX 256

static double dmin(double a, double b) { return a < b ? a : b; }
static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin3(double a, double b, double c) { return dmin(a, dmin(b, c)); }
static double dmax3(double a, double b, double c) { return dmax(a, dmax(b, c)); }

/* cs_limiters on one cell: a[0] = mean, a[1] = top edge, a[2] = bottom edge, a[3] = curvature */
static void cs_limiters(int extm, double *a, int iv) {
  if (iv == 0) { /* positive definite */
    if (a[0] <= 0.0) {
      a[1] = a[0], a[2] = a[0], a[3] = 0.0;
    } else if (fabs(a[2] - a[1]) < -a[3]) {
      if (a[0] + 0.25 * (a[2] - a[1]) * (a[2] - a[1]) / a[3] + a[3] * R12 < 0.0) { /* the local minimum is negative */
        if (a[0] < a[2] && a[0] < a[1]) {
          a[2] = a[0], a[1] = a[0], a[3] = 0.0;
        } else if (a[2] > a[1]) {
          a[3] = 3.0 * (a[1] - a[0]);
          a[2] = a[1] - a[3];
        } else {
          a[3] = 3.0 * (a[2] - a[0]);
          a[1] = a[2] - a[3];
        }
      }
    }
  } else {
    int flat = iv == 1 ? ((a[0] - a[1]) * (a[0] - a[2]) >= 0.0) : extm;
    if (flat) {
      a[1] = a[0], a[2] = a[0], a[3] = 0.0;
    } else {
      double da1 = a[2] - a[1], da2 = da1 * da1, a6da = a[3] * da1;
      if (a6da < -da2) {
        a[3] = 3.0 * (a[1] - a[0]);
        a[2] = a[1] - a[3];
      } else if (a6da > da2) {
        a[3] = 3.0 * (a[2] - a[0]);
        a[1] = a[2] - a[3];
      }
    }
  }
}

/* cs_profile / scalar_profile, kord = 9: a4[k][0] holds the means on entry; on exit a4[k][1..3] = top edge, bottom edge,
 * curvature.  iv: 0 positive definite scalars, 1 others, -1 winds, -2 vertical velocity with the surface value qs.
 * use_qmin: scalar_profile's extra flattening of local extrema below qmin. */
static void cs_profile9(double qs, double a4[][4], const double *delp, int km, int iv, int use_qmin, double qmin) {
  double gam[KMAX + 2], q[KMAX + 2];
  int extm[KMAX + 2];
  int k;
  if (iv == -2) {
    double grat, bet;
    gam[1] = 0.5;
    q[0] = 1.5 * a4[0][0];
    for (k = 1; k < km - 1; ++k) {
      grat = delp[k - 1] / delp[k];
      bet = 2.0 + grat + grat - gam[k];
      q[k] = (3.0 * (a4[k - 1][0] + a4[k][0]) - q[k - 1]) / bet;
      gam[k + 1] = grat / bet;
    }
    grat = delp[km - 2] / delp[km - 1];
    q[km - 1] = (3.0 * (a4[km - 2][0] + a4[km - 1][0]) - grat * qs - q[km - 2]) / (2.0 + grat + grat - gam[km - 1]);
    q[km] = qs;
    for (k = km - 2; k >= 0; --k) q[k] = q[k] - gam[k + 1] * q[k + 1];
  } else {
    double grat = delp[1] / delp[0], bet = grat * (grat + 0.5), d4 = 0.0, a_bot;
    q[0] = ((grat + grat) * (grat + 1.0) * a4[0][0] + a4[1][0]) / bet;
    gam[0] = (1.0 + grat * (grat + 1.5)) / bet;
    for (k = 1; k < km; ++k) {
      d4 = delp[k - 1] / delp[k];
      bet = 2.0 + d4 + d4 - gam[k - 1];
      q[k] = (3.0 * (a4[k - 1][0] + d4 * a4[k][0]) - q[k - 1]) / bet;
      gam[k] = d4 / bet;
    }
    a_bot = 1.0 + d4 * (d4 + 1.5);
    q[km] = (2.0 * d4 * (d4 + 1.0) * a4[km - 1][0] + a4[km - 2][0] - a_bot * q[km - 1]) / (d4 * (d4 + 0.5) - a_bot * gam[km - 1]);
    for (k = km - 1; k >= 0; --k) q[k] = q[k] - gam[k] * q[k + 1];
  }
  /* large-scale constraints on the edge values (gam[k] := a4(k) - a4(k-1), k = 1 .. km-1) */
  q[1] = dmin(q[1], dmax(a4[0][0], a4[1][0]));
  q[1] = dmax(q[1], dmin(a4[0][0], a4[1][0]));
  for (k = 1; k < km; ++k) gam[k] = a4[k][0] - a4[k - 1][0];
  for (k = 2; k < km - 1; ++k) {
    if (gam[k - 1] * gam[k + 1] > 0.0) {
      q[k] = dmin(q[k], dmax(a4[k - 1][0], a4[k][0]));
      q[k] = dmax(q[k], dmin(a4[k - 1][0], a4[k][0]));
    } else if (gam[k - 1] > 0.0) { /* a local maximum */
      q[k] = dmax(q[k], dmin(a4[k - 1][0], a4[k][0]));
    } else { /* a local minimum */
      q[k] = dmin(q[k], dmax(a4[k - 1][0], a4[k][0]));
      if (iv == 0) q[k] = dmax(0.0, q[k]);
    }
  }
  q[km - 1] = dmin(q[km - 1], dmax(a4[km - 2][0], a4[km - 1][0]));
  q[km - 1] = dmax(q[km - 1], dmin(a4[km - 2][0], a4[km - 1][0]));
  for (k = 0; k < km; ++k) {
    a4[k][1] = q[k];
    a4[k][2] = q[k + 1];
  }
  for (k = 0; k < km; ++k) {
    if (k == 0 || k == km - 1)
      extm[k] = (a4[k][1] - a4[k][0]) * (a4[k][2] - a4[k][0]) > 0.0;
    else
      extm[k] = gam[k] * gam[k + 1] < 0.0;
  }
  /* sub-grid constraints: the two top and two bottom layers always monotone */
  if (iv == 0) {
    a4[0][1] = dmax(0.0, a4[0][1]);
  } else if (iv == -1) {
    if (a4[0][1] * a4[0][0] <= 0.0) a4[0][1] = 0.0;
  }
  a4[0][3] = 3.0 * (2.0 * a4[0][0] - (a4[0][1] + a4[0][2]));
  cs_limiters(extm[0], a4[0], 1);
  a4[1][3] = 3.0 * (2.0 * a4[1][0] - (a4[1][1] + a4[1][2]));
  cs_limiters(extm[1], a4[1], 2);
  /* interior, kord = 9 */
  for (k = 2; k < km - 2; ++k) {
    if ((extm[k] && extm[k - 1]) || (extm[k] && extm[k + 1]) || (use_qmin && extm[k] && a4[k][0] < qmin)) {
      a4[k][1] = a4[k][0], a4[k][2] = a4[k][0], a4[k][3] = 0.0; /* grid-scale 2-delta-z wave (or below the floor): flat */
    } else {
      a4[k][3] = 6.0 * a4[k][0] - 3.0 * (a4[k][1] + a4[k][2]);
      if (fabs(a4[k][3]) > fabs(a4[k][1] - a4[k][2])) { /* non-monotonic sub-grid profile within the smooth region */
        double pmp_1 = a4[k][0] - 2.0 * gam[k + 1], lac_1 = pmp_1 + 1.5 * gam[k + 2];
        double pmp_2, lac_2;
        a4[k][1] = dmin(dmax(a4[k][1], dmin3(a4[k][0], pmp_1, lac_1)), dmax3(a4[k][0], pmp_1, lac_1));
        pmp_2 = a4[k][0] + 2.0 * gam[k];
        lac_2 = pmp_2 - 1.5 * gam[k - 1];
        a4[k][2] = dmin(dmax(a4[k][2], dmin3(a4[k][0], pmp_2, lac_2)), dmax3(a4[k][0], pmp_2, lac_2));
        a4[k][3] = 6.0 * a4[k][0] - 3.0 * (a4[k][1] + a4[k][2]);
      }
    }
    if (iv == 0) cs_limiters(extm[k], a4[k], 0);
  }
  if (iv == 0) {
    a4[km - 1][2] = dmax(0.0, a4[km - 1][2]);
  } else if (iv == -1) {
    if (a4[km - 1][2] * a4[km - 1][0] <= 0.0) a4[km - 1][2] = 0.0;
  }
  a4[km - 2][3] = 3.0 * (2.0 * a4[km - 2][0] - (a4[km - 2][1] + a4[km - 2][2]));
  cs_limiters(extm[km - 2], a4[km - 2], 2);
  a4[km - 1][3] = 3.0 * (2.0 * a4[km - 1][0] - (a4[km - 1][1] + a4[km - 1][2]));
  cs_limiters(extm[km - 1], a4[km - 1], 1);
}

/* map1_ppm / map_scalar: conservative remap of q1 (layer means on the interfaces pe1) to the interfaces pe2 */
static void remap_column(int km, const double *pe1, const double *q1, const double *pe2, double *q2, int iv, double qs, int use_qmin, double qmin) {
  double a4[KMAX][4], dp1[KMAX];
  int k, l, m, k0 = 0;
  for (k = 0; k < km; ++k) {
    dp1[k] = pe1[k + 1] - pe1[k];
    a4[k][0] = q1[k];
  }
  cs_profile9(qs, a4, dp1, km, iv, use_qmin, qmin);
  for (k = 0; k < km; ++k) {
    int done = 0;
    for (l = k0; l < km && !done; ++l) {
      if (pe2[k] >= pe1[l] && pe2[k] <= pe1[l + 1]) {
        double pl = (pe2[k] - pe1[l]) / dp1[l];
        if (pe2[k + 1] <= pe1[l + 1]) { /* the new layer lies within the old one */
          double pr = (pe2[k + 1] - pe1[l]) / dp1[l];
          q2[k] = a4[l][1] + 0.5 * (a4[l][3] + a4[l][2] - a4[l][1]) * (pr + pl) - a4[l][3] * R3 * (pr * (pr + pl) + pl * pl);
          k0 = l;
          done = 1;
        } else { /* fractional first layer, whole layers, fractional last layer */
          double qsum = (pe1[l + 1] - pe2[k]) * (a4[l][1] + 0.5 * (a4[l][3] + a4[l][2] - a4[l][1]) * (1.0 + pl) - a4[l][3] * (R3 * (1.0 + pl * (1.0 + pl))));
          for (m = l + 1; m < km; ++m) {
            if (pe2[k + 1] > pe1[m + 1]) {
              qsum = qsum + dp1[m] * a4[m][0];
            } else {
              double dp = pe2[k + 1] - pe1[m], esl = dp / dp1[m];
              qsum = qsum + dp * (a4[m][1] + 0.5 * esl * (a4[m][2] - a4[m][1] + a4[m][3] * (1.0 - R23 * esl)));
              k0 = m;
              break;
            }
          }
          q2[k] = qsum / (pe2[k + 1] - pe2[k]);
          done = 1;
        }
      }
    }
    if (!done) q2[k] = q1[k < km ? k : km - 1]; /* (cannot happen for nested interface sets: both start at ptop and end at ps) */
  }
}

typedef struct {
  int ni, nj, nz, nh, nx, ny; /* allocation extents (ni, nj, nz + 1 levels), halo width, compute cells */
  double ptop, akap, rrg;     /* rrg = -rdgas / grav */
  double t_min;               /* 184 K: floor of the remapped temperature's flattening (map_scalar) */
} remap_geom;

#define AT(a, i, j, k) ((a)[((size_t)(i)*g->nj + (j)) * (g->nz + 1) + (k)])

/* One rank.  Fields in place: delp, pt (the loop's theta_v / pkz form), delz, w, u, v, tracers; pe, peln, pk, pkz are
 * rewritten for the Eulerian levels; ps (2-D, [i][j]) is written.  ws = the surface vertical velocity of the last acoustic
 * sub-step (wsd).  delp must be valid on the halo row / column the D-grid winds average over. */
void remap_rank(const remap_geom *g, const double *ak, const double *bk, double *delp, double *pt, double *delz, double *w, double *u, double *v, const double *cappa,
                double *pe, double *peln, double *pk, double *pkz, double *ps, const double *ws, int nq, double **tracers) {
  const int nz = g->nz, nh = g->nh;
  int i, j, k, n;
  double pe1[KMAX + 1], pe2[KMAX + 1], pn1[KMAX + 1], pn2[KMAX + 1], q1[KMAX], q2[KMAX], dp2[KMAX], pe0[KMAX + 1], pe3[KMAX + 1];
  /* ---- scalars on the compute cells */
  for (i = nh; i < nh + g->nx; ++i)
    for (j = nh; j < nh + g->ny; ++j) {
      double psv;
      pe1[0] = g->ptop;
      for (k = 0; k < nz; ++k) pe1[k + 1] = pe1[k] + AT(delp, i, j, k);
      psv = pe1[nz];
      for (k = 0; k <= nz; ++k) pe2[k] = ak[k] + bk[k] * psv;
      pe2[0] = g->ptop;
      pe2[nz] = psv;
      for (k = 0; k <= nz; ++k) {
        pn1[k] = log(pe1[k]);
        pn2[k] = log(pe2[k]);
      }
      pn2[nz] = pn1[nz];
      for (k = 0; k < nz; ++k) dp2[k] = pe2[k + 1] - pe2[k];
      /* T_v from the loop's pt (kord_tm < 0: remap the temperature), in log(p) */
      for (k = 0; k < nz; ++k) {
        const double ptv = AT(pt, i, j, k), cp = AT(cappa, i, j, k);
        q1[k] = ptv * exp(cp / (1.0 - cp) * log(g->rrg * AT(delp, i, j, k) / AT(delz, i, j, k) * ptv));
      }
      remap_column(nz, pn1, q1, pn2, q2, 1, 0.0, 1, g->t_min);
      for (k = 0; k < nz; ++k) AT(pt, i, j, k) = q2[k];
      /* tracers (mass weighted, positive definite) */
      for (n = 0; n < nq; ++n) {
        for (k = 0; k < nz; ++k) q1[k] = AT(tracers[n], i, j, k);
        remap_column(nz, pe1, q1, pe2, q2, 0, 0.0, 0, 0.0);
        for (k = 0; k < nz; ++k) AT(tracers[n], i, j, k) = q2[k];
      }
      /* vertical velocity (surface value ws), layer thickness (as -delz / delp) */
      for (k = 0; k < nz; ++k) q1[k] = AT(w, i, j, k);
      remap_column(nz, pe1, q1, pe2, q2, -2, ws[(size_t)i * g->nj + j], 0, 0.0);
      for (k = 0; k < nz; ++k) AT(w, i, j, k) = q2[k];
      for (k = 0; k < nz; ++k) q1[k] = -AT(delz, i, j, k) / AT(delp, i, j, k);
      remap_column(nz, pe1, q1, pe2, q2, 1, 0.0, 0, 0.0);
      for (k = 0; k < nz; ++k) AT(delz, i, j, k) = -q2[k] * dp2[k];
      /* Eulerian pressures; pkz = p^cappa of the full (non-hydrostatic) pressure rho R T_v from the remapped T_v [moist_pkz]; pt back to the
       * loop's form T_v / pkz.  (The conversion at the top has the exponent cappa / (1 - cappa) because it starts from pt = T_v / pkz.) */
      for (k = 0; k <= nz; ++k) {
        AT(pe, i, j, k) = pe2[k];
        AT(peln, i, j, k) = pn2[k];
        AT(pk, i, j, k) = exp(g->akap * pn2[k]);
      }
      ps[(size_t)i * g->nj + j] = psv;
      for (k = 0; k < nz; ++k) {
        const double cp = AT(cappa, i, j, k);
        const double pz = exp(cp * log(g->rrg * dp2[k] / AT(delz, i, j, k) * AT(pt, i, j, k)));
        AT(pkz, i, j, k) = pz;
        AT(pt, i, j, k) = AT(pt, i, j, k) / pz;
      }
    }
  /* ---- D-grid winds: interfaces averaged to the wind points (u: rows j and j-1; v: columns i and i-1), old delp in the halo */
  for (i = nh; i < nh + g->nx; ++i)
    for (j = nh; j <= nh + g->ny; ++j) {
      double a = g->ptop, b = g->ptop, psm;
      pe0[0] = g->ptop;
      for (k = 0; k < nz; ++k) {
        a += AT(delp, i, j - 1, k);
        b += AT(delp, i, j, k);
        pe0[k + 1] = 0.5 * (a + b);
      }
      psm = 0.5 * (a + b);
      for (k = 0; k <= nz; ++k) pe3[k] = ak[k] + bk[k] * psm;
      pe3[0] = g->ptop;
      pe3[nz] = pe0[nz];
      for (k = 0; k < nz; ++k) q1[k] = AT(u, i, j, k);
      remap_column(nz, pe0, q1, pe3, q2, -1, 0.0, 0, 0.0);
      for (k = 0; k < nz; ++k) AT(u, i, j, k) = q2[k];
    }
  for (i = nh; i <= nh + g->nx; ++i)
    for (j = nh; j < nh + g->ny; ++j) {
      double a = g->ptop, b = g->ptop, psm;
      pe0[0] = g->ptop;
      for (k = 0; k < nz; ++k) {
        a += AT(delp, i - 1, j, k);
        b += AT(delp, i, j, k);
        pe0[k + 1] = 0.5 * (a + b);
      }
      psm = 0.5 * (a + b);
      for (k = 0; k <= nz; ++k) pe3[k] = ak[k] + bk[k] * psm;
      pe3[0] = g->ptop;
      pe3[nz] = pe0[nz];
      for (k = 0; k < nz; ++k) q1[k] = AT(v, i, j, k);
      remap_column(nz, pe0, q1, pe3, q2, -1, 0.0, 0, 0.0);
      for (k = 0; k < nz; ++k) AT(v, i, j, k) = q2[k];
    }
  /* ---- the new layer thickness last (the winds above read the Lagrangian one, halo included) */
  for (i = nh; i < nh + g->nx; ++i)
    for (j = nh; j < nh + g->ny; ++j)
      for (k = 0; k < nz; ++k) AT(delp, i, j, k) = AT(pe, i, j, k + 1) - AT(pe, i, j, k);
}

/* single-column entry for the property tests */
void remap_one(int km, const double *pe1, const double *q1, const double *pe2, double *q2, int iv, double qs, int use_qmin, double qmin) {
  remap_column(km, pe1, q1, pe2, q2, iv, qs, use_qmin, qmin);
}
