// fv3_dsw.hip -- D-grid full step d_sw: fxadv flux preparation, transport of delp / w / q_con /
// pt, corner kinetic energy (xtp_u / ytp_v), divergence damping, vorticity flux, damping heat.
// CPU twin: oracle/fv3_oracle/d_sw.py.  [SURVEY A.3; reference operator
// DGridShallowWaterLagrangianDynamics, REF tests/savepoint/thresholds/fv_dynamics.yaml:76-170]
//
// Per-level parameters (nord, damp, d_con of the sponge layers) come from device tables indexed
// by k = blockIdx.z, so one launch serves all levels (the reference builds one stencil per
// k-range through restrict_vertical).
#include "fv3_a2b.h"
#include "fv3_ops.h"
#include "fv3_ppm.h"

namespace {

// ---------------------------------------------------------------------------------------------
// fxadv: contravariant winds as pure functions of (uc, vc)
// ---------------------------------------------------------------------------------------------
struct FxAdv {
  Geo g;
  const Real *uc, *vc;  // level bases
  long m2;
  Real dt;
  bool W, E, S, N;

  FV3_HD Real UC(int i, int j) const { return uc[IX(i, j)]; }
  FV3_HD Real VC(int i, int j) const { return vc[IX(i, j)]; }
  FV3_HD Real ut_gen(int i, int j) const {
    return (UC(i, j) - (Real)0.25 * (g.cosa_u + m2)[IX(i, j)] * (VC(i - 1, j) + VC(i, j) + VC(i - 1, j + 1) + VC(i, j + 1))) * (g.rsin_u + m2)[IX(i, j)];
  }
  FV3_HD Real vt_gen(int i, int j) const {
    return (VC(i, j) - (Real)0.25 * (g.cosa_v + m2)[IX(i, j)] * (UC(i, j - 1) + UC(i + 1, j - 1) + UC(i, j) + UC(i + 1, j))) * (g.rsin_v + m2)[IX(i, j)];
  }
  FV3_HD bool edge_col(int i) const { return (W && i == 1) || (E && i == g.npx); }
  FV3_HD bool edge_row(int j) const { return (S && j == 1) || (N && j == g.npy); }
  FV3_HD bool ut_special_row(int j) const { return (S && (j == 0 || j == 1)) || (N && (j == g.npy - 1 || j == g.npy)); }
  FV3_HD Real ut_edge(int i, int j) const {
    const Real u_ = UC(i, j);
    return u_ * dt > (Real)0 ? u_ / (g.sin_sg3 + m2)[IX(i - 1, j)] : u_ / (g.sin_sg1 + m2)[IX(i, j)];
  }
  FV3_HD Real vt_edge(int i, int j) const {
    const Real v_ = VC(i, j);
    return v_ * dt > (Real)0 ? v_ / (g.sin_sg4 + m2)[IX(i, j - 1)] : v_ / (g.sin_sg2 + m2)[IX(i, j)];
  }
  // stage-1 values (before the S/N row, W/E column and corner refinements)
  FV3_HD Real ut1(int i, int j) const {
    if (edge_col(i)) return ut_edge(i, j);
    if (ut_special_row(j)) return (Real)0;  // not produced by the general formula
    return ut_gen(i, j);
  }
  FV3_HD Real vt1(int i, int j) const {
    if (edge_row(j)) return vt_edge(i, j);
    return vt_gen(i, j);
  }
  // ranges of the edge-parallel refinements
  FV3_HD bool in_i_rng(int i) const { return i >= (W ? 3 : 1) && i <= (E ? g.npx - 2 : g.nx + 1); }
  FV3_HD bool in_j_rng(int j) const { return j >= (S ? 3 : 1) && j <= (N ? g.npy - 2 : g.ny + 1); }
  FV3_HD bool vt_edge_col(int i) const { return (W && (i == 0 || i == 1)) || (E && (i == g.npx - 1 || i == g.npx)); }
  // stage-2: all but the cube-corner solves
  FV3_HD Real ut2(int i, int j) const {
    if (edge_col(i)) return ut_edge(i, j);
    if (ut_special_row(j)) {
      if (in_i_rng(i))
        return UC(i, j) - (Real)0.25 * (g.cosa_u + m2)[IX(i, j)] * (vt1(i - 1, j) + vt1(i, j) + vt1(i - 1, j + 1) + vt1(i, j + 1));
      return (Real)0;
    }
    return ut_gen(i, j);
  }
  FV3_HD Real vt2(int i, int j) const {
    if (edge_row(j)) return vt_edge(i, j);
    if (vt_edge_col(i) && in_j_rng(j))
      return VC(i, j) - (Real)0.25 * (g.cosa_v + m2)[IX(i, j)] * (ut1(i, j - 1) + ut1(i + 1, j - 1) + ut1(i, j) + ut1(i + 1, j));
    return vt_gen(i, j);
  }
  // cube-corner coupled solves (oracle _corner_solve): the target's one unknown neighbour -- its
  // partner across the corner -- is eliminated with the partner's own averaging formula.
  FV3_HD bool ut_corner_target(int i, int j, bool &west, bool &south) const {
    const int npx = g.npx, npy = g.npy;
    if (W && i == 2) {
      if (S && (j == 0 || j == 1)) { west = true; south = true; return true; }
      if (N && (j == npy - 1 || j == npy)) { west = true; south = false; return true; }
    }
    if (E && i == npx - 1) {
      if (S && (j == 0 || j == 1)) { west = false; south = true; return true; }
      if (N && (j == npy - 1 || j == npy)) { west = false; south = false; return true; }
    }
    return false;
  }
  FV3_HD bool vt_corner_target(int i, int j, bool &west, bool &south) const {
    const int npx = g.npx, npy = g.npy;
    if (S && j == 2) {
      if (W && (i == 0 || i == 1)) { west = true; south = true; return true; }
      if (E && (i == npx - 1 || i == npx)) { west = false; south = true; return true; }
    }
    if (N && j == npy - 1) {
      if (W && (i == 0 || i == 1)) { west = true; south = false; return true; }
      if (E && (i == npx - 1 || i == npx)) { west = false; south = false; return true; }
    }
    return false;
  }
  FV3_HD Real ut_corner(int it, int j, bool west, bool south) const {
    const int npx = g.npx, npy = g.npy;
    const int pi = west ? 1 : npx - 1;
    const int jedge = south ? 1 : npy;
    const int nbi[4] = {it - 1, it, it - 1, it}, nbj[4] = {j, j, j + 1, j + 1};
    const int pj = (j != jedge) ? j : j + 1;  // first neighbour row that is not the edge row
    Real s_v = (Real)0;
    for (int n = 0; n < 4; ++n)
      if (!(nbi[n] == pi && nbj[n] == pj)) s_v = s_v + vt2(nbi[n], nbj[n]);
    const int pbi[4] = {pi, pi + 1, pi, pi + 1}, pbj[4] = {pj - 1, pj - 1, pj, pj};
    Real s_u = (Real)0;
    for (int n = 0; n < 4; ++n)
      if (!(pbi[n] == it && pbj[n] == j)) s_u = s_u + ut2(pbi[n], pbj[n]);
    const Real cu = (g.cosa_u + m2)[IX(it, j)], cv = (g.cosa_v + m2)[IX(pi, pj)];
    const Real damp = (Real)1 / ((Real)1 - (Real)0.0625 * cu * cv);
    return (UC(it, j) - (Real)0.25 * cu * (s_v + VC(pi, pj) - (Real)0.25 * cv * s_u)) * damp;
  }
  FV3_HD Real vt_corner(int i, int jt, bool west, bool south) const {
    const int npx = g.npx, npy = g.npy;
    const int pj = south ? 1 : npy - 1;
    const int iedge = west ? 1 : npx;
    const int nbi[4] = {i, i + 1, i, i + 1}, nbj[4] = {jt - 1, jt - 1, jt, jt};
    const int pi = (i != iedge) ? i : i + 1;
    Real s_u = (Real)0;
    for (int n = 0; n < 4; ++n)
      if (!(nbi[n] == pi && nbj[n] == pj)) s_u = s_u + ut2(nbi[n], nbj[n]);
    const int pbi[4] = {pi - 1, pi, pi - 1, pi}, pbj[4] = {pj, pj, pj + 1, pj + 1};
    Real s_v = (Real)0;
    for (int n = 0; n < 4; ++n)
      if (!(pbi[n] == i && pbj[n] == jt)) s_v = s_v + vt2(pbi[n], pbj[n]);
    const Real cv = (g.cosa_v + m2)[IX(i, jt)], cu = (g.cosa_u + m2)[IX(pi, pj)];
    const Real damp = (Real)1 / ((Real)1 - (Real)0.0625 * cu * cv);
    return (VC(i, jt) - (Real)0.25 * cv * (s_u + UC(pi, pj) - (Real)0.25 * cu * s_v)) * damp;
  }
  FV3_HD Real ut_final(int i, int j) const {
    bool w_, s_;
    if (ut_corner_target(i, j, w_, s_)) return ut_corner(i, j, w_, s_);
    return ut2(i, j);
  }
  FV3_HD Real vt_final(int i, int j) const {
    bool w_, s_;
    if (vt_corner_target(i, j, w_, s_)) return vt_corner(i, j, w_, s_);
    return vt2(i, j);
  }
};

// points where the final contravariant wind is the general interior expression (ut_final == ut_gen, vt_final == vt_gen)
FV3_HD inline bool fxadv_int_u(const Geo &g, int fl, int i, int j) {
  const bool W = fl & FV3_W, E = fl & FV3_E, S = fl & FV3_S, N = fl & FV3_N;
  return i >= 0 && !((W && i == 1) || (E && i == g.npx)) && !((S && (j == 0 || j == 1)) || (N && (j == g.npy - 1 || j == g.npy)));
}
FV3_HD inline bool fxadv_int_v(const Geo &g, int fl, int i, int j) {
  const bool W = fl & FV3_W, E = fl & FV3_E, S = fl & FV3_S, N = fl & FV3_N;
  return j >= 0 && !((S && j == 1) || (N && j == g.npy)) && !((W && (i == 0 || i == 1)) || (E && (i == g.npx - 1 || i == g.npx)));
}

// cx / cy (optional): accumulated Courant numbers of the tracer sub-cycling, cx += crx, cy += cry on the
// faces d_sw accumulates them (i in 1..nx+1 resp. j in 1..ny+1, all halo rows / columns)
// thin_ut: store ut / vt only within 4 cells of a cube-tile edge -- all d_sw reads of them afterwards are the tile-edge
// forms of the corner kinetic energy (the standalone operator stores them everywhere, as the reference does)
void fxadv(fv3_ctx *c, fv3_stream_t s, const Real *uc, const Real *vc, Real *crx, Real *cry, Real *xfx, Real *yfx, Real *ut, Real *vt, Real dt, Real *cx,
           Real *cy, bool thin_ut) {
  const Geo g = c->g;
  const bool first = c->seq_acc_first;  // (the sequencer's first sub-step of a call: cx / cy hold nothing yet -- 0 + cr, the field is not read)
  static const bool full_ut = getenv("FV3_FXADV_FULL_UT") != nullptr;  // A/B switch
  if (full_ut) thin_ut = false;
  const int isd = 1 - g.nh, ied = g.nx + g.nh, jsd = 1 - g.nh, jed = g.ny + g.nh;
  // Interior kernel, two levels per thread: away from the tile-edge rows / columns the contravariant wind is one
  // expression in (uc, vc) and seven metric terms per component; those are read once and used for both
  // levels (they are most of this kernel's bytes).  The tile-edge rows / columns (and the 16 cube-corner
  // solves) are a separate thin launch of the general per-level forms: keeping them out of the level
  // loop keeps its register footprint small.
  const int npair = (g.nz + FV3_KC - 1) / FV3_KC;
  launch3(c, s, Box{isd, ied, jsd, jed, 0, npair - 1}, [=] FV3_HD(int t, int kp, int i, int j) {
    const int fl = g.flags[t];
    const long m2 = t * g.st2;
    bool int_u = fxadv_int_u(g, fl, i, j), int_v = fxadv_int_v(g, fl, i, j);
    const unsigned p = IX(i, j);
    const bool out_x = int_u && i >= 1 && i <= g.nx + 1, out_y = int_v && j >= 1 && j <= g.ny + 1;
    const bool st_ut = !thin_ut || ((fl & FV3_W) && i <= 4) || ((fl & FV3_E) && i >= g.npx - 3) || ((fl & FV3_S) && j <= 4) || ((fl & FV3_N) && j >= g.npy - 3);
    int_u = int_u && (out_x || st_ut);  // (nothing to do where neither the Courant number nor ut itself is wanted)
    int_v = int_v && (out_y || st_ut);
    if (!int_u && !int_v) return;
    Real cu = 0, ru = 0, rdxa_m = 0, rdxa_0 = 0, dy_ = 0, s3 = 0, s1 = 0;
    Real cv = 0, rv = 0, rdya_m = 0, rdya_0 = 0, dx_ = 0, s4 = 0, s2 = 0;
    if (int_u) {
      cu = (g.cosa_u + m2)[p];
      ru = (g.rsin_u + m2)[p];
    }
    if (out_x) {
      rdxa_m = (g.rdxa + m2)[IX(i - 1, j)];
      rdxa_0 = (g.rdxa + m2)[p];
      dy_ = (g.dy + m2)[p];
      s3 = (g.sin_sg3 + m2)[IX(i - 1, j)];
      s1 = (g.sin_sg1 + m2)[p];
    }
    if (int_v) {
      cv = (g.cosa_v + m2)[p];
      rv = (g.rsin_v + m2)[p];
    }
    if (out_y) {
      rdya_m = (g.rdya + m2)[IX(i, j - 1)];
      rdya_0 = (g.rdya + m2)[p];
      dx_ = (g.dx + m2)[p];
      s4 = (g.sin_sg4 + m2)[IX(i, j - 1)];
      s2 = (g.sin_sg2 + m2)[p];
    }
#ifndef FV3_FXADV_UNROLL
#define FV3_FXADV_UNROLL 1
#endif
#pragma unroll FV3_FXADV_UNROLL
    for (int kk = 0; kk < FV3_KC; ++kk) {
      const int k = FV3_KC * kp + kk;
      if (k > g.nz - 1) break;
      const long b = t * g.st + k * g.sk;
      const Real *ucl = uc + b, *vcl = vc + b;
      if (int_u) {
        const Real utv = (ucl[p] - (Real)0.25 * cu * (vcl[IX(i - 1, j)] + vcl[p] + vcl[IX(i - 1, j + 1)] + vcl[IX(i, j + 1)])) * ru;
        if (st_ut) (ut + b)[p] = utv;
        if (out_x) {
          const Real x = dt * utv;
          Real cr;
          if (x > (Real)0) {
            cr = x * rdxa_m;
            FV3_ST_NT((xfx + b)[p], dy_ * x * s3);
          } else {
            cr = x * rdxa_0;
            FV3_ST_NT((xfx + b)[p], dy_ * x * s1);
          }
          FV3_ST_NT((crx + b)[p], cr);
          if (cx) FV3_ST_NT((cx + b)[p], (first ? (Real)0 : (cx + b)[p]) + cr);
        }
      }
      if (int_v) {
        const Real vtv = (vcl[p] - (Real)0.25 * cv * (ucl[IX(i, j - 1)] + ucl[IX(i + 1, j - 1)] + ucl[p] + ucl[IX(i + 1, j)])) * rv;
        if (st_ut) (vt + b)[p] = vtv;
        if (out_y) {
          const Real y = dt * vtv;
          Real cr;
          if (y > (Real)0) {
            cr = y * rdya_m;
            FV3_ST_NT((yfx + b)[p], dx_ * y * s4);
          } else {
            cr = y * rdya_0;
            FV3_ST_NT((yfx + b)[p], dx_ * y * s2);
          }
          FV3_ST_NT((cry + b)[p], cr);
          if (cy) FV3_ST_NT((cy + b)[p], (first ? (Real)0 : (cy + b)[p]) + cr);
        }
      }
    }
  });
  // tile-edge frame: rows 0, 1, npy-1, npy and columns 0, 1, npx-1, npx of the sub-domains that have those edges
  const int nfr = std::max(ied - isd + 1, jed - jsd + 1);
  launch3(c, s, Box{0, nfr - 1, 1, 8, 0, g.nz - 1}, [=] FV3_HD(int t, int k, int a, int side) {
    const int fl = g.flags[t];
    int i, j;
    if (side <= 4) {  // rows
      i = isd + a;
      if (i > ied) return;
      j = side <= 2 ? side - 1 : g.npy - 4 + side;
      if (!(fl & (side <= 2 ? FV3_S : FV3_N))) return;
    } else {  // columns (minus the points the row sides cover)
      j = jsd + a;
      if (j > jed) return;
      i = side <= 6 ? side - 5 : g.npx - 8 + side;
      if (!(fl & (side <= 6 ? FV3_W : FV3_E))) return;
      if (((fl & FV3_S) && (j == 0 || j == 1)) || ((fl & FV3_N) && (j == g.npy - 1 || j == g.npy))) return;
    }
    const bool int_u = fxadv_int_u(g, fl, i, j), int_v = fxadv_int_v(g, fl, i, j);
    const long b = t * g.st + k * g.sk, m2 = t * g.st2;
    FxAdv f{g, uc + b, vc + b, m2, dt, (fl & FV3_W) != 0, (fl & FV3_E) != 0, (fl & FV3_S) != 0, (fl & FV3_N) != 0};
    const unsigned p = IX(i, j);
    if (i >= 0 && !int_u) {  // ut on is-1..ie+3, jsd..jed
      const Real utv = f.ut_final(i, j);
      (ut + b)[p] = utv;
      if (i >= 1 && i <= g.nx + 1) {
        const Real x = dt * utv;
        Real cr;
        if (x > (Real)0) {
          cr = x * (g.rdxa + m2)[IX(i - 1, j)];
          (xfx + b)[p] = (g.dy + m2)[p] * x * (g.sin_sg3 + m2)[IX(i - 1, j)];
        } else {
          cr = x * (g.rdxa + m2)[p];
          (xfx + b)[p] = (g.dy + m2)[p] * x * (g.sin_sg1 + m2)[p];
        }
        (crx + b)[p] = cr;
        if (cx) (cx + b)[p] = (first ? (Real)0 : (cx + b)[p]) + cr;
      }
    }
    if (j >= 0 && !int_v) {  // vt on isd..ied, js-1..je+3
      const Real vtv = f.vt_final(i, j);
      (vt + b)[p] = vtv;
      if (j >= 1 && j <= g.ny + 1) {
        const Real y = dt * vtv;
        Real cr;
        if (y > (Real)0) {
          cr = y * (g.rdya + m2)[IX(i, j - 1)];
          (yfx + b)[p] = (g.dx + m2)[p] * y * (g.sin_sg4 + m2)[IX(i, j - 1)];
        } else {
          cr = y * (g.rdya + m2)[p];
          (yfx + b)[p] = (g.dx + m2)[p] * y * (g.sin_sg2 + m2)[p];
        }
        (cry + b)[p] = cr;
        if (cy) (cy + b)[p] = (first ? (Real)0 : (cy + b)[p]) + cr;
      }
    }
  });
}

// fill_corners of the D-grid staggered pair (x = "vc" at (cell, iface), y = "uc" at (iface, cell))
// used inside the divergence-damping iteration; returns the value a filled read would see.
FV3_HD inline Real dg_x(const Real *x, const Real *y, const Geo &g, int fl, int i, int j, Real sign) {
  // x lives at (x cell i, y interface j): corner block when i outside [1,nx] and j outside [1,npy]
  const int npx = g.npx, npy = g.npy;
  if ((i >= 1 && i <= g.nx) || (j >= 1 && j <= npy)) return x[IX(i, j)];
  if (i < 1 && j < 1) {
    if ((fl & (FV3_W | FV3_S)) != (FV3_W | FV3_S)) return x[IX(i, j)];
    const int a = 1 - i, b = 1 - j;  // target (1-a, 1-b)
    return sign * y[IX(1 - b, a)];
  }
  if (i < 1 && j > npy) {
    if ((fl & (FV3_W | FV3_N)) != (FV3_W | FV3_N)) return x[IX(i, j)];
    const int a = 1 - i, b = j - npy;
    return y[IX(1 - b, npy - a)];
  }
  if (i > g.nx && j < 1) {
    if ((fl & (FV3_E | FV3_S)) != (FV3_E | FV3_S)) return x[IX(i, j)];
    const int a = i - (npx - 1), b = 1 - j;  // target (npx-1+a, 1-b)
    return y[IX(npx + b, a)];
  }
  if ((fl & (FV3_E | FV3_N)) != (FV3_E | FV3_N)) return x[IX(i, j)];
  const int a = i - (npx - 1), b = j - npy;
  return sign * y[IX(npx + b, npy - a)];
}
FV3_HD inline Real dg_y(const Real *x, const Real *y, const Geo &g, int fl, int i, int j, Real sign) {
  // y lives at (x interface i, y cell j): corner block when i outside [1,npx] and j outside [1,ny]
  const int npx = g.npx, npy = g.npy;
  if ((i >= 1 && i <= npx) || (j >= 1 && j <= g.ny)) return y[IX(i, j)];
  if (i < 1 && j < 1) {
    if ((fl & (FV3_W | FV3_S)) != (FV3_W | FV3_S)) return y[IX(i, j)];
    const int b = 1 - i, a = 1 - j;  // target (1-b, 1-a)
    return sign * x[IX(a, 1 - b)];
  }
  if (i < 1 && j > g.ny) {
    if ((fl & (FV3_W | FV3_N)) != (FV3_W | FV3_N)) return y[IX(i, j)];
    const int b = 1 - i, a = j - (npy - 1);  // target (1-b, npy-1+a)
    return x[IX(a, npy + b)];
  }
  if (i > npx && j < 1) {
    if ((fl & (FV3_E | FV3_S)) != (FV3_E | FV3_S)) return y[IX(i, j)];
    const int b = i - npx, a = 1 - j;  // target (npx+b, 1-a)
    return x[IX(npx - a, 1 - b)];
  }
  if ((fl & (FV3_E | FV3_N)) != (FV3_E | FV3_N)) return y[IX(i, j)];
  const int b = i - npx, a = j - (npy - 1);
  return sign * x[IX(npx - a, npy + b)];
}

}  // namespace

extern "C" int fv3_fxadv(fv3_ctx *c, const fv3_field *uc_, const fv3_field *vc_, const fv3_field *crx_, const fv3_field *cry_, const fv3_field *xfx_,
                         const fv3_field *yfx_, const fv3_field *ut_, const fv3_field *vt_, double dt, void *stream) {
  if (!c) return FV3_ERR_ARG;
  FV3_FIELD(uc, uc_) FV3_FIELD(vc, vc_) FV3_FIELD(crx, crx_) FV3_FIELD(cry, cry_) FV3_FIELD(xfx, xfx_) FV3_FIELD(yfx, yfx_) FV3_FIELD(ut, ut_) FV3_FIELD(vt, vt_)
  fxadv(c, (fv3_stream_t)stream, uc, vc, crx, cry, xfx, yfx, ut, vt, (Real)dt, nullptr, nullptr, false);
  return fv3_post(c, (fv3_stream_t)stream, "fxadv");
}

// ---------------------------------------------------------------------------------------------
// Corner kinetic energy, marching form (interior corners: no tile-edge formula within reach).
// Lane = corner column, the wave walks j.  ytp_v runs from a 4-row register window of v two rows
// behind the loaded row (shared edge values / cell reconstructions between consecutive faces);
// xtp_u, ub and vb are evaluated on that same row with u exchanged through one LDS line.
//   ke = 0.5 * (vb * ytp_v(vb, v) + ub * xtp_u(ub, u))
// ---------------------------------------------------------------------------------------------
#define KE_OUT 58
#define KE_SEG 64
#define KE_PF 2

// wk (optional, round 5): the march also forms the cell-mean relative vorticity of the cells (i, jf - 1) under its owned corners -- it holds u of rows jf - 1 / jf
// and v of row jf - 1 in every lane (v of column i + 1 in the next lane) -- so that the vorticity launch only serves the frame the march does not cover
// (experiment R5-30, FV3_DSW_VORT_IN_KE=1; off by default: what the vorticity launch saves, this march pays).  vabs: the absolute
// vorticity, stored on the levels below fdw_k0 (the sponge layers' transport reads it).  The expressions are the vorticity launch's.
template <bool VORT>
static void ke_stream_t(fv3_ctx *c, fv3_stream_t s, const Real *u, const Real *v, const Real *uc, const Real *vc, Real *ke, Real dt, int hord, int k0, int k1, Real *wk,
                        Real *vabs, int fdw_k0) {
  const Geo g = c->g;
  const int nk = k1 - k0 + 1;
  const int nstrip = (g.nx + 1 + KE_OUT - 1) / KE_OUT;
  const int seg = fv3_pick_seg((long)nstrip * ((g.ny + 64) / 64) * g.nsub * nk, 4);
  const int nseg = (g.ny + 1 + seg - 1) / seg;
  const size_t smem = sizeof(Real) * (FV3_WAVE + 6);
  const Geo *gp = c->g_dev;
  const int nx = g.nx, ny = g.ny, nh = g.nh, npx = g.npx, npy = g.npy, sj32 = g.sj32, go = g.o;
  const long st = g.st, sk = g.sk, st2 = g.st2;
  const MPtr cosa = g.cosa, rsina = g.rsina, rdx = g.rdx, rdy = g.rdy;
  const MPtr gdx = g.dx, gdy = g.dy, gra = g.rarea, gf0 = g.f0;
  // Level-major launch geometry (round 4; as in the transport marches, fv3_tp4.hip): KB levels of one (strip, segment) tile are consecutive
  // workgroups of an XCD, so the tile's four metric rows (cosa, rsina, rdx, rdy: 4 of the 9 row reads of a step) are fetched into that XCD's L2
  // once per KB levels -- plane-major, the 4.8 MB of metric terms of a 384^2 sub-domain do not survive a 4 MB L2 from one level to the next.
  // FV3_KE_KB=0: plane-major (A/B).
  static const int kb_env = getenv("FV3_KE_KB") ? atoi(getenv("FV3_KE_KB")) : 16;
  const int KB = kb_env > 0 ? (kb_env < nk ? kb_env : nk) : 0;
  const int nblk = KB ? (nk + KB - 1) / KB : 0;
  launch_waves<4>(c, s, KB ? KB : nstrip, KB ? nstrip * nseg : nseg, KB ? g.nsub * nblk : g.nsub * nk, smem, [=] FV3_HD(const Blk &blk_, char *smem_) {
    Blk blk = blk_;
    int t, k;
    if (KB) {
      t = blk_.bz / nblk;
      const int kk = (blk_.bz - t * nblk) * KB + blk_.bx;
      if (kk >= nk) return;
      k = k0 + kk;
      blk.by = blk_.by / nstrip;
      blk.bx = blk_.by - blk.by * nstrip;
    } else {
      t = blk.bz / nk;
      k = k0 + (blk.bz - t * nk);
    }
    const int fl = gp->flags[t];
    const long b = t * st + k * sk, m2 = t * st2;
    // corners of this sub-domain that take the interior formulas
    const int ia = (fl & FV3_W) ? 4 : 1, ib = (fl & FV3_E) ? npx - 3 : nx + 1;
    const int ja_ = (fl & FV3_S) ? 4 : 1, jb_ = (fl & FV3_N) ? npy - 3 : ny + 1;
    const int i0 = 1 + blk.bx * KE_OUT;
    int j0 = 1 + blk.by * seg, j1 = j0 + seg - 1;
    if (j0 < ja_) j0 = ja_;
    if (j1 > jb_) j1 = jb_;
    if (j0 > j1) return;
    const int ilo = i0 > ia ? i0 : ia, ihi = i0 + KE_OUT - 1 < ib ? i0 + KE_OUT - 1 : ib;
    if (ilo > ihi) return;
    Real *lu = (Real *)smem_;  // u of row jf;  lu[lane + 3 + d] <-> column i + d
    const int imax = nx + nh + 1, jmax = ny + nh;
    const Real *ub_ = u + b, *vb_ = v + b, *ucb = uc + b, *vcb = vc + b;
    const MPtr cob = cosa + m2, rsb = rsina + m2, rdxb = rdx + m2, rdyb = rdy + m2;
    const MPtr dxb = gdx + m2, dyb = gdy + m2, rab = gra + m2, f0b = gf0 + m2;
    const bool abs_on = VORT && vabs && k < fdw_k0;
    const Real dt5 = (Real)0.5 * dt;
    struct Row {  // row jf = r - 2 of everything but v, whose window runs two rows ahead
      Real vnew, uu, ucc, vcc, vcm, co, rs, rx, rxm, ry;
      Real dxr, dyc, rac;  // VORT: dx of row jf; dy, 1 / area of row jf - 1 (the vorticity's cell row)
    };
    Row pf[KE_PF][FV3_LPT];
    Real w2[FV3_LPT], w3[FV3_LPT], w4[FV3_LPT], w5[FV3_LPT], al_v[FV3_LPT];
    PpmCell cv[FV3_LPT];
    Real uc_prev[FV3_LPT], ry_prev[FV3_LPT];
    Real a_prev[FV3_LPT];  // VORT: u * dx of row jf - 1
    unsigned pcol[FV3_LPT];
    bool own[FV3_LPT];
    const int r_beg = j0 - 1, r_end = j1 + 2;  // v rows j0-3 .. are fed by the start-up below
    auto load_row = [&](int r, int l) -> Row {
      int rv = r < 1 - nh ? 1 - nh : r;
      if (rv > jmax) rv = jmax;
      int jf = r - 2 < 1 - nh ? 1 - nh : r - 2;
      if (jf > jmax) jf = jmax;
      const unsigned pv = pcol[l] + (unsigned)(rv * sj32), p = pcol[l] + (unsigned)(jf * sj32);
      Row w;
      w.vnew = vb_[pv];
      w.uu = ub_[p];
      w.ucc = ucb[p];
      w.vcc = vcb[p];
      w.vcm = vcb[p - (p != 0u)];
      w.co = cob[p];
      w.rs = rsb[p];
      w.rx = rdxb[p];
      w.rxm = rdxb[p - (p != 0u)];
      w.ry = rdyb[p];
      w.dxr = w.dyc = w.rac = (Real)0;
      if (VORT) {
        int jc = r - 3 < 1 - nh ? 1 - nh : r - 3;
        if (jc > jmax) jc = jmax;
        const unsigned pc = pcol[l] + (unsigned)(jc * sj32);
        w.dxr = dxb[p];
        w.dyc = dyb[pc];
        w.rac = rab[pc];
      }
      return w;
    };
    FV3_LANES(blk, lane, l) {
      const int i = i0 - 3 + lane, ic = i < imax ? i : imax;
      pcol[l] = (unsigned)(go * sj32 + ic + go);
      own[l] = i >= ilo && i <= ihi;
      w2[l] = w3[l] = w4[l] = w5[l] = al_v[l] = uc_prev[l] = ry_prev[l] = a_prev[l] = (Real)0;
      cv[l] = PpmCell{(Real)0, (Real)0, (Real)0, false};
      if (lane < 3) lu[lane] = lu[FV3_WAVE + 3 + lane] = (Real)0;
    }
    // start-up: the window needs v rows j0-3 .. j0+2 before the first face j0; steps r = j0-3 .. feed it
    const int r_first = r_beg - 2;
    FV3_LANES(blk, lane, l) {
#pragma unroll
      for (int n = 0; n < KE_PF; ++n) pf[n][l] = load_row(r_first + n, l);
    }
    for (int r = r_first; r <= r_end; ++r) {
      const int rn = r + KE_PF;
      const int jf = r - 2;  // face of ytp_v / row of everything else
      Real uu_[FV3_LPT], ubv_[FV3_LPT], vbv_[FV3_LPT], vfl_[FV3_LPT], rx_[FV3_LPT], rxm_[FV3_LPT];
      Real va_[FV3_LPT], va1_[FV3_LPT], ve_[FV3_LPT], vra_[FV3_LPT];  // VORT: u * dx of rows jf - 1 / jf, v * dy, 1 / area of row jf - 1
      FV3_LANES(blk, lane, l) {
        const Row cu = pf[0][l];
#pragma unroll
        for (int n = 0; n + 1 < KE_PF; ++n) pf[n][l] = pf[n + 1][l];
        pf[KE_PF - 1][l] = load_row(rn, l);
        // v window: rows r-3 .. r;  al(r-1), cell r-2, face jf = r-2 between cells r-3 and r-2
        w2[l] = w3[l];
        w3[l] = w4[l];
        w4[l] = w5[l];
        w5[l] = cu.vnew;
        const Real al_new = PPM_P1 * (w3[l] + w4[l]) + PPM_P2 * (w2[l] + w5[l]);
        const PpmCell co = ppm_cell(al_v[l], al_new, w3[l], hord);
        al_v[l] = al_new;
        const Real vbv = dt5 * (cu.vcm + cu.vcc - (uc_prev[l] + cu.ucc) * cu.co) * cu.rs;
        const Real ubv = dt5 * (uc_prev[l] + cu.ucc - (cu.vcm + cu.vcc) * cu.co) * cu.rs;
        vfl_[l] = ppm_face_cfl(cv[l], co, vbv, ry_prev[l], cu.ry);
        cv[l] = co;
        uc_prev[l] = cu.ucc;
        ry_prev[l] = cu.ry;
        vbv_[l] = vbv;
        ubv_[l] = ubv;
        uu_[l] = cu.uu;
        rx_[l] = cu.rx;
        rxm_[l] = cu.rxm;
        lu[3 + lane] = cu.uu;
        if (VORT) {
          const Real a1 = cu.uu * cu.dxr;
          va_[l] = a_prev[l];
          va1_[l] = a1;
          a_prev[l] = a1;
          ve_[l] = w2[l] * cu.dyc;  // (w2 = v of row r - 3 = jf - 1)
          vra_[l] = cu.rac;
        }
      }
      blk.wave_sync();
      const bool row_ok = jf >= j0 && jf <= j1;
      FV3_LANES(blk, lane, l) {
        const Real *a = lu + lane;  // a[0] = u(i-3, jf)
        const Real ufl = ppm_flux_int_cfl(a[0], a[1], a[2], a[3], a[4], a[5], ubv_[l], hord, rxm_[l], rx_[l]);
        if (row_ok && own[l]) (ke + b)[pcol[l] + (unsigned)(jf * sj32)] = (Real)0.5 * (vbv_[l] * vfl_[l] + ubv_[l] * ufl);
        if (VORT) {
          // wk = rarea * (u*dx - (u*dx)[j+1] - v*dy + (v*dy)[i+1]) of cell (i, jf - 1)
          const Real e1 = FV3_LANE_SHL(1, ve_, l, lane);
          const Real wkv = vra_[l] * (va_[l] - va1_[l] - ve_[l] + e1);
          if (row_ok && own[l]) {
            const unsigned pw = pcol[l] + (unsigned)((jf - 1) * sj32);
            (wk + b)[pw] = wkv;
            if (abs_on) (vabs + b)[pw] = wkv + f0b[pw];  // (the sponge layers only: f0 fetched where it is used, not with the row ahead)
          }
        }
      }
      blk.wave_sync();
    }
  });
}

static void ke_stream(fv3_ctx *c, fv3_stream_t s, const Real *u, const Real *v, const Real *uc, const Real *vc, Real *ke, Real dt, int hord, int k0, int k1,
                      Real *wk = nullptr, Real *vabs = nullptr, int fdw_k0 = 0) {
  if (wk)
    ke_stream_t<true>(c, s, u, v, uc, vc, ke, dt, hord, k0, k1, wk, vabs, fdw_k0);
  else
    ke_stream_t<false>(c, s, u, v, uc, vc, ke, dt, hord, k0, k1, nullptr, nullptr, 0);
}

// ---------------------------------------------------------------------------------------------
// divergence damping: the nord-fold Laplacian-type iteration of the corner divergence.
// Staged form (two launches per iteration, in place on divgd, uc / vc as work arrays exactly like
// the reference); `win` restricts every launch to a window (results exact >= 4 points inside an
// artificial window boundary) -- used for the cube-corner patches of the marching form.
// ---------------------------------------------------------------------------------------------
struct DdNone {};
// head / tail: extra stages run before / after the iteration in the same launch (window mode: the corner patches
// copy their window in and their result out there)
template <class Head, class Tail>
static void divdamp_staged_t(fv3_ctx *c, fv3_stream_t s, Real *divgd, Real *uc, Real *vc, int nord_max, int k0, int k1, const Wins *wins_, Head head, Tail tail) {
  const Geo g = c->g;
  Wins ws;
  ws.n = 0;
  if (wins_) ws = *wins_;
  const int nz1 = k1;
  // vc = d(divg)/dx * divg_u ; uc = d(divg)/dy * divg_v   (fill_corners via remapped reads when nt != 0)
  auto boxA = [=](int n) { const int ntm = nord_max - n; return Box{1 - 1 - ntm, g.nx + 1 + ntm, 1 - 1 - ntm, g.ny + 1 + ntm, k0, nz1}; };
  auto mkA = [=](int n) {
    return [=] FV3_HD(int t, int k, int i, int j) {
      const int nord = g.nord[k];
      if (n > nord) return;
      const int nt = nord - n;
      const int fl = g.flags[t];
      const long b = t * g.st + k * g.sk, m2 = t * g.st2;
      const Real *dg = divgd + b;
      const bool fill = nt != 0;
      if (i >= 1 - 1 - nt && i <= g.nx + 1 + nt && j >= 1 - nt && j <= g.ny + 1 + nt) {
        const Real a = fill ? dg[bc_index<1>(g, fl, i + 1, j)] : dg[IX(i + 1, j)];
        const Real e = fill ? dg[bc_index<1>(g, fl, i, j)] : dg[IX(i, j)];
        (vc + b)[IX(i, j)] = (a - e) * (g.divg_u + m2)[IX(i, j)];
      }
      if (i >= 1 - nt && i <= g.nx + 1 + nt && j >= 1 - 1 - nt && j <= g.ny + 1 + nt) {
        const Real a = fill ? dg[bc_index<2>(g, fl, i, j + 1)] : dg[IX(i, j + 1)];
        const Real e = fill ? dg[bc_index<2>(g, fl, i, j)] : dg[IX(i, j)];
        (uc + b)[IX(i, j)] = (a - e) * (g.divg_v + m2)[IX(i, j)];
      }
    };
  };
  auto boxB = [=](int n) { const int ntm = nord_max - n; return Box{1 - ntm, g.nx + 1 + ntm, 1 - ntm, g.ny + 1 + ntm, k0, nz1}; };
  auto mkB = [=](int n) {
    return [=] FV3_HD(int t, int k, int i, int j) {
      const int nord = g.nord[k];
      if (n > nord) return;
      const int nt = nord - n;
      if (i < 1 - nt || i > g.nx + 1 + nt || j < 1 - nt || j > g.ny + 1 + nt) return;
      const int fl = g.flags[t];
      const long b = t * g.st + k * g.sk, m2 = t * g.st2;
      const bool W = fl & FV3_W, E = fl & FV3_E, S = fl & FV3_S, N = fl & FV3_N;
      const int npx = g.npx, npy = g.npy;
      const bool fill = nt != 0;
      const Real *x = vc + b, *y = uc + b;
      auto UCR = [&](int ii, int jj) { return fill ? dg_y(x, y, g, fl, ii, jj, (Real)-1) : y[IX(ii, jj)]; };
      auto VCR = [&](int ii, int jj) { return fill ? dg_x(x, y, g, fl, ii, jj, (Real)-1) : x[IX(ii, jj)]; };
      Real d = UCR(i, j - 1) - UCR(i, j) + VCR(i - 1, j) - VCR(i, j);
      if (W && S && i == 1 && j == 1) d -= UCR(1, 0);
      if (E && S && i == npx && j == 1) d -= UCR(npx, 0);
      if (E && N && i == npx && j == npy) d += UCR(npx, npy);
      if (W && N && i == 1 && j == npy) d += UCR(1, npy);
      (divgd + b)[IX(i, j)] = d * (g.rarea_c + m2)[IX(i, j)];
    };
  };
  if constexpr (!std::is_same<Head, DdNone>::value) {
    if (ws.n > 0 && nord_max <= 3) {
      launch_chain(c, s, ws, k0, k1, head, chain_stage(boxA(1), mkA(1)), chain_stage(boxB(1), mkB(1)), chain_stage(boxA(2), mkA(2)), chain_stage(boxB(2), mkB(2)),
                   chain_stage(boxA(3), mkA(3)), chain_stage(boxB(3), mkB(3)), tail);
      return;
    }
    launch3w(c, s, head.nat, ws, head.f);
  }
  for (int n = 1; n <= nord_max; ++n) {
    launch3w(c, s, boxA(n), ws, mkA(n));
    launch3w(c, s, boxB(n), ws, mkB(n));
  }
  if constexpr (!std::is_same<Tail, DdNone>::value) launch3w(c, s, tail.nat, ws, tail.f);
}
static void divdamp_staged(fv3_ctx *c, fv3_stream_t s, Real *divgd, Real *uc, Real *vc, int nord_max, int k0, int k1, const Wins *wins_) {
  divdamp_staged_t(c, s, divgd, uc, vc, nord_max, k0, k1, wins_, DdNone{}, DdNone{});
}

// Marching form (see fv3_tp2d.hip).  A wave owns 58 corner columns and walks j; iteration n runs
// n rows behind the row being loaded and keeps a 3-row window of its input (with the i-neighbours
// of the middle row) in registers; one LDS line per iteration passes the freshly produced row to
// the neighbouring lanes.  divgd is read once, the damped divergence is written once to `out`.
// The cube-corner terms / halo remaps are not tile-local: an 8 x 8 patch per corner is recomputed
// with the staged form on a private copy and overwrites the marching result.
#define DD_OUT 58
#define DD_SEG 64
#define DD_NMAX 3
#define DD_PF 2
#define DD_PATCH WS_PATCH

static void divdamp_patches(fv3_ctx *c, fv3_stream_t s, const Real *divgd, Real *out, Real *uc, Real *vc, Real *tmp, int nord_max, int k0, int k1);
static void divdamp_stream(fv3_ctx *c, fv3_stream_t s, const Real *divgd, Real *out, Real *uc, Real *vc, Real *tmp, int nord_max, int k0, int k1) {
  const Geo g = c->g;
  const int nk = k1 - k0 + 1;
  const int nstrip = (g.nx + 1 + DD_OUT - 1) / DD_OUT;
  const int seg = fv3_pick_seg((long)nstrip * ((g.ny + 64) / 64) * g.nsub * nk, 4);
  const int nseg = (g.ny + 1 + seg - 1) / seg;
  const int LW = FV3_WAVE + 2;
  const size_t smem = sizeof(Real) * DD_NMAX * LW;
  const int nx = g.nx, ny = g.ny, nh = g.nh, sj32 = g.sj32, go = g.o;
  const long st = g.st, sk = g.sk, st2 = g.st2;
  const int *nord_k = g.nord;
  const MPtr divg_u = g.divg_u, divg_v = g.divg_v, rarea_c = g.rarea_c;
  // level-major launch geometry (see ke_stream): the three metric rows are 3 of the 4 row reads of a step
  static const int kb_env = getenv("FV3_KE_KB") ? atoi(getenv("FV3_KE_KB")) : 16;
  const int KB = kb_env > 0 ? (kb_env < nk ? kb_env : nk) : 0;
  const int nblk = KB ? (nk + KB - 1) / KB : 0;
  launch_waves<4>(c, s, KB ? KB : nstrip, KB ? nstrip * nseg : nseg, KB ? g.nsub * nblk : g.nsub * nk, smem, [=] FV3_HD(const Blk &blk_, char *smem_) {
    Blk blk = blk_;
    int t, k;
    if (KB) {
      t = blk_.bz / nblk;
      const int kk = (blk_.bz - t * nblk) * KB + blk_.bx;
      if (kk >= nk) return;
      k = k0 + kk;
      blk.by = blk_.by / nstrip;
      blk.bx = blk_.by - blk.by * nstrip;
    } else {
      t = blk.bz / nk;
      k = k0 + (blk.bz - t * nk);
    }
    const int nord = nord_k[k];
    if (nord == 0) return;
    const long b = t * st + k * sk, m2 = t * st2;
    const int i0 = 1 + blk.bx * DD_OUT;
    const int ja = 1 + blk.by * seg;
    const int jb = ja + seg - 1 < ny + 1 ? ja + seg - 1 : ny + 1;
    const int imax = nx + nh + 1, jsd = 1 - nh, jmax = ny + nh + 1;  // last stored corner column / row
    Real *ld = (Real *)smem_ + 1;  // ld[(n) * LW + lane]: row produced by iteration n (n = 0: the loaded row)
    const Real *dq = divgd + b;
    const MPtr dub = divg_u + m2, dvb = divg_v + m2, rab = rarea_c + m2;
    struct Row {
      Real d, du, dum, dv, ra;
    };
    Row pf[DD_PF][FV3_LPT];
    // window of the input of iteration n (n = 1..nord): rows rho-1 (a), rho (bL bC bR), rho+1 (cL cC cR)
    Real wa[DD_NMAX][FV3_LPT], wbL[DD_NMAX][FV3_LPT], wbC[DD_NMAX][FV3_LPT], wbR[DD_NMAX][FV3_LPT];
    Real wcL[DD_NMAX][FV3_LPT], wcC[DD_NMAX][FV3_LPT], wcR[DD_NMAX][FV3_LPT];
    Real ucp[DD_NMAX][FV3_LPT];  // uc of iteration n at row rho-1
    // metrics of row r - n (delay lines), index 0 = row being loaded
    Real mdu[DD_NMAX + 1][FV3_LPT], mdum[DD_NMAX + 1][FV3_LPT], mdv[DD_NMAX + 1][FV3_LPT], mra[DD_NMAX + 1][FV3_LPT];
    Real newrow[DD_NMAX][FV3_LPT];
    unsigned pcol[FV3_LPT];
    bool own[FV3_LPT];
    int r_beg = ja - nord, r_end = jb + nord;
    if (r_beg < jsd) r_beg = jsd;
    if (r_end > jmax) r_end = jmax;
    auto load_row = [&](int r, int l) -> Row {
      const unsigned p0 = pcol[l] + (unsigned)(r * sj32);
      Row w;
      w.d = dq[p0];
      w.du = dub[p0];
      w.dum = dub[p0 - (p0 != 0u)];
      w.dv = dvb[p0];
      w.ra = rab[p0];
      return w;
    };
    FV3_LANES(blk, lane, l) {
      const int i = i0 - 3 + lane, ic = i < imax ? i : imax;
      pcol[l] = (unsigned)(go * sj32 + ic + go);
      own[l] = i >= i0 && i < i0 + DD_OUT && i <= nx + 1;
#pragma unroll
      for (int n = 0; n < DD_NMAX; ++n) {
        wa[n][l] = wbL[n][l] = wbC[n][l] = wbR[n][l] = wcL[n][l] = wcC[n][l] = wcR[n][l] = ucp[n][l] = newrow[n][l] = (Real)0;
        if (lane == 0) {
          ld[n * LW - 1] = (Real)0;
          ld[n * LW + FV3_WAVE] = (Real)0;
        }
      }
#pragma unroll
      for (int n = 0; n <= DD_NMAX; ++n) mdu[n][l] = mdum[n][l] = mdv[n][l] = mra[n][l] = (Real)0;
#pragma unroll
      for (int n = 0; n < DD_PF; ++n) pf[n][l] = load_row(r_beg + n < r_end ? r_beg + n : r_end, l);
    }
    for (int r = r_beg; r <= r_end; ++r) {
      const int rn = r + DD_PF < r_end ? r + DD_PF : r_end;
      // ---- phase A: the new row of every iteration (own lane): iteration n produces row r - n
      FV3_LANES(blk, lane, l) {
        const Row cu = pf[0][l];
#pragma unroll
        for (int n = 0; n + 1 < DD_PF; ++n) pf[n][l] = pf[n + 1][l];
        pf[DD_PF - 1][l] = load_row(rn, l);
#pragma unroll
        for (int n = DD_NMAX; n >= 1; --n) {
          mdu[n][l] = mdu[n - 1][l];
          mdum[n][l] = mdum[n - 1][l];
          mdv[n][l] = mdv[n - 1][l];
          mra[n][l] = mra[n - 1][l];
        }
        mdu[0][l] = cu.du;
        mdum[0][l] = cu.dum;
        mdv[0][l] = cu.dv;
        mra[0][l] = cu.ra;
        newrow[0][l] = cu.d;
        ld[lane] = cu.d;
      }
      blk.wave_sync();
#pragma unroll
      for (int n = 1; n <= DD_NMAX; ++n) {
        if (n <= nord) {
          // the row produced by iteration n-1 at this step enters the window of iteration n as row c
          FV3_LANES(blk, lane, l) {
            wa[n - 1][l] = wbC[n - 1][l];
            wbL[n - 1][l] = wcL[n - 1][l];
            wbC[n - 1][l] = wcC[n - 1][l];
            wbR[n - 1][l] = wcR[n - 1][l];
            wcL[n - 1][l] = ld[(n - 1) * LW + lane - 1];
            wcC[n - 1][l] = newrow[n - 1][l];
            wcR[n - 1][l] = ld[(n - 1) * LW + lane + 1];
            // iteration n on row rho = r - n (window middle row b), metrics of that row = delay n
            const Real ucc = (wcC[n - 1][l] - wbC[n - 1][l]) * mdv[n][l];    // uc(i, rho)
            const Real vcm = (wbC[n - 1][l] - wbL[n - 1][l]) * mdum[n][l];   // vc(i-1, rho)
            const Real vcc = (wbR[n - 1][l] - wbC[n - 1][l]) * mdu[n][l];    // vc(i, rho)
            const Real dn_ = (ucp[n - 1][l] - ucc + vcm - vcc) * mra[n][l];
            ucp[n - 1][l] = ucc;
            if (n < DD_NMAX) {
              newrow[n][l] = dn_;
              ld[n * LW + lane] = dn_;
            }
            if (n == nord) {
              const int jo = r - n;
              if (jo >= ja && jo <= jb && own[l]) (out + b)[pcol[l] + (unsigned)(jo * sj32)] = dn_;
            }
          }
          blk.wave_sync();
        }
      }
    }
  });
  divdamp_patches(c, s, divgd, out, uc, vc, tmp, nord_max, k0, k1);
}

// The chain's cube-corner patches: the DD_PATCH^2 corners next to every cube corner of the context's sub-domains, recomputed by the staged form on a private
// copy (tmp) of a window around them and written into `out` (all corners in one chained launch when their windows are disjoint).  uc / vc: the work arrays
// of the iteration (the reference uses the dead C-grid winds; the fused wind stage, whose march still reads them, hands in scratch).
// the windows (patch + margin) around the cube corners of the context's sub-domains, the patches themselves, and the corner each belongs to
static void dd_windows(const Geo &g, int k0, int k1, Wins &wins, Wins &patches, int needs[4]) {
  int any = 0;
  for (int t = 0; t < g.nsub; ++t) any |= g.flags[t];
  const int P = DD_PATCH, M = 4;
  struct Corner {
    int need;
    bool west, south;
  };
  const Corner corners[4] = {{FV3_W | FV3_S, true, true}, {FV3_E | FV3_S, false, true}, {FV3_E | FV3_N, false, false}, {FV3_W | FV3_N, true, false}};
  wins.n = patches.n = 0;
  for (const Corner &cn : corners) {
    if ((any & cn.need) != cn.need) continue;
    Box w{cn.west ? -3 : g.nx + 1 - P - M, cn.west ? P + M : g.nx + 5, cn.south ? -3 : g.ny + 1 - P - M, cn.south ? P + M : g.ny + 5, k0, k1};
    Box pb{cn.west ? 1 : std::max(g.nx + 2 - P, 1), cn.west ? P : g.nx + 1, cn.south ? 1 : std::max(g.ny + 2 - P, 1), cn.south ? P : g.ny + 1, k0, k1};
    needs[wins.n] = cn.need;
    wins.w[wins.n++] = w;
    patches.w[patches.n++] = pb;
  }
}
// (a, b) -> (a2, b2) on the chain's windows: the fused wind stage runs the patch chain on scratch copies of the C-grid winds (its march still reads them) and puts
// the chain's work values back afterwards -- the reference's iteration leaves them in uc / vc (FVDynamics-Out carries uc / vc), and so does the staged form
// grow: the copy covers that many cells beyond each window (the chain reads its work arrays one cell outside its window -- stale values that only reach the
// discarded margin of its result, but do reach the work values it leaves: the scratch copies must hold what uc / vc hold there)
static void dd_copy_windows(fv3_ctx *c, fv3_stream_t s, const Real *a, const Real *b, Real *a2, Real *b2, int k0, int k1, int grow) {
  const Geo g = c->g;
  Wins wins, patches;
  int needs[4];
  dd_windows(g, k0, k1, wins, patches, needs);
  if (wins.n == 0) return;
  for (int w = 0; w < wins.n; ++w) {
    wins.w[w].i0 -= grow;
    wins.w[w].i1 += grow;
    wins.w[w].j0 -= grow;
    wins.w[w].j1 += grow;
  }
  auto body = [=] FV3_HD(int t, int k, int i, int j) {
    const long p = t * g.st + k * g.sk + IX(i, j);
    a2[p] = a[p];
    b2[p] = b[p];
  };
  const Box nat{1 - g.nh, g.nx + g.nh + 1, 1 - g.nh, g.ny + g.nh + 1, k0, k1};
  if (fv3_wins_disjoint(wins)) {  // (one launch for the four corners; tiny sub-domains, whose windows overlap: one corner at a time)
    launch3w(c, s, nat, wins, body);
    return;
  }
  for (int w = 0; w < wins.n; ++w) {
    Wins one;
    one.n = 1;
    one.w[0] = wins.w[w];
    launch3w(c, s, nat, one, body);
  }
}
static void divdamp_patches(fv3_ctx *c, fv3_stream_t s, const Real *divgd, Real *out, Real *uc, Real *vc, Real *tmp, int nord_max, int k0, int k1) {
  const Geo g = c->g;
  const int *nk_ = g.nord;
  Wins wins, patches;
  int needs[4];
  dd_windows(g, k0, k1, wins, patches, needs);
  if (wins.n == 0) return;
  auto run = [&](const Wins &ws, const Wins &ps, const int *nd) {
    // private copy of the windows (the staged form works in place), the iteration and the copy of the patches into
    // the result: one chained launch
    auto copy_in = [=] FV3_HD(int t, int k, int i, int j) {
      const long p = t * g.st + k * g.sk + IX(i, j);
      tmp[p] = divgd[p];
    };
    Wins psc = ps;
    int n0 = nd[0], n1 = ps.n > 1 ? nd[1] : 0, n2 = ps.n > 2 ? nd[2] : 0, n3 = ps.n > 3 ? nd[3] : 0;
    auto copy_out = [=] FV3_HD(int t, int k, int i, int j) {
      if (nk_[k] == 0) return;
      int need = 0;
      const int nn[4] = {n0, n1, n2, n3};
      for (int w = 0; w < psc.n; ++w)
        if (i >= psc.w[w].i0 && i <= psc.w[w].i1 && j >= psc.w[w].j0 && j <= psc.w[w].j1) need = nn[w];
      if (need == 0 || (g.flags[t] & need) != need) return;
      const long p = t * g.st + k * g.sk + IX(i, j);
      out[p] = tmp[p];
    };
    divdamp_staged_t(c, s, tmp, uc, vc, nord_max, k0, k1, &ws, chain_stage(Box{1 - g.nh, g.nx + g.nh + 1, 1 - g.nh, g.ny + g.nh + 1, k0, k1}, copy_in),
                     chain_stage(Box{1, g.nx + 1, 1, g.ny + 1, k0, k1}, copy_out));
  };
  if (fv3_wins_disjoint(wins)) {
    run(wins, patches, needs);
  } else {
    for (int w = 0; w < wins.n; ++w) {
      Wins one, pone;
      one.n = pone.n = 1;
      one.w[0] = wins.w[w];
      pone.w[0] = patches.w[w];
      run(one, pone, &needs[w]);
    }
  }
}

// Does d_sw take the accumulators' first-sub-step form (fv3_ctx::seq_acc_first) in this configuration?  (the fused scalar marches and fxadv do; the
// round-1 "separate" transports accumulate through tp2d's epilogue and do not) -- fv3_acoustic_step asks before it leaves out the four zero launches.
bool dsw_honors_acc_first(const fv3_ctx *c) {
  int nmax = 0;
  for (int k = 0; k < c->g.nz; ++k) nmax = std::max(nmax, std::max(c->nord_v_h[k], std::max(c->nord_w_h[k], c->nord_t_h[k])));
  const char *sc_env = getenv("FV3_DSW_SCALARS");
  return !(sc_env && !strcmp(sc_env, "separate")) && nmax <= 2 && c->zeros;
}

// Deferred accumulation of the Courant numbers (fv3_ctx::acc_slots): crx / cry of the n sub-steps of a call, each in its own array, summed ONCE into cx / cy on
// their write sets -- ((0 + s1) + s2) + ..., the association the read-modify-write of every sub-step has (0 + s1: what the first sub-step's store computes on the
// zeroed field, -0 included).  Per accumulator n reads and one write per call where fxadv read n - 1 times and wrote n times.  (The air-mass fluxes stay with the
// read-modify-write of the air-mass march: measured, taking it out saves the march 0.8 ms per sub-step and the sum of twelve more arrays costs 1.15.)
struct AccSlots {
  const Real *crx[FV3_ACC_MAXSTEPS], *cry[FV3_ACC_MAXSTEPS];
};
static int acc_sum(fv3_ctx *c, Real *cx, Real *cy, int n, void *stream) {
  if (n < 1 || n > FV3_ACC_MAXSTEPS || (int)c->acc_slots.size() < 2 * n) return fv3_fail(c, FV3_ERR_ARG, "acc_sum: no Courant-number slots for this many sub-steps");
  AccSlots sl;
  for (int q = 0; q < FV3_ACC_MAXSTEPS; ++q) {
    const int m = q < n ? q : n - 1;
    sl.crx[q] = c->acc_slots[2 * m];
    sl.cry[q] = c->acc_slots[2 * m + 1];
  }
  const Geo g = c->g;
  const int isd = 1 - g.nh, ied = g.nx + g.nh, jsd = 1 - g.nh, jed = g.ny + g.nh;
  launch3<4>(c, (fv3_stream_t)stream, Box{isd, ied, jsd, jed, 0, g.nz - 1}, [=] FV3_HD(int t, int k, int i, int j) {
    const long p = t * g.st + k * g.sk + IX(i, j);
    if (i >= 1 && i <= g.nx + 1) {  // cx: [1, nx+1] x [jsd, jed]
      Real a = (Real)0 + sl.crx[0][p];
#pragma unroll
      for (int q = 1; q < FV3_ACC_MAXSTEPS; ++q)
        if (q < n) a = a + sl.crx[q][p];
      cx[p] = a;
    }
    if (j >= 1 && j <= g.ny + 1) {  // cy: [isd, ied] x [1, ny+1]
      Real a = (Real)0 + sl.cry[0][p];
#pragma unroll
      for (int q = 1; q < FV3_ACC_MAXSTEPS; ++q)
        if (q < n) a = a + sl.cry[q][p];
      cy[p] = a;
    }
  });
  return fv3_post(c, (fv3_stream_t)stream, "acc_sum");
}

// Can d_sw leave cx / cy alone and keep the sub-step's Courant numbers in arrays of their own (fv3_ctx::seq_acc_defer)?  fxadv stores them exactly where it
// accumulates them; the round-1 ("separate") scalar form does not go through it with the sequencer's flags.
bool dsw_can_defer_acc(const fv3_ctx *c) { return dsw_honors_acc_first(c); }

// o_*: where the new delp / pt / w / q_con go.  Null: in place (the operator's own contract: the marches write beside the old
// fields, which their neighbours still read, and a copy-back follows); fv3_acoustic_step hands in the other half of its
// ping-pong pair instead and the copy-back disappears.
int fv3_d_sw_out(fv3_ctx *c, const fv3_field *delpc_, const fv3_field *delp_, const fv3_field *pt_, const fv3_field *u_, const fv3_field *v_,
                 const fv3_field *w_, const fv3_field *uc_, const fv3_field *vc_, const fv3_field *ua_, const fv3_field *va_,
                 const fv3_field *divgd_, const fv3_field *mfx_, const fv3_field *mfy_, const fv3_field *cx_, const fv3_field *cy_,
                 const fv3_field *crx_, const fv3_field *cry_, const fv3_field *xfx_, const fv3_field *yfx_, const fv3_field *q_con_,
                 const fv3_field *zh_, const fv3_field *heat_source_, const fv3_field *diss_est_, double dtd, void *stream,
                 const fv3_field *o_delp_, const fv3_field *o_pt_, const fv3_field *o_w_, const fv3_field *o_q_con_, int (*after_scalars)(void *), void *after_user) {
  if (!c) return FV3_ERR_ARG;
  FV3_FIELD(delpc, delpc_) FV3_FIELD(delp, delp_) FV3_FIELD(pt, pt_) FV3_FIELD(u, u_) FV3_FIELD(v, v_) FV3_FIELD(w, w_) FV3_FIELD(uc, uc_)
  FV3_FIELD(vc, vc_) FV3_FIELD(ua, ua_) FV3_FIELD(va, va_) FV3_FIELD(divgd, divgd_) FV3_FIELD(mfx, mfx_) FV3_FIELD(mfy, mfy_) FV3_FIELD(cx, cx_)
  FV3_FIELD(cy, cy_) FV3_FIELD(crx, crx_) FV3_FIELD(cry, cry_) FV3_FIELD(xfx, xfx_) FV3_FIELD(yfx, yfx_) FV3_FIELD(q_con, q_con_)
  FV3_FIELD(heat_source, heat_source_)
  (void)zh_;
  (void)diss_est_;  // accumulated only with do_skeb (unsupported); kept for signature parity
  const bool oop = o_delp_ != nullptr;  // out of place
  Real *o_delp = delp, *o_pt = pt, *o_w = w, *o_q_con = q_con;
  if (oop) {
    if (!o_pt_ || !o_w_ || !o_q_con_) return fv3_fail(c, FV3_ERR_ARG, "d_sw: the four output fields come together");
    o_delp = fv3_chk(c, o_delp_, "o_delp");
    o_pt = fv3_chk(c, o_pt_, "o_pt");
    o_w = fv3_chk(c, o_w_, "o_w");
    o_q_con = fv3_chk(c, o_q_con_, "o_q_con");
    if (!o_delp || !o_pt || !o_w || !o_q_con) return FV3_ERR_ARG;
    if (o_delp == delp || o_pt == pt || o_w == w || o_q_con == q_con) return fv3_fail(c, FV3_ERR_ARG, "d_sw: an output field aliases its input");
  }
  const Geo g = c->g;
  fv3_stream_t s = (fv3_stream_t)stream;
  const Real dt = (Real)dtd;
  const fv3_acoustic_config cf = c->cfg;
  const DampTables tab = c->tab;
  const int nz1 = g.nz - 1;
  const int isd = 1 - g.nh, ied = g.nx + g.nh, jsd = 1 - g.nh, jed = g.ny + g.nh;
  Real *ut = c->scratch[SC_A], *vt = c->scratch[SC_B], *fx = c->scratch[SC_C], *fy = c->scratch[SC_D], *gx = c->scratch[SC_E], *gy = c->scratch[SC_F];
  Real *heat_s = c->scratch[SC_H], *ke = c->scratch[SC_I], *wk = c->scratch[SC_J];
  int nord_max_v = 0, nord_max_w = 0, nord_max_t = 0, nord_max = 0;
  for (int k = 0; k < g.nz; ++k) {
    nord_max_v = std::max(nord_max_v, c->nord_v_h[k]);
    nord_max_w = std::max(nord_max_w, c->nord_w_h[k]);
    nord_max_t = std::max(nord_max_t, c->nord_t_h[k]);
    nord_max = std::max(nord_max, c->nord_h[k]);
  }

  // The del-n damping chains of delp / w / q_con / pt are bandwidth-bound and need only the field itself;
  // the transports are issue-bound.  The chains therefore run on the auxiliary stream, one transport
  // ahead (two pairs of flux arrays, A and B), and overlap with fxadv and the transports on the main stream.
  //   events: 0 fork | 1..4 chain of delp, w, q_con, pt ready | 5 pair A consumed | 6 pair B consumed
  fv3_stream_t s2 = fv3_aux(c, s);
  Real *dA_x = c->scratch[SC_DN_FX], *dA_y = c->scratch[SC_DN_FY], *dB_x = c->scratch[SC_S], *dB_y = c->scratch[SC_T], *d2w = c->scratch[SC_DN_D2];
  Deln dn_vt{g.nord_v, tab.tp_vt, g.damp_vt, 0, (Real)0, false, (Real)1.0e-4, nord_max_v};
  Deln dn_w{g.nord_w, tab.d6_w, g.damp_w, 0, (Real)0, false, (Real)1.0e-5, nord_max_w};
  Deln dn_t{g.nord_t, tab.tp_t, g.damp_t, 0, (Real)0, false, (Real)1.0e-4, nord_max_t};
  // FV3_DSW_SCALARS=separate: the four transports as four launches + the division kernel (round-1 form, A/B reference)
  const char *sc_env = getenv("FV3_DSW_SCALARS");  // (read per call: the A/B parity test flips it in one process)
  const bool fused_scalars = !(sc_env && !strcmp(sc_env, "separate")) && nord_max_v <= 2 && nord_max_t <= 2 && nord_max_w <= 2;
  if (c->seq_acc_first && !fused_scalars) return fv3_fail(c, FV3_ERR_ARG, "d_sw: the sequencer's first-sub-step form of the accumulators needs the fused scalar marches");
  const int scalars_mode = sc_env && !strcmp(sc_env, "quad") ? 0 : 1;
  // deferred accumulation of the Courant numbers (fv3_step.hip): cx / cy are not touched by fxadv; crx / cry are this sub-step's own arrays (the sequencer hands them in)
  const bool acc_defer = c->seq_acc_defer;
  if (acc_defer && !fused_scalars) return fv3_fail(c, FV3_ERR_ARG, "d_sw: the sequencer's deferred accumulation needs the fused scalar marches (fxadv with the Courant numbers)");
  if (fused_scalars) {
    // ---- air mass, vertical velocity, condensate, potential temperature: the four del-n chains (bandwidth-bound, the
    //      fields themselves are their only input) first, fxadv beside them, then ONE march for the four transports, the
    //      divisions by the new air mass and w's damping increment / heat (fv3_tp4.hip).  The march reads the old
    //      fields through its halo columns / rows while it produces the new ones, so it writes beside them.
    Real *dQ_x = c->scratch[SC_C], *dQ_y = c->scratch[SC_D], *dC_x = c->scratch[SC_G], *dC_y = c->scratch[SC_U];
    Real *n_dp = oop ? o_delp : c->scratch[SC_N], *n_w = oop ? o_w : c->scratch[SC_O], *n_qc = oop ? o_q_con : c->scratch[SC_P], *n_pt = oop ? o_pt : c->scratch[SC_Q];
    // Levels from fd_k0 on (every chain switched on and of order 2: all but the sponge layers) run the chains INSIDE the marches
    // (fv3_tp4.hip, FD); only the faces on the cube-corner patches are computed here for them.  FV3_DSW_DELN=arrays: the
    // chains of every level as del6_stream launches (round-2 form, A/B reference).
    const char *dn_env = getenv("FV3_DSW_DELN");  // (read per call: the A/B parity test flips it in one process)
    int fd_k0 = g.nz;
    if (!(dn_env && !strcmp(dn_env, "arrays")) && scalars_mode == 1) {
      for (int k = g.nz - 1; k >= 0; --k) {
        const bool all2 = c->nord_v_h[k] == 2 && c->nord_w_h[k] == 2 && c->nord_t_h[k] == 2 && c->damp_vt_h[k] > 1.0e-4 && c->damp_w_h[k] > 1.0e-5 && c->damp_t_h[k] > 1.0e-4;
        if (!all2) break;
        fd_k0 = k;
      }
    }
    if (getenv("FV3_DEBUG_FD")) fprintf(stderr, "[d_sw] fd_k0 = %d of %d\n", fd_k0, g.nz);
    fv3_signal(c, s, 0);
    fxadv(c, s, uc, vc, crx, cry, xfx, yfx, ut, vt, dt, acc_defer ? nullptr : cx, acc_defer ? nullptr : cy, true);
    fv3_signal(c, s, 5);  // (fxadv done: what the wind branch on the auxiliary stream waits for, see below)
    fv3_wait(c, s2, 0);
    // (the patches first: they are all the marches of the levels from fd_k0 on wait for; the chains of the sponge layers are read by
    //  the sponge-layer marches, which follow them on the auxiliary stream)
    del6_vt_flux_patches(c, s2, delp, d2w, dA_x, dA_y, dn_vt, false, fd_k0, nz1);
    del6_vt_flux_patches(c, s2, w, d2w, dC_x, dC_y, dn_w, false, fd_k0, nz1);
    del6_vt_flux_patches(c, s2, q_con, d2w, dQ_x, dQ_y, dn_t, true, fd_k0, nz1);
    del6_vt_flux_patches(c, s2, pt, d2w, dB_x, dB_y, dn_vt, true, fd_k0, nz1);
    const bool split_levels = s2 != s && scalars_mode == 1 && fd_k0 > 0 && fd_k0 <= nz1;  // (dsw_scalars_stream puts the sponge layers on s2)
    if (split_levels) fv3_signal(c, s2, 1);
    del6_vt_flux(c, s2, delp, d2w, dA_x, dA_y, dn_vt, false, 0, fd_k0 - 1);
    del6_vt_flux(c, s2, w, d2w, dC_x, dC_y, dn_w, false, 0, fd_k0 - 1);
    del6_vt_flux(c, s2, q_con, d2w, dQ_x, dQ_y, dn_t, true, 0, fd_k0 - 1);
    del6_vt_flux(c, s2, pt, d2w, dB_x, dB_y, dn_vt, true, 0, fd_k0 - 1);
    if (!split_levels) fv3_signal(c, s2, 1);
    fv3_wait(c, s, 1);
    DswScalars q4{delp, w, q_con, pt, n_dp, n_w, n_qc, n_pt, heat_s, crx, cry, xfx, yfx, mfx, mfy, gx, gy, dA_x, dA_y, dQ_x, dQ_y, dB_x, dB_y, dC_x, dC_y,
                  cf.hord_dp, cf.hord_vt, cf.hord_tm, dn_vt, dn_t, dt, dn_w, fd_k0};
    q4.acc_first = c->seq_acc_first;
    q4.zeros = c->zeros;
    dsw_scalars_stream(c, s, q4, scalars_mode);
    if (!oop) launch3<4>(c, s, Box{1, g.nx, 1, g.ny, 0, nz1}, [=] FV3_HD(int t, int k, int i, int j) {
      const long p = t * g.st + k * g.sk + IX(i, j);
      delp[p] = n_dp[p];
      pt[p] = n_pt[p];
      w[p] = n_w[p];
      q_con[p] = n_qc[p];
    });
  } else {
  fv3_signal(c, s, 0);
    fv3_wait(c, s2, 0);
    del6_vt_flux(c, s2, delp, d2w, dA_x, dA_y, dn_vt, false, 0, nz1);
    fv3_signal(c, s2, 1);
    // (w's damping fluxes get their own pair: the kernel that applies them runs after the transports)
    Real *dC_x = c->scratch[SC_G], *dC_y = c->scratch[SC_U];
    del6_vt_flux(c, s2, w, d2w, dC_x, dC_y, dn_w, false, 0, nz1);
    fv3_signal(c, s2, 2);

    fxadv(c, s, uc, vc, crx, cry, xfx, yfx, ut, vt, dt, cx, cy, true);

    // ---- air mass.  The flux-form updates (delp + div, delp * q + div) are formed inside the transport
    //      kernel (TpEpi); the tracer fluxes gx / gy never reach memory.
    Real *dpn = c->scratch[SC_N], *w_dp = c->scratch[SC_O], *qc_dp = c->scratch[SC_P], *pt_dp = c->scratch[SC_Q];
    {
      // cx += crx, cy += cry happen in fxadv; mfx += fx, mfy += fy in the store stage of this transport
      const TpEpi e{dpn, nullptr, true, mfx, mfy, nullptr, nullptr, nullptr, false, nullptr, nullptr, nullptr, dA_x, dA_y, nullptr, nullptr, nullptr, nullptr, nullptr};
      fv3_wait(c, s, 1);
      tp2d(c, s, delp, crx, cry, xfx, yfx, fx, fy, nullptr, nullptr, nullptr, cf.hord_dp, &dn_vt, 0, nz1, &e);
      fv3_signal(c, s, 5);
    }
    fv3_wait(c, s2, 5);
    del6_vt_flux(c, s2, q_con, d2w, dA_x, dA_y, dn_t, true, 0, nz1);
    fv3_signal(c, s2, 3);

    // ---- vertical velocity: transport with the mass fluxes; its del-n damping increment dw and the heat it
    //      dissipates are formed by the post-transport kernel below straight from the damping fluxes (the old w is
    //      still in place there), so neither a dw field nor a separate pass over the fluxes exists
    fv3_wait(c, s, 2);
    fv3_signal(c, s, 6);
    fv3_wait(c, s2, 6);
    del6_vt_flux(c, s2, pt, d2w, dB_x, dB_y, dn_vt, true, 0, nz1);
    fv3_signal(c, s2, 4);
    {
      const TpEpi e{w_dp, delp, false, nullptr, nullptr, nullptr, nullptr, nullptr, false, nullptr, nullptr, nullptr, nullptr, nullptr};  // delp * w + div
      tp2d(c, s, w, crx, cry, xfx, yfx, gx, gy, fx, fy, nullptr, cf.hord_vt, nullptr, 0, nz1, &e);
    }
    // ---- condensate
    {
      const TpEpi e{qc_dp, delp, false, nullptr, nullptr, nullptr, nullptr, nullptr, false, nullptr, nullptr, nullptr, dA_x, dA_y};
      fv3_wait(c, s, 3);
      tp2d(c, s, q_con, crx, cry, xfx, yfx, gx, gy, fx, fy, delp, cf.hord_dp, &dn_t, 0, nz1, &e);
    }
    // ---- potential temperature, then the divisions by the new air mass
    {
      const TpEpi e{pt_dp, delp, false, nullptr, nullptr, nullptr, nullptr, nullptr, false, nullptr, nullptr, nullptr, dB_x, dB_y};
      fv3_wait(c, s, 4);  // (also the join: nothing is left on the auxiliary stream after this chain)
      tp2d(c, s, pt, crx, cry, xfx, yfx, gx, gy, fx, fy, delp, cf.hord_tm, &dn_vt, 0, nz1, &e);
    }
    launch3(c, s, Box{1, g.nx, 1, g.ny, 0, nz1}, [=] FV3_HD(int t, int k, int i, int j) {
      const long b = t * g.st + k * g.sk;
      const unsigned q = IX(i, j);
      const long p = b + q;
      const Real dpnv = dpn[p];
      o_delp[p] = dpnv;
      o_pt[p] = pt_dp[p] / dpnv;
      Real wn = w_dp[p] / dpnv, hs = (Real)0;
      if (g.damp_w[k] > (Real)1.0e-5) {
        const Real dd8 = g.ke_bg[k] * fabs(dt);
        const Real dwv = ((dC_x + b)[q] - (dC_x + b)[IX(i + 1, j)] + (dC_y + b)[q] - (dC_y + b)[IX(i, j + 1)]) * g.rarea[t * g.st2 + q];
        hs = dd8 - dwv * (w[p] + (Real)0.5 * dwv);  // (w[p]: still the pre-transport value)
        wn = wn + dwv;
      }
      o_w[p] = wn;
      heat_s[p] = hs;
      o_q_con[p] = qc_dp[p] / dpnv;
    });
  }

  // ---- The wind branch up to the divergence-damping coefficient: cell-mean vorticity, corner kinetic energy, divergence damping, corner
  //      interpolation of the vorticity.  It needs fxadv's contravariant winds and the D-grid winds, NOT the scalar transports -- and its
  //      kernels are bandwidth-bound thread-per-point / light marching kernels, while the scalar marches next to them are bound by
  //      instruction issue at two waves per SIMD.  Round 4: with an auxiliary stream it is queued THERE, behind the sponge-layer launches,
  //      and can run beside the scalar marches of the main stream (FV3_DSW_WIND_OVERLAP=1; default: in program order behind them, the round-3
  //      sequence; same values).  Arrays: it writes wk / vabs / ke / the damping field (scratch the marches do not touch: the del-n work
  //      array is free once the corner patches are done, and the sponge-layer chains that still use it precede the branch on the same
  //      stream), delpc, divgd and the dead C-grid winds; the flux slots SC_E / SC_F it would share with the air-mass fluxes are only used
  //      in the keep_uv_dx configuration, which keeps the program order.
  Real *vabs = c->scratch[SC_R];
  Real *vdamp = c->scratch[SC_DN_D2];  // damping field "vort" on corners
  int fdw_k0 = g.nz;
  bool keep_uv_dx = false;
  const char *he = getenv("FV3_DSW_HEAT");
  bool heat_in_march = false;  // the vorticity march forms the damping heat (set with fdw_k0 below)
  const fv3_stream_t s_main = s;
  bool sponge_forked = false;  // the sponge levels' wind chain runs on the auxiliary stream (forked at the top of wind_branch, joined after the vorticity march)
  int side_from = 1 << 30;  // first level whose damping-heat side copies the fused wind stage has stored (see WindStage::u_side)
  auto wind_branch = [&](fv3_stream_t s) {
  // ---- cell-mean relative vorticity (+ absolute vorticity)
  // Levels that form the damping heat (d_con > 1e-5) WITHOUT vorticity damping (damp_vt <= 1e-5) read the
  // pre-update u*dx / v*dy where the damping fluxes would be (the work arrays keep those values in the
  // reference's data flow); only non-default configs (do_vort_damp off / vtdm4 = 0) have such levels.
  for (int k = 0; k < g.nz; ++k) keep_uv_dx = keep_uv_dx || (c->d_con_h[k] > 1.0e-5 && !(c->damp_vt_h[k] > 1.0e-5));
  Real *ut2 = c->scratch[SC_E], *vt2 = c->scratch[SC_F];  // = the utd / vtd slots below (damping fluxes overwrite them on damped levels)
  // Levels from fdw_k0 on (vorticity damping of order 2: all but the sponge layers): the vorticity transport loads wk and adds f0
  // itself and runs wk's del-n chain inside its march (tp2d TF_WIND | TF_FD) -- no absolute-vorticity field, no del6_stream launch
  // over those levels but for the tile-edge strips.  FV3_DSW_VORT_DELN=arrays: the round-2 form (A/B reference).
  {
    const char *e = getenv("FV3_DSW_VORT_DELN"), *m = getenv("FV3_TP2D_MODE"), *m6 = getenv("FV3_DEL6_MODE");
    const bool off = (e && !strcmp(e, "arrays")) || (m && !strcmp(m, "staged")) || (m6 && !strcmp(m6, "staged")) || keep_uv_dx;
    if (!off)
      for (int k = g.nz - 1; k >= 0; --k) {
        if (!(c->nord_v_h[k] == 2 && c->damp_vt_h[k] > 1.0e-5)) break;
        fdw_k0 = k;
      }
  }
  heat_in_march = cf.d_con > 1.0e-5 && !(he && !strcmp(he, "separate")) && fdw_k0 <= nz1 && tp2d_fd_lean(c, cf.hord_vt, fdw_k0, nz1);
  // Round 5, FV3_DSW_VORT_IN_KE=1 (experiment R5-30, off by default): the corner-KE march, which reads u and v anyway, forms the vorticity of the cells under
  // its corners -- columns 4 .. nx - 2, rows 3 .. ny - 3 at least -- and this launch only serves the four windows of the frame around them (sub-domains
  // without a tile edge on a side get those cells from both: the same values).  Same bits; measured neutral: the launch goes from 1.61 to 0.25 ms, the march
  // from 3.00 to 4.15 (23 more registers at four waves per SIMD, three more metric rows, one more store stream).
  const char *vk_env = getenv("FV3_DSW_VORT_IN_KE");  // (read per call: the parity test flips it in one process)
  const bool vort_in_ke = vk_env && vk_env[0] == '1' && !keep_uv_dx && getenv("FV3_KE_STAGED") == nullptr && g.nx >= 12 && g.ny >= 12;
  // Round 6: levels from kfz on run the FUSED wind stage (fv3_wind.hip: vorticity, corner KE, damping chain, corner interpolation and the damping in one
  // march); the launches below then only serve the levels under kfz (the sponge layers: no chain, absolute-vorticity field).  FV3_DSW_WINDSTAGE=staged:
  // every level through the staged kernels (A/B; read per call: the parity test flips it).
  int kfz = g.nz;
  {
    const char *we = getenv("FV3_DSW_WINDSTAGE");
    static const bool ke_staged_env = getenv("FV3_KE_STAGED") != nullptr, dd_staged_env = getenv("FV3_DIVDAMP_STAGED") != nullptr;
    const bool off = (we && !strcmp(we, "staged")) || ke_staged_env || dd_staged_env || keep_uv_dx || vort_in_ke || nord_max > DD_NMAX || nord_max <= 0 || g.nx < 8 || g.ny < 8;
    if (!off)
      for (int k = g.nz - 1; k >= fdw_k0; --k) {
        if (!(c->nord_h[k] >= 1 && c->nord_h[k] <= DD_NMAX)) break;
        kfz = k;
      }
  }
  const int ks1 = kfz - 1;  // last level of the staged kernels
  // Round 6: with the fused stage on every other level, the staged kernels only serve the sponge layers (3 of 79 levels: launches of a few dozen waves, 0.05 -
  // 0.3 ms each, ~0.8 ms in a row with the gaps between them).  Their whole chain -- vorticity, corner KE, divergence + damping, corner interpolation, the
  // vorticity's del-n fluxes, the vorticity transport with the wind update -- touches no level the march works on, so it goes to the auxiliary stream and runs
  // BESIDE the fused stage and the vorticity march (events 2 = fork here, 4 = the staged fields are there: levels from fdw_k0 on feed the vorticity march,
  // 3 = join after the march).  FV3_DSW_SPONGE_WIND=serial: program order (A/B; per call).
  fv3_stream_t ss = s;
  {
    const char *spe = getenv("FV3_DSW_SPONGE_WIND");
    if (kfz <= nz1 && ks1 >= 0 && fdw_k0 <= kfz && s == s_main && !(spe && !strcmp(spe, "serial"))) ss = fv3_aux(c, s);
    if (ss != s) {
      fv3_signal(c, s, 2);
      fv3_wait(c, ss, 2);
      sponge_forked = true;
    }
    if (getenv("FV3_DEBUG_FD")) fprintf(stderr, "[d_sw] sponge levels' wind chain: levels 0..%d (fdw_k0 %d) %s\n", ks1, fdw_k0, ss != s ? "on the auxiliary stream" : "in program order");
  }
  if (getenv("FV3_DEBUG_FD")) fprintf(stderr, "[d_sw] fused wind stage on levels %d..%d of %d\n", kfz, nz1, g.nz);
  // (two levels per thread: the six metric terms are read once)
  auto vort_cells = [=] FV3_HD(int t, int kp, int i, int j) {
    const long m2 = t * g.st2;
    const unsigned p = IX(i, j), pn = IX(i, j + 1), pe_ = IX(i + 1, j);
    const Real dx0 = (g.dx + m2)[p], dx1 = (g.dx + m2)[pn], dy0 = (g.dy + m2)[p], dy1 = (g.dy + m2)[pe_], ra = (g.rarea + m2)[p], f0v = (g.f0 + m2)[p];
#pragma unroll 1
    for (int kk = 0; kk < FV3_KC; ++kk) {
      const int k = FV3_KC * kp + kk;
      if (k > ks1) break;
      const long b = t * g.st + k * g.sk;
      // wk = rarea * (u*dx - (u*dx)[j+1] - v*dy + (v*dy)[i+1])
      const Real a = (u + b)[p] * dx0, a1 = (u + b)[pn] * dx1;
      const Real e = (v + b)[p] * dy0, e1 = (v + b)[pe_] * dy1;
      const Real wkv = ra * (a - a1 - e + e1);
      (wk + b)[p] = wkv;
      if (k < fdw_k0) (vabs + b)[p] = wkv + f0v;  // absolute vorticity for the transport below (formed on load from fdw_k0 on)
      if (keep_uv_dx) {
        (vt2 + b)[p] = a;
        (ut2 + b)[p] = e;
      }
    }
  };
  {
    const int nkc = (ks1 + FV3_KC) / FV3_KC;
    if (nkc <= 0) {
    } else if (vort_in_ke)
      launch_frame(c, ss, Frame{{Box{isd, 3, jsd, jed, 0, nkc - 1}, Box{g.nx - 1, ied, jsd, jed, 0, 0}, Box{4, g.nx - 2, jsd, 2, 0, 0}, Box{4, g.nx - 2, g.ny - 2, jed, 0, 0}}}, vort_cells);
    else
      launch3(c, ss, Box{isd, ied, jsd, jed, 0, nkc - 1}, vort_cells);
  }
  // ---- kinetic energy on corners (vb * ytp_v + ub * xtp_u)
  static const bool ke_staged = getenv("FV3_KE_STAGED") != nullptr;  // A/B switch for profiling
  // ke_value: one corner, any position (tile-edge forms of ub / vb, one-sided PPM, corner overrides)
  auto ke_value = [=] FV3_HD(int t, int k, int i, int j) -> Real {
    const int fl = g.flags[t];
    const long b = t * g.st + k * g.sk, m2 = t * g.st2;
    const bool W = fl & FV3_W, E = fl & FV3_E, S = fl & FV3_S, N = fl & FV3_N;
    const int npx = g.npx, npy = g.npy;
    const unsigned p = IX(i, j);
    const Real dt5 = (Real)0.5 * dt, dt4 = (Real)0.25 * dt;
    const Real *utl = ut + b, *vtl = vt + b, *ucl = uc + b, *vcl = vc + b, *ul = u + b, *vl = v + b;
    Real kev;
    const bool cW = W && i == 1, cE = E && i == npx, cS = S && j == 1, cN = N && j == npy;
    if ((cW || cE) && (cS || cN)) {
      const Real dt6 = dt / (Real)6.0;
      if (cW && cS)
        kev = dt6 * ((utl[IX(1, 1)] + utl[IX(1, 0)]) * ul[IX(1, 1)] + (vtl[IX(1, 1)] + vtl[IX(0, 1)]) * vl[IX(1, 1)] +
                     (utl[IX(1, 1)] + vtl[IX(1, 1)]) * ul[IX(0, 1)]);
      else if (cE && cS)
        kev = dt6 * ((utl[IX(i, 1)] + utl[IX(i, 0)]) * ul[IX(i - 1, 1)] + (vtl[IX(i, 1)] + vtl[IX(i - 1, 1)]) * vl[IX(i, 1)] +
                     (utl[IX(i, 1)] - vtl[IX(i - 1, 1)]) * ul[IX(i, 1)]);
      else if (cE && cN)
        kev = dt6 * ((utl[IX(i, j)] + utl[IX(i, j - 1)]) * ul[IX(i - 1, j)] + (vtl[IX(i, j)] + vtl[IX(i - 1, j)]) * vl[IX(i, j - 1)] +
                     (utl[IX(i, j - 1)] + vtl[IX(i - 1, j)]) * ul[IX(i, j)]);
      else
        kev = dt6 * ((utl[IX(1, j)] + utl[IX(1, j - 1)]) * ul[IX(1, j)] + (vtl[IX(1, j)] + vtl[IX(0, j)]) * vl[IX(1, j - 1)] +
                     (utl[IX(1, j - 1)] - vtl[IX(1, j)]) * ul[IX(0, j)]);
    } else {
      Real vbv, ubv;
      if (cS || cN)
        vbv = dt5 * (vtl[IX(i - 1, j)] + vtl[p]);
      else if (cW || cE)
        vbv = dt4 * (-vtl[IX(i - 2, j)] + (Real)3.0 * (vtl[IX(i - 1, j)] + vtl[p]) - vtl[IX(i + 1, j)]);
      else
        vbv = dt5 * (vcl[IX(i - 1, j)] + vcl[p] - (ucl[IX(i, j - 1)] + ucl[p]) * (g.cosa + m2)[p]) * (g.rsina + m2)[p];
      if (cW || cE)
        ubv = dt5 * (utl[IX(i, j - 1)] + utl[p]);
      else if (cS || cN)
        ubv = dt4 * (-utl[IX(i, j - 2)] + (Real)3.0 * (utl[IX(i, j - 1)] + utl[p]) - utl[IX(i, j + 1)]);
      else
        ubv = dt5 * (ucl[IX(i, j - 1)] + ucl[p] - (vcl[IX(i - 1, j)] + vcl[p]) * (g.cosa + m2)[p]) * (g.rsina + m2)[p];
      // ytp_v: advect v along y with vb ; xtp_u: advect u along x with ub
      Real vflux, uflux;
      {
        auto Q = [&](int s_) { return vl[IX(i, s_)]; };
        auto M = [&](int s_) { return (g.dy + m2)[IX(i, s_)]; };
        const bool zc = (W && i == 1) || (E && i == npx);  // tile corner columns of v
        const bool zm = zc && ((S && (j - 1 == 0 || j - 1 == 1)) || (N && (j - 1 == npy - 1 || j - 1 == npy)));
        const bool z0 = zc && ((S && (j == 0 || j == 1)) || (N && (j == npy - 1 || j == npy)));
        vflux = ppm_flux(Q, M, vbv, j, S, N, npy, cf.hord_mt, zm, z0, (g.rdy + m2)[IX(i, j - 1)], (g.rdy + m2)[p]);
      }
      {
        auto Q = [&](int s_) { return ul[IX(s_, j)]; };
        auto M = [&](int s_) { return (g.dx + m2)[IX(s_, j)]; };
        const bool zr = (S && j == 1) || (N && j == npy);
        const bool zm = zr && ((W && (i - 1 == 0 || i - 1 == 1)) || (E && (i - 1 == npx - 1 || i - 1 == npx)));
        const bool z0 = zr && ((W && (i == 0 || i == 1)) || (E && (i == npx - 1 || i == npx)));
        uflux = ppm_flux(Q, M, ubv, i, W, E, npx, cf.hord_mt, zm, z0, (g.rdx + m2)[IX(i - 1, j)], (g.rdx + m2)[p]);
      }
      kev = (Real)0.5 * (vbv * vflux + ubv * uflux);
    }
    return kev;
  };
  auto ke_point = [=] FV3_HD(int t, int k, int i, int j) { (ke + t * g.st + k * g.sk)[IX(i, j)] = ke_value(t, k, i, j); };
  if (ks1 < 0) {
  } else if (ke_staged) {
    launch3(c, ss, Box{1, g.nx + 1, 1, g.ny + 1, 0, ks1}, ke_point);
  } else {
    ke_stream(c, ss, u, v, uc, vc, ke, dt, cf.hord_mt, 0, ks1, vort_in_ke ? wk : nullptr, vabs, fdw_k0);
    // frame: the 3 outermost corner rows / columns next to a cube-tile edge
    // (W / E: columns 1..3 / npx-2..npx as narrow windows, S / N: rows 1..3 / npy-2..npy; the corner cells belong to the column windows)
    launch_frame_w(c, ss, Frame{{Box{1, 3, 1, g.ny + 1, 0, ks1}, Box{g.npx - 2, g.npx, 1, g.ny + 1, 0, 0}, Box{1, g.nx + 1, 1, 3, 0, 0}, Box{1, g.nx + 1, g.npy - 2, g.npy, 0, 0}}},
                   [=] FV3_HD(int w_, int t, int k, int i, int j) {
      const int fl = g.flags[t];
      if (!(fl & (w_ == 0 ? FV3_W : w_ == 1 ? FV3_E : w_ == 2 ? FV3_S : FV3_N))) return;
      if (w_ >= 2 && (((fl & FV3_W) && i <= 3) || ((fl & FV3_E) && i >= g.npx - 2))) return;  // covered by the column windows
      ke_point(t, k, i, j);
    });
  }

  // ---- divergence damping.  delpc: un-iterated divergence; divgd iterated in place; uc / vc are
  //      the work arrays of the iteration exactly as in the reference (their C-grid values are dead).
  // nord == 0 levels: divergence of the D-grid wind on the fly; nord > 0 levels: delpc = divgd
  static const bool staged_dd = getenv("FV3_DIVDAMP_STAGED") != nullptr;  // A/B switch for profiling
  // the marching iteration writes its result beside divgd, so the un-iterated divergence stays readable there and
  // the nord > 0 levels need no copy of it into delpc
  const bool dd_sep = !(staged_dd || nord_max > DD_NMAX);
  // (with the separate result array only the nord == 0 levels -- the sponge layers -- have work here: the launch covers their range)
  int kd0 = 0, kd1 = nz1;
  if (dd_sep) {
    kd0 = nz1 + 1;
    kd1 = -1;
    for (int k = 0; k <= nz1; ++k)
      if (c->nord_h[k] == 0) {
        kd0 = std::min(kd0, k);
        kd1 = std::max(kd1, k);
      }
  }
  // (ss: with the fused stage on, dd_sep holds and the levels kd0 .. kd1 -- those without the iteration -- are sponge levels)
  launch3(c, ss, Box{1, g.nx + 1, 1, g.ny + 1, kd0, kd1}, [=] FV3_HD(int t, int k, int i, int j) {
    const int fl = g.flags[t];
    const long b = t * g.st + k * g.sk, m2 = t * g.st2;
    const bool W = fl & FV3_W, E = fl & FV3_E, S = fl & FV3_S, N = fl & FV3_N;
    const int npx = g.npx, npy = g.npy;
    const unsigned p = IX(i, j);
    if (g.nord[k] != 0) {
      if (!dd_sep) (delpc + b)[p] = (divgd + b)[p];
      return;
    }
    auto PTC = [&](int ii, int jj) -> Real {
      const unsigned q = IX(ii, jj), qm = IX(ii, jj - 1);
      if ((S && jj == 1) || (N && jj == npy))
        return (vc + b)[q] * dt > (Real)0 ? (u + b)[q] * (g.dyc + m2)[q] * (g.sin_sg4 + m2)[qm] : (u + b)[q] * (g.dyc + m2)[q] * (g.sin_sg2 + m2)[q];
      return ((u + b)[q] - (Real)0.5 * ((va + b)[qm] + (va + b)[q]) * (g.cosa_v + m2)[q]) * (g.dyc + m2)[q] * (g.sina_v + m2)[q];
    };
    auto VRT = [&](int ii, int jj) -> Real {
      const unsigned q = IX(ii, jj), qm = IX(ii - 1, jj);
      if ((W && ii == 1) || (E && ii == npx))
        return (uc + b)[q] * dt > (Real)0 ? (v + b)[q] * (g.dxc + m2)[q] * (g.sin_sg3 + m2)[qm] : (v + b)[q] * (g.dxc + m2)[q] * (g.sin_sg1 + m2)[q];
      return ((v + b)[q] - (Real)0.5 * ((ua + b)[qm] + (ua + b)[q]) * (g.cosa_u + m2)[q]) * (g.dxc + m2)[q] * (g.sina_u + m2)[q];
    };
    Real d = VRT(i, j - 1) - VRT(i, j) + PTC(i - 1, j) - PTC(i, j);
    if (W && S && i == 1 && j == 1) d -= VRT(1, 0);
    if (E && S && i == npx && j == 1) d -= VRT(npx, 0);
    if (E && N && i == npx && j == npy) d += VRT(npx, npy);
    if (W && N && i == 1 && j == npy) d += VRT(1, npy);
    d = (g.rarea_c + m2)[p] * d;
    (delpc + b)[p] = d;
    const Real damp = g.da_min_c * fv3_max(g.d2_divg[k], fv3_min((Real)0.20, (Real)cf.dddmp * fabs(d * dt)));
    const Real vd = damp * d;
    (vdamp + b)[p] = vd;
    (ke + b)[p] += vd;
  });
  // divergence-damping iteration: divgd -> dnew (levels with nord > 0)
  Real *dnew = divgd;
  {
    if (!dd_sep) {
      divdamp_staged(c, s, divgd, uc, vc, nord_max, 0, nz1, nullptr);
    } else if (nord_max > 0) {
      dnew = c->scratch[SC_L];
      if (ks1 >= 0) divdamp_stream(c, ss, divgd, dnew, uc, vc, c->scratch[SC_M], nord_max, 0, ks1);
    }
  }
  // Smagorinsky-type coefficient from the corner-interpolated vorticity, levels with nord > 0
  // (wkb, the corner vorticity, is no longer a field: see the a2b epilogue below)
  // Corner vorticity (a2b_ord4 of wk) is consumed once, pointwise, by the Smagorinsky-type damping below: that
  // kernel runs as the epilogue of the corner interpolation and the corner field is never stored.
  {
    const Real dddmp = (Real)cf.dddmp;
    // The operator's contract leaves the iterated divergence in divgd (D_SW-Out carries it).  Inside fv3_acoustic_step nobody reads it: the
    // next c_sw overwrites the workspace field -- the sequencer says so (seq_divgd_dead) and the copy of the iteration's result into divgd
    // (one field write per call) is skipped when the iteration wrote beside it.
    const bool keep_divgd = !(c->seq_divgd_dead && dd_sep);
    if (ks1 >= 0) a2b_ord4_t<8>(c, ss, wk, 0, 0, ks1 + 1, (Real)1, [=] FV3_HD(int t, int k, unsigned p, Real wkbv) {
      const int nord = g.nord[k];
      if (nord == 0) return;
      const long b = t * g.st + k * g.sk;
      const Real dpc = dd_sep ? (divgd + b)[p] : (delpc + b)[p];  // the un-iterated divergence
      Real vo = (Real)0;
      if (dddmp >= (Real)1.0e-5) vo = fabs(dt) * sqrt(dpc * dpc + wkbv * wkbv);
      const Real damp2 = g.da_min_c * fv3_max(g.d2_divg[k], fv3_min((Real)0.20, dddmp * vo));
      const Real dn_ = (dnew + b)[p];
      if (keep_divgd) (divgd + b)[p] = dn_;  // (no-op for the in-place staged form; skipped inside the sequencer: see seq_divgd_dead)
      const Real vd = damp2 * dpc + tab.dd8[k] * dn_;
      (vdamp + b)[p] = vd;
      (ke + b)[p] += vd;
    });
    if (ss != s) fv3_signal(c, ss, 4);  // (ke, the damping field and wk of the staged levels are final)
    // ---- the fused wind stage on the levels kfz .. nz1 (fv3_wind.hip).  Order: the damping chain's cube-corner patches first (the march reads their values
    //      where its own chain is wrong; their work arrays are scratch here -- the march still reads the C-grid winds), the march, then two per-point launches on
    //      the three outermost corner rows / columns next to a cube-tile edge: kinetic energy (ke_point), then corner vorticity (a2b_point on the wk the march has
    //      stored) and the damping, with the iterated divergence the march exported there.
    if (kfz <= nz1) {
      Real *const wuc = c->scratch[SC_TP_FY2], *const wvc = c->scratch[SC_TP_FX2];
      dd_copy_windows(c, s, uc, vc, wuc, wvc, kfz, nz1, 2);
      divdamp_patches(c, s, divgd, dnew, wuc, wvc, c->scratch[SC_M], nord_max, kfz, nz1);
      WindStage ws{u, v, uc, vc, divgd, ke, vdamp, wk, dnew, tab.dd8, dt, dddmp, cf.hord_mt, keep_divgd, kfz, nz1};
      // The vorticity march's damping-heat epilogue wants copies of the winds on its segment / strip boundaries, made before it updates them in place: this
      // march has every row of u and v in registers, so it stores them (sx_side_copy then only serves the levels under kfz).  Not beside the scalar marches:
      // the arrays hold their new fields then.  FV3_DSW_SIDE=copy: the separate copies (A/B; read per call).
      const char *sd = getenv("FV3_DSW_SIDE");
      if (heat_in_march && s == s_main && !(sd && !strcmp(sd, "copy"))) {
        ws.u_side = c->scratch[SC_N];
        ws.v_side = c->scratch[SC_O];
        ws.side_seg = sx_march_seg(c, nz1 - fdw_k0 + 1);
        side_from = kfz;
      }
      if (getenv("FV3_DEBUG_FD")) fprintf(stderr, "[d_sw] side copies by the wind stage: %s (segment %d)\n", ws.u_side ? "yes" : "no", ws.side_seg);
      wind_stage_march(c, s, ws);
      // (two launches: with ke_value and a2b_point in one closure the geometry block went to scratch memory -- 1600 B per lane, 7 ms for these 3 % of the corners)
      launch_frame_w(c, s, Frame{{Box{1, 3, 1, g.ny + 1, kfz, nz1}, Box{g.npx - 2, g.npx, 1, g.ny + 1, 0, 0}, Box{1, g.nx + 1, 1, 3, 0, 0}, Box{1, g.nx + 1, g.npy - 2, g.npy, 0, 0}}},
                     [=] FV3_HD(int w_, int t, int k, int i, int j) {
        const int fl = g.flags[t];
        if (!(fl & (w_ == 0 ? FV3_W : w_ == 1 ? FV3_E : w_ == 2 ? FV3_S : FV3_N))) return;
        if (w_ >= 2 && (((fl & FV3_W) && i <= 3) || ((fl & FV3_E) && i >= g.npx - 2))) return;  // covered by the column windows
        ke_point(t, k, i, j);
      });
      const int nfr = g.nx + 1 > g.ny + 1 ? g.nx + 1 : g.ny + 1;
      launch3(c, s, Box{1, nfr, 1, 12, kfz, nz1}, [=] FV3_HD(int t, int k, int a_, int side) {
        const int fl = g.flags[t];
        int i, j;
        // side 1..3: columns 1..3 (W)   4..6: columns npx-2..npx (E)   7..9: rows 1..3 (S)   10..12: rows npy-2..npy (N)
        if (side <= 6) {
          if (a_ > g.ny + 1) return;
          if (!(fl & (side <= 3 ? FV3_W : FV3_E))) return;
          i = side <= 3 ? side : g.npx - 6 + side;
          j = a_;
        } else {
          if (a_ > g.nx + 1) return;
          if (!(fl & (side <= 9 ? FV3_S : FV3_N))) return;
          j = side <= 9 ? side - 6 : g.npy - 12 + side;
          i = a_;
          if (((fl & FV3_W) && i <= 3) || ((fl & FV3_E) && i >= g.npx - 2)) return;  // covered by the column sides
        }
        const long b = t * g.st + k * g.sk;
        const unsigned p = IX(i, j);
        const Real wkbv = a2b_point(g, wk + b, t, i, j);
        const Real dpc = (divgd + b)[p];
        Real vo = (Real)0;
        if (dddmp >= (Real)1.0e-5) vo = fabs(dt) * sqrt(dpc * dpc + wkbv * wkbv);
        const Real damp2 = g.da_min_c * fv3_max(g.d2_divg[k], fv3_min((Real)0.20, dddmp * vo));
        const Real vd = damp2 * dpc + tab.dd8[k] * (dnew + b)[p];
        (vdamp + b)[p] = vd;
        (ke + b)[p] += vd;
      });
      dd_copy_windows(c, s, wuc, wvc, uc, vc, kfz, nz1, 0);  // (the chain's work values, where the staged form leaves them)
      // (the operator's own contract leaves the iterated divergence in divgd; inside the sequencer nobody reads it)
      if (keep_divgd) launch3(c, s, Box{1, g.nx + 1, 1, g.ny + 1, kfz, nz1}, [=] FV3_HD(int t, int k, int i, int j) {
        const long pp = t * g.st + k * g.sk + IX(i, j);
        divgd[pp] = dnew[pp];
      });
    }
  }

  };
  // Measured (same box, alternating runs, C768): d_sw 51.6 -> 50.8 ms, but the halo copies that run inside d_sw 1.96 -> 2.5 and the sub-step
  // 113.6 -> 114.0 ms: the branch takes from the marches what it gains -- one d_sw call moves 252 GB in 51 ms = 4.9 TB/s, i.e. the operator
  // as a whole is at the bandwidth the chip sustains; overlapping its parts creates no capacity.  Off by default (FV3_DSW_WIND_OVERLAP=1).
  static const bool wind_overlap_off = !(getenv("FV3_DSW_WIND_OVERLAP") && getenv("FV3_DSW_WIND_OVERLAP")[0] == '1');
  for (int k = 0; k < g.nz; ++k) keep_uv_dx = keep_uv_dx || (c->d_con_h[k] > 1.0e-5 && !(c->damp_vt_h[k] > 1.0e-5));
  const bool wind_overlap = fused_scalars && s2 != s && !wind_overlap_off && !keep_uv_dx;
  if (wind_overlap) {
    fv3_wait(c, s2, 5);
    wind_branch(s2);
    fv3_signal(c, s2, 6);
  }

  // The new delp / w / q_con / pt are final here; the winds take the second half of the operator.  The sequencer starts the
  // halo update of delp / pt / q_con at this point (out-of-place form only: nothing below writes them, and the one later read --
  // the new delp of the damping-heat kernel -- is on compute cells, which a halo update does not touch), so that the exchange
  // overlaps the whole wind part instead of following the operator.
  if (after_scalars) {
    const int hst = after_scalars(after_user);
    if (hst != FV3_OK) return hst;
  }
  if (wind_overlap)
    fv3_wait(c, s, 6);
  else
    wind_branch(s);
  if (sponge_forked) fv3_wait(c, s, 4);  // (the staged levels from fdw_k0 on feed the march below: level 2 of the 79 -- no divergence-damping chain there, but the vorticity's)

  // ---- del-n damping fluxes of the relative vorticity (they depend on wk alone): ahead of the transport, whose wind
  //      epilogue applies them
  Real *utd = c->scratch[SC_E], *vtd = c->scratch[SC_F];  // (the tracer-flux slots: free since the tracer transports are done)
  {
    Deln dn_v{g.nord_v, tab.d6_vt, g.damp_vt, 0, (Real)0, false, (Real)1.0e-5, nord_max_v};
    del6_vt_flux(c, sponge_forked ? fv3_aux(c, s) : s, wk, c->scratch[SC_TP_QI], utd, vtd, dn_v, false, 0, fdw_k0 - 1);
    if (tp2d_fd_lean(c, cf.hord_vt, fdw_k0, nz1))  // (the round-5 march runs the chain on every strip: only the cube-corner patches come from the staged chain)
      del6_vt_flux_patches(c, s, wk, c->scratch[SC_TP_QI], utd, vtd, dn_v, false, fdw_k0, nz1);
    else
      del6_vt_flux_edge_strips(c, s, wk, c->scratch[SC_TP_QI], utd, vtd, dn_v, false, fdw_k0, nz1);
  }
  // ---- vorticity transport; the wind update u = u*dx + ke - ke[i+1] + fy, v = v*dy + ke - ke[j+1] - fx is the
  //      transport kernel's epilogue (the vorticity fluxes are never stored), and so is the vorticity damping
  //      u += vtd, v -= utd on the levels that have it; the winds BEFORE that damping -- what the damping heat is
  //      formed from -- go to scratch beside
  Real *u_pre = c->scratch[SC_N], *v_pre = c->scratch[SC_O];
  int heat_k1 = nz1;  // last level the damping-heat kernel serves
  {
    TpEpi e{nullptr, nullptr, false, nullptr, nullptr, u, v, ke, false, nullptr, nullptr, nullptr, nullptr, nullptr, vtd, utd, g.damp_vt, u_pre, v_pre};
    // (the levels without the chain: a small launch, on the auxiliary stream beside the others -- events 2 = fork, 3 = join)
    fv3_stream_t sv = fdw_k0 > 0 && fdw_k0 <= nz1 ? fv3_aux(c, s) : s;
    if (sv != s && !sponge_forked) {  // (forked: the stream already carries the sponge levels' chain, which is all this launch depends on)
      fv3_signal(c, s, 2);
      fv3_wait(c, sv, 2);
    }
    tp2d(c, sv, vabs, crx, cry, xfx, yfx, fx, fy, nullptr, nullptr, nullptr, cf.hord_vt, nullptr, 0, fdw_k0 - 1, &e);
    if (sv != s) fv3_signal(c, sv, 3);
    e.fd = 1;
    e.fd_coef = tab.d6_vt;
    e.fd_add = (const Real *)g.f0;
    // Round 5: on these levels the damping heat is the epilogue of the march (fv3_tp2x.hip, HEAT): the pre-damping winds and the two damping
    // increments are not stored and the damping-heat kernel below only serves the levels under fdw_k0.  FV3_DSW_HEAT=separate: the round-4
    // sequence (A/B; read per call).
    TpHeat th{vdamp, o_delp, heat_s, g.d_con, heat_source, c->seq_heat_first ? c->zeros : nullptr};
    th.side_from = side_from;
    if (heat_in_march) e.heat = &th;
    tp2d(c, s, wk, crx, cry, xfx, yfx, fx, fy, nullptr, nullptr, nullptr, cf.hord_vt, nullptr, fdw_k0, nz1, &e);
    // (whether the march took the heat over is what the dispatch DID, not what was predicted above: a form that ignores TpEpi::heat leaves every level to the kernel below)
    if (th.consumed) heat_k1 = fdw_k0 - 1;
    if (sv != s) fv3_wait(c, s, 3);
  }

  const bool heat_on = cf.d_con > 1.0e-5;
  const bool heat_first = c->seq_heat_first;  // (the sequencer's first sub-step of a call: heat_source holds nothing yet -- 0 + heat, the field is not read)
  // (two levels per thread: the six metric terms are read once)
  if (heat_k1 >= 0) launch3(c, s, Box{1, g.nx, 1, g.ny, 0, (heat_k1 + FV3_KC) / FV3_KC - 1}, [=] FV3_HD(int t, int kp, int i, int j) {
    const long m2 = t * g.st2;
    const unsigned p = IX(i, j), pn = IX(i, j + 1), pe_ = IX(i + 1, j), pne = IX(i + 1, j + 1);
    const Real rdx0 = (g.rdx + m2)[p], rdx1 = (g.rdx + m2)[pn], rdy0 = (g.rdy + m2)[p], rdy1 = (g.rdy + m2)[pe_];
    const Real rs2 = (g.rsin2 + m2)[p], cs = (g.cosa_s + m2)[p];
#pragma unroll 1
    for (int kk = 0; kk < FV3_KC; ++kk) {
      const int k = FV3_KC * kp + kk;
      if (k > heat_k1) break;
      const long b = t * g.st + k * g.sk;
      Real hs = (heat_s + b)[p];
      const Real dcon = g.d_con[k];
      if (dcon > (Real)1.0e-5) {
        // ub on (i, j), (i, j+1): (vort - vort[i+1] + vt) * rdx ; vb on (i, j), (i+1, j): (vort - vort[j+1] - ut) * rdy
        // (vtd / utd: damping fluxes, or the kept u*dx / v*dy on undamped levels)
        const Real vd00 = (vdamp + b)[p], vd10 = (vdamp + b)[pe_], vd01 = (vdamp + b)[pn], vd11 = (vdamp + b)[pne];
        const Real ub0 = (vd00 - vd10 + (vtd + b)[p]) * rdx0, ub1 = (vd01 - vd11 + (vtd + b)[pn]) * rdx1;
        const Real vb0 = (vd00 - vd01 - (utd + b)[p]) * rdy0, vb1 = (vd10 - vd11 - (utd + b)[pe_]) * rdy1;
        const Real fy0 = (u_pre + b)[p] * rdx0, fy1 = (u_pre + b)[pn] * rdx1;
        const Real fx0 = (v_pre + b)[p] * rdy0, fx1 = (v_pre + b)[pe_] * rdy1;
        const Real gy0 = fy0 * ub0, gy1 = fy1 * ub1, gx0 = fx0 * vb0, gx1 = fx1 * vb1;
        const Real u2 = fy0 + fy1, du2 = ub0 + ub1, v2 = fx0 + fx1, dv2 = vb0 + vb1;
        hs = (o_delp + b)[p] * (hs - (Real)0.25 * dcon * rs2 *
                                     ((ub0 * ub0 + ub1 * ub1 + vb0 * vb0 + vb1 * vb1) + (Real)2.0 * (gy0 + gy1 + gx0 + gx1) - cs * (u2 * dv2 + v2 * du2 + du2 * dv2)));
      }
      if (heat_on) FV3_ST_NT((heat_source + b)[p], (heat_first ? (Real)0 : (heat_source + b)[p]) + hs);
    }
  });
  // (deferred accumulation: the last d_sw of an acoustic call forms cx / cy from the sub-steps' Courant numbers)
  if (acc_defer && c->seq_acc_sum_n > 0) {
    const int st_ = acc_sum(c, cx, cy, c->seq_acc_sum_n, s);
    if (st_ != FV3_OK) return st_;
  }
  return fv3_post(c, s, "d_sw");
}

extern "C" int fv3_d_sw(fv3_ctx *c, const fv3_field *delpc_, const fv3_field *delp_, const fv3_field *pt_, const fv3_field *u_, const fv3_field *v_,
                        const fv3_field *w_, const fv3_field *uc_, const fv3_field *vc_, const fv3_field *ua_, const fv3_field *va_,
                        const fv3_field *divgd_, const fv3_field *mfx_, const fv3_field *mfy_, const fv3_field *cx_, const fv3_field *cy_,
                        const fv3_field *crx_, const fv3_field *cry_, const fv3_field *xfx_, const fv3_field *yfx_, const fv3_field *q_con_,
                        const fv3_field *zh_, const fv3_field *heat_source_, const fv3_field *diss_est_, double dtd, void *stream) {
  return fv3_d_sw_out(c, delpc_, delp_, pt_, u_, v_, w_, uc_, vc_, ua_, va_, divgd_, mfx_, mfy_, cx_, cy_, crx_, cry_, xfx_, yfx_, q_con_, zh_, heat_source_,
                      diss_est_, dtd, stream, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}
