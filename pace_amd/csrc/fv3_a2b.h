// fv3_a2b.h -- 4th-order cell-centre -> corner interpolation (a2b_ord4): the kernels, as a template over the
// output stage so that a pointwise consumer of the corner values can run as their epilogue (the corner field is
// then never stored).  fv3_a2b.hip instantiates the plain store.
// CPU twin: oracle/fv3_oracle/a2b_ord4.py.  [SURVEY A.13; reference operator AGrid2BGridFourthOrder]
//
// The reference runs ~10 dependent stencils (qx, qy, edge values, qxx, qyy, average); every one
// of them is a pure function of qin and 2-D metric terms, so a single launch evaluates the chain
// per output corner from L1/L2-resident qin -- no intermediate field ever reaches HBM.
#pragma once
#include "fv3_ops.h"

#define A2B_A1 ((Real)0.5625)
#define A2B_A2 ((Real)-0.0625)
#define A2B_B1 ((Real)(7.0 / 12.0))
#define A2B_B2 ((Real)(-1.0 / 12.0))
#define A2B_C1 ((Real)(2.0 / 3.0))
#define A2B_C2 ((Real)(-1.0 / 6.0))
#define A2B_R3 ((Real)(1.0 / 3.0))

namespace {

struct A2B {
  const Geo &g;     // (a reference: a by-value copy of the geometry block can end up in scratch memory -- 672 bytes per lane in fv3_pgf.hip's frame kernels)
  const Real *q;    // level base of qin for this (t, k)
  MPtr dxa;  // metric planes for this t
  MPtr dya;
  bool W, E, S, N;

  Real sc;          // input scale (1, or g for the interface heights of nh_p_grad: gz = g * zh is never stored)
  FV3_HD Real Q(int i, int j) const { return sc * q[IX(i, j)]; }

  FV3_HD Real qx_int(int i, int j) const { return A2B_B2 * (Q(i - 2, j) + Q(i + 1, j)) + A2B_B1 * (Q(i - 1, j) + Q(i, j)); }
  FV3_HD Real qy_int(int i, int j) const { return A2B_B2 * (Q(i, j - 2) + Q(i, j + 1)) + A2B_B1 * (Q(i, j - 1) + Q(i, j)); }

  FV3_HD Real qx_w1(int j) const {
    const Real g_in = dxa[IX(2, j)] / dxa[IX(1, j)], g_ou = dxa[IX(-1, j)] / dxa[IX(0, j)];
    return (Real)0.5 * ((((Real)2 + g_in) * Q(1, j) - Q(2, j)) / ((Real)1 + g_in) + (((Real)2 + g_ou) * Q(0, j) - Q(-1, j)) / ((Real)1 + g_ou));
  }
  FV3_HD Real qx_e1(int j) const {
    const int npx = g.npx;
    const Real g_in = dxa[IX(npx - 2, j)] / dxa[IX(npx - 1, j)], g_ou = dxa[IX(npx + 1, j)] / dxa[IX(npx, j)];
    return (Real)0.5 *
           ((((Real)2 + g_in) * Q(npx - 1, j) - Q(npx - 2, j)) / ((Real)1 + g_in) + (((Real)2 + g_ou) * Q(npx, j) - Q(npx + 1, j)) / ((Real)1 + g_ou));
  }
  FV3_HD Real qx(int i, int j) const {
    const int npx = g.npx;
    if (W) {
      if (i == 1) return qx_w1(j);
      if (i == 2) {
        const Real g_in = dxa[IX(2, j)] / dxa[IX(1, j)];
        return ((Real)3 * (g_in * Q(1, j) + Q(2, j)) - (g_in * qx_w1(j) + qx_int(3, j))) / ((Real)2 + (Real)2 * g_in);
      }
    }
    if (E) {
      if (i == npx) return qx_e1(j);
      if (i == npx - 1) {
        const Real g_in = dxa[IX(npx - 2, j)] / dxa[IX(npx - 1, j)];
        return ((Real)3 * (Q(npx - 2, j) + g_in * Q(npx - 1, j)) - (g_in * qx_e1(j) + qx_int(npx - 2, j))) / ((Real)2 + (Real)2 * g_in);
      }
    }
    return qx_int(i, j);
  }
  FV3_HD Real qy_s1(int i) const {
    const Real g_in = dya[IX(i, 2)] / dya[IX(i, 1)], g_ou = dya[IX(i, -1)] / dya[IX(i, 0)];
    return (Real)0.5 * ((((Real)2 + g_in) * Q(i, 1) - Q(i, 2)) / ((Real)1 + g_in) + (((Real)2 + g_ou) * Q(i, 0) - Q(i, -1)) / ((Real)1 + g_ou));
  }
  FV3_HD Real qy_n1(int i) const {
    const int npy = g.npy;
    const Real g_in = dya[IX(i, npy - 2)] / dya[IX(i, npy - 1)], g_ou = dya[IX(i, npy + 1)] / dya[IX(i, npy)];
    return (Real)0.5 *
           ((((Real)2 + g_in) * Q(i, npy - 1) - Q(i, npy - 2)) / ((Real)1 + g_in) + (((Real)2 + g_ou) * Q(i, npy) - Q(i, npy + 1)) / ((Real)1 + g_ou));
  }
  FV3_HD Real qy(int i, int j) const {
    const int npy = g.npy;
    if (S) {
      if (j == 1) return qy_s1(i);
      if (j == 2) {
        const Real g_in = dya[IX(i, 2)] / dya[IX(i, 1)];
        return ((Real)3 * (g_in * Q(i, 1) + Q(i, 2)) - (g_in * qy_s1(i) + qy_int(i, 3))) / ((Real)2 + (Real)2 * g_in);
      }
    }
    if (N) {
      if (j == npy) return qy_n1(i);
      if (j == npy - 1) {
        const Real g_in = dya[IX(i, npy - 2)] / dya[IX(i, npy - 1)];
        return ((Real)3 * (Q(i, npy - 2) + g_in * Q(i, npy - 1)) - (g_in * qy_n1(i) + qy_int(i, npy - 2))) / ((Real)2 + (Real)2 * g_in);
      }
    }
    return qy_int(i, j);
  }
  // values on the tile edge lines (linear in the along-edge direction)
  FV3_HD Real q2x(int ie_, int j) const {  // between cell columns ie_-1 and ie_
    return (Q(ie_ - 1, j) * dxa[IX(ie_, j)] + Q(ie_, j) * dxa[IX(ie_ - 1, j)]) / (dxa[IX(ie_ - 1, j)] + dxa[IX(ie_, j)]);
  }
  FV3_HD Real q1y(int i, int je_) const {
    return (Q(i, je_ - 1) * dya[IX(i, je_)] + Q(i, je_) * dya[IX(i, je_ - 1)]) / (dya[IX(i, je_ - 1)] + dya[IX(i, je_)]);
  }
  FV3_HD Real edge_x(int ie_, int j, Real w) const { return w * q2x(ie_, j - 1) + ((Real)1 - w) * q2x(ie_, j); }
  FV3_HD Real edge_y(int i, int je_, Real w) const { return w * q1y(i - 1, je_) + ((Real)1 - w) * q1y(i, je_); }
  FV3_HD Real qxx_int(int i, int j) const { return A2B_A2 * (qx(i, j - 2) + qx(i, j + 1)) + A2B_A1 * (qx(i, j - 1) + qx(i, j)); }
  FV3_HD Real qyy_int(int i, int j) const { return A2B_A2 * (qy(i - 2, j) + qy(i + 1, j)) + A2B_A1 * (qy(i - 1, j) + qy(i, j)); }
};

FV3_HD inline Real extrap(Real fac, Real q1, Real q2) { return q1 + fac * (q1 - q2); }

// one output corner, any position: tile-edge formulas included
// (always inlined: called out of line, the geometry block of the kernel's closure has to be addressable and is copied to scratch
//  memory first -- 672 bytes per lane, 3 waves per SIMD, 4 x the time in fv3_pgf.hip's frame kernels)
FV3_HD inline __attribute__((always_inline)) Real a2b_point(const Geo &g, const Real *qlev, int t, int i, int j, Real scale = (Real)1) {
  const int fl = g.flags[t];
  const A2B a{g, qlev, g.dxa + t * g.st2, g.dya + t * g.st2, (fl & FV3_W) != 0, (fl & FV3_E) != 0, (fl & FV3_S) != 0, (fl & FV3_N) != 0, scale};
  const int npx = g.npx, npy = g.npy;
  const bool onW = a.W && i == 1, onE = a.E && i == npx, onS = a.S && j == 1, onN = a.N && j == npy;
  const Real *ce = g.corner_extrap + t * 12;
  Real r;
  if ((!a.W || i >= 3) && (!a.E || i <= npx - 2) && (!a.S || j >= 3) && (!a.N || j <= npy - 2)) {
    // interior: branch-free, the same expressions the general path reduces to
    const Real *q = a.q;
    Real qx[4], qy[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int jj = j - 2 + n, ii = i - 2 + n;
      // (every input value scaled first, as Q() and the marching kernel do: the interior branch used to ignore `scale`, which no
      //  caller noticed -- the marching kernel serves the interior corners of the scaled field -- until fv3_pgf.hip evaluated
      //  interior corners of g * zh per point)
      qx[n] = A2B_B2 * (scale * q[IX(i - 2, jj)] + scale * q[IX(i + 1, jj)]) + A2B_B1 * (scale * q[IX(i - 1, jj)] + scale * q[IX(i, jj)]);
      qy[n] = A2B_B2 * (scale * q[IX(ii, j - 2)] + scale * q[IX(ii, j + 1)]) + A2B_B1 * (scale * q[IX(ii, j - 1)] + scale * q[IX(ii, j)]);
    }
    const Real qxx = A2B_A2 * (qx[0] + qx[3]) + A2B_A1 * (qx[1] + qx[2]);
    const Real qyy = A2B_A2 * (qy[0] + qy[3]) + A2B_A1 * (qy[1] + qy[2]);
    r = (Real)0.5 * (qxx + qyy);
  } else if (onW && onS) {
    r = (extrap(ce[0], a.Q(1, 1), a.Q(2, 2)) + extrap(ce[1], a.Q(0, 1), a.Q(-1, 2)) + extrap(ce[2], a.Q(1, 0), a.Q(2, -1))) * A2B_R3;
  } else if (onE && onS) {
    r = (extrap(ce[3], a.Q(npx - 1, 1), a.Q(npx - 2, 2)) + extrap(ce[4], a.Q(npx - 1, 0), a.Q(npx - 2, -1)) + extrap(ce[5], a.Q(npx, 1), a.Q(npx + 1, 2))) *
        A2B_R3;
  } else if (onE && onN) {
    r = (extrap(ce[6], a.Q(npx - 1, npy - 1), a.Q(npx - 2, npy - 2)) + extrap(ce[7], a.Q(npx, npy - 1), a.Q(npx + 1, npy - 2)) +
         extrap(ce[8], a.Q(npx - 1, npy), a.Q(npx - 2, npy + 1))) *
        A2B_R3;
  } else if (onW && onN) {
    r = (extrap(ce[9], a.Q(1, npy - 1), a.Q(2, npy - 2)) + extrap(ce[10], a.Q(0, npy - 1), a.Q(-1, npy - 2)) + extrap(ce[11], a.Q(1, npy), a.Q(2, npy + 1))) *
        A2B_R3;
  } else if (onW) {
    r = a.edge_x(1, j, g.edge_w[t * g.nj + j + g.o]);
  } else if (onE) {
    r = a.edge_x(npx, j, g.edge_e[t * g.nj + j + g.o]);
  } else if (onS) {
    r = a.edge_y(i, 1, g.edge_s[t * g.ni + i + g.o]);
  } else if (onN) {
    r = a.edge_y(i, npy, g.edge_n[t * g.ni + i + g.o]);
  } else {
    Real qxx, qyy;
    if (a.S && j == 2)
      qxx = A2B_C1 * (a.qx(i, 1) + a.qx(i, 2)) + A2B_C2 * (a.edge_y(i, 1, g.edge_s[t * g.ni + i + g.o]) + a.qxx_int(i, 3));
    else if (a.N && j == npy - 1)
      qxx = A2B_C1 * (a.qx(i, npy - 2) + a.qx(i, npy - 1)) + A2B_C2 * (a.edge_y(i, npy, g.edge_n[t * g.ni + i + g.o]) + a.qxx_int(i, npy - 2));
    else
      qxx = a.qxx_int(i, j);
    if (a.W && i == 2)
      qyy = A2B_C1 * (a.qy(1, j) + a.qy(2, j)) + A2B_C2 * (a.edge_x(1, j, g.edge_w[t * g.nj + j + g.o]) + a.qyy_int(3, j));
    else if (a.E && i == npx - 1)
      qyy = A2B_C1 * (a.qy(npx - 2, j) + a.qy(npx - 1, j)) + A2B_C2 * (a.edge_x(npx, j, g.edge_e[t * g.nj + j + g.o]) + a.qyy_int(npx - 2, j));
    else
      qyy = a.qyy_int(i, j);
    r = (Real)0.5 * (qxx + qyy);
  }
  return r;
}

}  // namespace

#define AB_OUT 61  // corners owned by a wave (64 columns of qin, 2 + 1 of them halo)
#define AB_PF 4    // rows of qin in flight ahead of the march

// Marching form (see fv3_tp2d.hip): a wave owns a strip of 61 corner columns and walks j.  Per row
// a lane loads one qin value (AB_PF rows ahead), forms the x-interpolated value of the new row and
// the y-interpolated value of the corner row from its 4-row register window, exchanges both with
// its i-neighbours through two LDS lines and stores one corner.  Corners within two points of a
// cube-tile edge use other formulas: the marching kernel skips them and a thin frame launch
// evaluates a2b_point there.
// epi(t, k_out, p, value): consumes the corner value of sub-domain t, output level k_out, in-plane offset p = IX(i, j);
// every corner 1..nx+1 x 1..ny+1 of every level is handed over exactly once (march + frame).  WPE: waves per SIMD the
// register allocation is sized for (8 for the plain store).
struct A2bStore {
  Real *out;
  long st, sk;
  FV3_HD void operator()(int t, int k, unsigned p, Real v) const { (out + t * st + k * sk)[p] = v; }
};
template <int WPE, class Epi>
static void a2b_ord4_t(fv3_ctx *c, fv3_stream_t s, const Real *qin, int kin0, int kout0, int nk, Real scale, Epi epi) {
  const Geo g = c->g;
  const int kshift = kout0 - kin0;
  const Geo *gp = c->g_dev;
  const int nx = g.nx, ny = g.ny, nh = g.nh, npx = g.npx, npy = g.npy, sj32 = g.sj32, go = g.o;
  const long st = g.st, sk = g.sk;
  const int nstrip = (nx + 1 + AB_OUT - 1) / AB_OUT;
  const int seg = fv3_pick_seg((long)nstrip * ((ny + 64) / 64) * g.nsub * nk, 8);
  int nseg = (ny + 1 + seg / 2) / seg;
  if (nseg < 1) nseg = 1;
  const int seglen = (ny + 1 + nseg - 1) / nseg;
  const size_t smem = sizeof(Real) * 2 * (FV3_WAVE + 3);
  launch_waves<WPE>(c, s, nstrip, nseg, g.nsub * nk, smem, [=] FV3_HD(const Blk &blk, char *smem_) {
    const int t = blk.bz / nk, k = kin0 + (blk.bz - t * nk);
    const int fl = gp->flags[t];
    const Real *q = qin + t * st + k * sk;
    // corners of this sub-domain that take the interior formula
    const int ia = (fl & FV3_W) ? 3 : 1, ib = (fl & FV3_E) ? npx - 2 : nx + 1;
    const int ja = (fl & FV3_S) ? 3 : 1, jb = (fl & FV3_N) ? npy - 2 : ny + 1;
    const int i0 = 1 + blk.bx * AB_OUT;
    int j0 = 1 + blk.by * seglen, j1 = j0 + seglen - 1;
    if (j0 < ja) j0 = ja;
    if (j1 > jb) j1 = jb;
    if (j0 > j1) return;
    const int ilo = i0 > ia ? i0 : ia, ihi = i0 + AB_OUT - 1 < ib ? i0 + AB_OUT - 1 : ib;
    if (ilo > ihi) return;
    Real *lq = (Real *)smem_ + 2;        // qin of the new row;        lq[lane] <-> column i0 - 2 + lane
    Real *ly = lq + FV3_WAVE + 3;        // y-interpolated corner row
    const int ied = nx + nh;
    Real qa[FV3_LPT], qb[FV3_LPT], qc[FV3_LPT], qd[FV3_LPT];  // qin rows r-3..r
    Real x0[FV3_LPT], x1[FV3_LPT], x2[FV3_LPT], x3[FV3_LPT];  // x-interpolated rows r-3..r
    Real pf[AB_PF][FV3_LPT];
    unsigned pcol[FV3_LPT];
    bool own[FV3_LPT];
    const int r_beg = j0 - 2, r_end = j1 + 1;
    FV3_LANES(blk, lane, l) {
      const int i = i0 - 2 + lane, ic = i < ied ? i : ied;
      pcol[l] = (unsigned)(go * sj32 + ic + go);
      own[l] = i >= ilo && i <= ihi;
      qa[l] = qb[l] = qc[l] = qd[l] = x0[l] = x1[l] = x2[l] = x3[l] = (Real)0;
      if (lane < 2) lq[lane - 2] = ly[lane - 2] = (Real)0;
      if (lane == 0) lq[FV3_WAVE] = ly[FV3_WAVE] = (Real)0;
#pragma unroll
      for (int n = 0; n < AB_PF; ++n) {
        const int rr = r_beg + n < r_end ? r_beg + n : r_end;
        pf[n][l] = q[pcol[l] + (unsigned)(rr * sj32)];
      }
    }
    for (int r = r_beg; r <= r_end; ++r) {
      const int rn = r + AB_PF < r_end ? r + AB_PF : r_end;
      FV3_LANES(blk, lane, l) {
        const Real qn = scale * pf[0][l];
#pragma unroll
        for (int n = 0; n + 1 < AB_PF; ++n) pf[n][l] = pf[n + 1][l];
        pf[AB_PF - 1][l] = q[pcol[l] + (unsigned)(rn * sj32)];
        qa[l] = qb[l];
        qb[l] = qc[l];
        qc[l] = qd[l];
        qd[l] = qn;
        lq[lane] = qn;
        ly[lane] = A2B_B2 * (qa[l] + qd[l]) + A2B_B1 * (qb[l] + qc[l]);  // corner row r-1
      }
      blk.wave_sync();
      const int j = r - 1;
      const bool row_ok = j >= j0 && j <= j1;
      FV3_LANES(blk, lane, l) {
        x0[l] = x1[l];
        x1[l] = x2[l];
        x2[l] = x3[l];
        x3[l] = A2B_B2 * (lq[lane - 2] + lq[lane + 1]) + A2B_B1 * (lq[lane - 1] + lq[lane]);
        const Real qxx = A2B_A2 * (x0[l] + x3[l]) + A2B_A1 * (x1[l] + x2[l]);
        const Real qyy = A2B_A2 * (ly[lane - 2] + ly[lane + 1]) + A2B_A1 * (ly[lane - 1] + ly[lane]);
        if (row_ok && own[l]) epi(t, k + kshift, pcol[l] + (unsigned)(j * sj32), (Real)0.5 * (qxx + qyy));
      }
      blk.wave_sync();
    }
  });
  // frame: the two outermost corner rows / columns on each side, wherever the sub-domain has a cube-tile edge there
  const int nfr = nx + 1 > ny + 1 ? nx + 1 : ny + 1;
  launch3(c, s, Box{1, nfr, 1, 8, kin0, kin0 + nk - 1}, [=] FV3_HD(int t, int k, int a, int side) {
    const int fl = g.flags[t];
    int i, j;
    // side 1,2: columns 1,2 (W)   3,4: columns npx-1, npx (E)   5,6: rows 1,2 (S)   7,8: rows npy-1, npy (N)
    if (side <= 4) {
      if (a > g.ny + 1) return;
      if (!(fl & (side <= 2 ? FV3_W : FV3_E))) return;
      i = side <= 2 ? side : g.npx - 4 + side;
      j = a;
    } else {
      if (a > g.nx + 1) return;
      if (!(fl & (side <= 6 ? FV3_S : FV3_N))) return;
      j = side <= 6 ? side - 4 : g.npy - 8 + side;
      i = a;
      // corners already covered by the column sides
      if (((fl & FV3_W) && i <= 2) || ((fl & FV3_E) && i >= g.npx - 1)) return;
    }
    const Real *q = qin + t * g.st + k * g.sk;
    epi(t, k + kshift, IX(i, j), a2b_point(g, q, t, i, j, scale));
  });
}

