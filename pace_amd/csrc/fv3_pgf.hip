// fv3_pgf.hip -- nh_p_grad as ONE marching kernel: the four corner interpolations (a2b_ord4 of pp, pk3, gz, delp) and the
// pressure-gradient update of the D-grid winds fused, so that the corner fields never exist.
// CPU twin: oracle/fv3_oracle/nh.py nh_p_grad.  [SURVEY A.10; reference operator NonHydrostaticPressureGradient,
// `nh_p_grad.py`; a2b_ord4: fv3_a2b.h]
//
// The staged form (fv3_nh.hip: four a2b_ord4 launches into scratch + a level-walking wind update) moves 16 field passes
// for 8 algorithmic ones: every corner field is written once and read back.  Here a wave owns a strip of 60 corner columns of
// ONE level and walks j like the a2b march; per row it interpolates the seven (field, interface) pairs the level needs --
// gz, pk3, pp at the upper and lower interface, delp -- from 4-row register windows (i-neighbours by wavefront shuffles),
// keeps the corner values of the previous row, and updates u / v of that row in place.  The lower interface of level k is the
// upper one of level k+1: both waves read it, the second read is served by L2 (level-major launch order).
// Corners whose stencil (their own, their east or their north neighbour's) reaches a tile-edge formula -- and, in the
// frame-first passes of the sequencer, the frame of every sub-domain -- are evaluated per point with a2b_point.
// Same expressions in the same order as the staged form: bitwise equal (FV3_NH_PGF=staged is the A/B switch).
#include "fv3_a2b.h"
#include "fv3_math.h"

#ifndef PG_WPE
#define PG_WPE (sizeof(Real) == 4 ? 3 : 2)  // waves per SIMD the fused march is sized for (fp32: three)
#endif

namespace {

#define PG_OUT 60  // corners owned by a wave: 64 columns of the inputs, 2 + 1 of them halo, one more for the east neighbour's corner
#define PG_PF 2
#define PG_NF 7    // gz(k), gz(k+1), pk3(k), pk3(k+1), pp(k), pp(k+1), delp(k)
#ifndef PG_KB
#define PG_KB 16   // levels of one tile that are consecutive workgroups of an XCD
#endif

struct PgfArgs {
  const Real *pp, *pk3, *gz, *delp;
  Real *u, *v;
  Real dt, top, gz_scale;
  int nz;
};

// the wind update at corner (i, j) of level k from the corner values of the seven pairs at p, pe = (i+1, j), pn = (i, j+1)
// (x0: upper interface k, x1: lower interface k+1)
struct PgfCorner {
  Real g0, g1, k0, k1, q0, q1, w;
};

FV3_HD inline void pgf_update(const Geo &g, const PgfArgs &a, int t, int k, int i, int j, unsigned p, const PgfCorner &c, const PgfCorner &e, const PgfCorner &n, Real rx,
                              Real ry) {
  const long b = t * g.st + k * g.sk;
  const Real wkp = c.k1 - c.k0;
  if (i <= g.nx) {
    const Real du = a.dt / (wkp + (e.k1 - e.k0)) * ((c.g1 - e.g0) * (e.k1 - c.k0) + (c.g0 - e.g1) * (c.k1 - e.k0));
    (a.u + b)[p] = ((a.u + b)[p] + du + a.dt / (c.w + e.w) * ((c.g1 - e.g0) * (e.q1 - c.q0) + (c.g0 - e.g1) * (c.q1 - e.q0))) * rx;
  }
  if (j <= g.ny) {
    const Real dv = a.dt / (wkp + (n.k1 - n.k0)) * ((c.g1 - n.g0) * (n.k1 - c.k0) + (c.g0 - n.g1) * (c.k1 - n.k0));
    (a.v + b)[p] = ((a.v + b)[p] + dv + a.dt / (c.w + n.w) * ((c.g1 - n.g0) * (n.q1 - c.q0) + (c.g0 - n.g1) * (c.q1 - n.q0))) * ry;
  }
}

// corners the march serves in sub-domain flags fl: its own, its east and its north neighbour's stencils are interior
FV3_HD inline void pgf_march_range(const Geo &g, int fl, int &ia, int &ib, int &ja, int &jb) {
  ia = (fl & FV3_W) ? 3 : 1;
  ib = (fl & FV3_E) ? g.npx - 3 : g.nx + 1;
  ja = (fl & FV3_S) ? 3 : 1;
  jb = (fl & FV3_N) ? g.npy - 3 : g.ny + 1;
}

}  // namespace

// pass (the sequencer's frame-first passes): 0 = every corner; 1 = the frame of every sub-domain only (FV3_FRAME_W wide, per point);
// 2 = the rest
void nh_pgf_fused(fv3_ctx *c, fv3_stream_t s, const Real *pp, const Real *pk3, const Real *gz, const Real *delp, Real *u, Real *v, Real dt, Real top, Real gz_scale,
                  int pass) {
  const Geo g = c->g;
  const PgfArgs a{pp, pk3, gz, delp, u, v, dt, top, gz_scale, g.nz};
  const int nz = g.nz;
  const int F = FV3_FRAME_W;
  const bool has_frame = (g.nx + 1 > 2 * F) && (g.ny + 1 > 2 * F);  // (launch3_pass: smaller boxes are all frame)
  // ---- per-point part: the corners next to the tile edges (pass 0), the sub-domain frames (pass 1) ----
  // Staged on thin bands: the corner values of the four fields on the bands into scratch, one a2b_point per corner and field
  // level, then the wind update from there.
  const MPtr rdx = g.rdx, rdy = g.rdy;
  Real *ppb = c->scratch[SC_B], *pk3b = c->scratch[SC_C], *gzb = c->scratch[SC_D], *wk1 = c->scratch[SC_A];
  const int nxp = g.nx + 1, nyp = g.ny + 1;
  auto winds_at = [=] FV3_HD(int t, int k, int i, int j) {
    const long b = t * g.st + k * g.sk, b1 = b + g.sk;
    const unsigned p = IX(i, j), pe_ = IX(i + 1, j), pn = IX(i, j + 1);
    const PgfCorner cc{(gzb + b)[p], (gzb + b1)[p], (pk3b + b)[p], (pk3b + b1)[p], (ppb + b)[p], (ppb + b1)[p], (wk1 + b)[p]};
    PgfCorner ce = cc, cn = cc;
    if (i <= g.nx) ce = PgfCorner{(gzb + b)[pe_], (gzb + b1)[pe_], (pk3b + b)[pe_], (pk3b + b1)[pe_], (ppb + b)[pe_], (ppb + b1)[pe_], (wk1 + b)[pe_]};
    if (j <= g.ny) cn = PgfCorner{(gzb + b)[pn], (gzb + b1)[pn], (pk3b + b)[pn], (pk3b + b1)[pn], (ppb + b)[pn], (ppb + b1)[pn], (wk1 + b)[pn]};
    pgf_update(g, a, t, k, i, j, p, cc, ce, cn, (rdx + t * g.st2)[p], (rdy + t * g.st2)[p]);
  };
  const bool edges_now = pass != 2;  // (tile edges are sub-domain edges: their bands belong to the frame pass)
  if (edges_now) {
    // Tile-edge frame corners (two columns / rows per side that has a tile edge -- the geometry of a2b_ord4_t's frame launch, lanes
    // along the edge).  The third corner column / row the band winds need has the interior formula: the march exports it.
    const int nfr = nxp > nyp ? nxp : nyp;
    // (one launch per field, one a2b_point per thread: four fields in one thread made a 3 ms kernel of what is 4 x 0.23 ms)
    auto sides = [=](const Real *qin, Real *out, int k0, int k1, Real scale) {
      launch3(c, s, Box{1, nfr, 1, 8, k0, k1}, [=] FV3_HD(int t, int k, int a_, int side) {
        const int fl = g.flags[t];
        int i, j;
        // side 1,2: columns 1,2 (W)   3,4: columns npx-1, npx (E)   5,6: rows 1,2 (S)   7,8: rows npy-1, npy (N)
        if (side <= 4) {
          if (a_ > nyp) return;
          if (!(fl & (side <= 2 ? FV3_W : FV3_E))) return;
          i = side <= 2 ? side : g.npx - 4 + side;
          j = a_;
        } else {
          if (a_ > nxp) return;
          if (!(fl & (side <= 6 ? FV3_S : FV3_N))) return;
          j = side <= 6 ? side - 4 : g.npy - 8 + side;
          i = a_;
          if (((fl & FV3_W) && i <= 2) || ((fl & FV3_E) && i >= g.npx - 1)) return;  // covered by the column sides
        }
        const long b = t * g.st + k * g.sk;
        (out + b)[IX(i, j)] = a2b_point(g, qin + b, t, i, j, scale);
      });
    };
    sides(a.gz, gzb, 0, nz, a.gz_scale);
    sides(a.pk3, pk3b, 1, nz, (Real)1);
    sides(a.pp, ppb, 1, nz, (Real)1);
    sides(a.delp, wk1, 0, nz - 1, (Real)1);
    // interface 0 of pk3 / pp: constants on every corner the band winds read (bands incl. their neighbour column / row)
    launch_frame(c, s, Frame{{Box{1, 3, 1, nyp, 0, 0}, Box{g.npx - 2, g.npx, 1, nyp, 0, 0}, Box{4, g.npx - 3, 1, 3, 0, 0}, Box{4, g.npx - 3, g.npy - 2, g.npy, 0, 0}}},
                 [=] FV3_HD(int t, int, int i, int j) {
      const long p = t * g.st + IX(i, j);
      pk3b[p] = a.top;
      ppb[p] = (Real)0;
    });
  }
  // ---- marching part ----
  const int nx = g.nx, ny = g.ny, nh = g.nh, sj32 = g.sj32, go = g.o;
  const long st = g.st, sk = g.sk, st2 = g.st2;
  const Geo *gp = c->g_dev;
  const int nstrip = (nx + 1 + PG_OUT - 1) / PG_OUT;
  const int seg = fv3_pick_seg((long)nstrip * ((ny + 64) / 64) * g.nsub * nz, 2);
  // sel: 0 = every strip, 1 = only the strips that hold columns of the sub-domain frame, 2 = only the others; rows jlo .. jhi
  auto march = [&](const int sel, const int jlo, const int jhi) {
  if (jlo > jhi) return;
  const int nrow = jhi - jlo + 1;
  int nseg = (nrow + seg / 2) / seg;
  if (nseg < 1) nseg = 1;
  const int seglen = (nrow + nseg - 1) / nseg;
  const bool exports = edges_now;
  // Level-major launch geometry: the workgroups an XCD walks are the levels of ONE (strip, segment) tile (PG_KB of them), then the
  // next tile -- the lower interface of level k is the upper one of level k+1, and adjacent levels resident together on an XCD read
  // it from its L2 (plane-major, consecutive levels land on different XCDs and every interface comes from HBM twice).
  const int nblk = (nz + PG_KB - 1) / PG_KB;
  launch_waves<PG_WPE>(c, s, PG_KB, nstrip * nseg, g.nsub * nblk, 0, [=] FV3_HD(const Blk &blk_, char *) {
    const int t = blk_.bz / nblk, k = (blk_.bz - t * nblk) * PG_KB + blk_.bx;
    if (k >= nz) return;
    Blk blk = blk_;
    blk.by = blk_.by / nstrip;
    blk.bx = blk_.by - blk.by * nstrip;
    const int fl = gp->flags[t];
    const long b = t * st + k * sk, m2 = t * st2;
    int ia, ib, ja, jb;
    pgf_march_range(*gp, fl, ia, ib, ja, jb);
    const int i0 = 1 + blk.bx * PG_OUT;
    if (sel != 0) {
      const bool fstrip = i0 <= F || i0 + PG_OUT - 1 >= nx + 2 - F;
      if ((sel == 1) != fstrip) return;
    }
    int j0 = jlo + blk.by * seglen, j1 = j0 + seglen - 1;
    if (j1 > jhi) j1 = jhi;
    if (j0 < ja) j0 = ja;
    if (j1 > jb) j1 = jb;
    if (j0 > j1) return;
    const int ilo = i0 > ia ? i0 : ia, ihi = i0 + PG_OUT - 1 < ib ? i0 + PG_OUT - 1 : ib;
    if (ilo > ihi) return;
    const int ied = nx + nh, jed = ny + nh;
    // the seven inputs: level bases and scales; pk3 / pp of interface 0 are constants (top, 0)
    const Real *src[PG_NF] = {a.gz + b, a.gz + b + sk, a.pk3 + b, a.pk3 + b + sk, a.pp + b, a.pp + b + sk, a.delp + b};
    const bool top_level = k == 0;
    Real qb[PG_NF][FV3_LPT], qc[PG_NF][FV3_LPT], qd[PG_NF][FV3_LPT];  // input rows r-2 .. r (r-3 .. r-1 before a step's shift)
    Real x1[PG_NF][FV3_LPT], x2[PG_NF][FV3_LPT], x3[PG_NF][FV3_LPT];  // x-interpolated rows r-2 .. r (likewise)
    // Input rows r .. r+2 in three register sets that rotate by the step's STATIC index (the march is unrolled by three): a rolled
    // rotation is a register copy of a load still in flight, i.e. a full memory latency per step at two waves per SIMD
    // (measured: 2.2 us per step, 6.1 ms for the launch).  The winds / metric terms of a row are requested one step early likewise.
    Real pf[3][PG_NF][FV3_LPT];
    Real sy[PG_NF][FV3_LPT];   // the y-interpolated corner row (read by the neighbouring lanes, like the new input row qd)
    Real cp[PG_NF][FV3_LPT], cc[PG_NF][FV3_LPT];   // corner values of rows r-2 (previous step) and r-1 (this step)
    Real o_u[FV3_LPT], o_v[FV3_LPT], o_rx[FV3_LPT], o_ry[FV3_LPT];  // winds / metric terms of the row the NEXT step updates (requested at the end of a step)
    unsigned pcol[FV3_LPT];
    bool own[FV3_LPT];
    const int r_beg = j0 - 2, r_end = j1 + 2 < jed ? j1 + 2 : jed;
    auto ld = [&](int f, int r, int l) -> Real {
      if (top_level && (f == 2 || f == 4)) return (Real)0;  // (never used: the constants replace the corner values)
      return FV3_EL(src[f], pcol[l] + (unsigned)(r * sj32));
    };
    FV3_LANES(blk, lane, l) {
      const int i = i0 - 2 + lane, ic = i < ied ? i : ied;
      pcol[l] = (unsigned)(go * sj32 + ic + go);
      own[l] = i >= ilo && i <= ihi;
      o_u[l] = o_v[l] = o_rx[l] = o_ry[l] = (Real)0;
#pragma unroll
      for (int f = 0; f < PG_NF; ++f) {
        qb[f][l] = qc[f][l] = qd[f][l] = x1[f][l] = x2[f][l] = x3[f][l] = sy[f][l] = cp[f][l] = cc[f][l] = (Real)0;
#pragma unroll
        for (int n = 0; n < 3; ++n) pf[n][f][l] = ld(f, r_beg + n < r_end ? r_beg + n : r_end, l);
      }
    }
    // step r with Q = (r - r_beg) mod 3: consumes register set Q (row r), refills it with row r+3; the winds of row r-2 were
    // requested at the end of the previous step, those of row r-1 are requested at the end of this one
    auto step = [&](int r, auto q_tag) {
      constexpr int Q = decltype(q_tag)::value;
      const int rn = r + 3 < r_end ? r + 3 : r_end;
      const int jw = r - 2;  // row whose winds this step updates
      const bool row_ok = jw >= j0 && jw <= j1;
      const bool next_ok = jw + 1 >= j0 && jw + 1 <= j1;
      // ---- phase A: new input row into the windows, y-interpolated corner row r-1
      FV3_LANES(blk, lane, l) {
#pragma unroll
        for (int f = 0; f < PG_NF; ++f) {
          const Real sc = f < 2 ? a.gz_scale : (Real)1;
          const Real qn = f < 2 ? sc * pf[Q][f][l] : pf[Q][f][l];
          pf[Q][f][l] = ld(f, rn, l);
          sy[f][l] = A2B_B2 * (qb[f][l] + qn) + A2B_B1 * (qc[f][l] + qd[f][l]);  // corner row r-1 from rows r-3 .. r
          qb[f][l] = qc[f][l];
          qc[f][l] = qd[f][l];
          qd[f][l] = qn;
        }
      }
      // ---- phase B: x-interpolated row r, corner values of row r-1
      FV3_LANES(blk, lane, l) {
#pragma unroll
        for (int f = 0; f < PG_NF; ++f) {
          const Real qm2 = FV3_LANE_SHR(2, qd[f], l, lane), qm1 = FV3_LANE_SHR(1, qd[f], l, lane), qp1 = FV3_LANE_SHL(1, qd[f], l, lane);
          const Real ym2 = FV3_LANE_SHR(2, sy[f], l, lane), ym1 = FV3_LANE_SHR(1, sy[f], l, lane), yp1 = FV3_LANE_SHL(1, sy[f], l, lane);
          const Real xn = A2B_B2 * (qm2 + qp1) + A2B_B1 * (qm1 + qd[f][l]);  // x-interpolated row r
          const Real qxx = A2B_A2 * (x1[f][l] + xn) + A2B_A1 * (x2[f][l] + x3[f][l]);
          x1[f][l] = x2[f][l];
          x2[f][l] = x3[f][l];
          x3[f][l] = xn;
          const Real qyy = A2B_A2 * (ym2 + yp1) + A2B_A1 * (ym1 + sy[f][l]);
          cp[f][l] = cc[f][l];
          cc[f][l] = (Real)0.5 * (qxx + qyy);
        }
        if (top_level) {  // interface 0: pk3 = ptop^kappa, pp = 0 on every corner
          cp[2][l] = cc[2][l] = a.top;
          cp[4][l] = cc[4][l] = (Real)0;
        }
        // the corner column / row next to a tile-edge frame goes to scratch for the band winds (interior formula: only the march has it)
        if (exports && (fl & 15)) {
          const int i = i0 - 2 + lane, jc = r - 1;
          const bool ecol = ((fl & FV3_W) && i == 3) || ((fl & FV3_E) && i == nx - 1);
          const bool erow = ((fl & FV3_S) && jc == 3) || ((fl & FV3_N) && jc == ny - 1);
          // (only corners that HAVE the interior formula: the frame kernel owns the others)
          const bool interior = i >= ((fl & FV3_W) ? 3 : 1) && i <= ((fl & FV3_E) ? nx - 1 : nx + 1) && jc >= ((fl & FV3_S) ? 3 : 1) && jc <= ((fl & FV3_N) ? ny - 1 : ny + 1);
          if ((ecol || erow) && interior && lane >= 2 && lane <= FV3_WAVE - 2 && jc >= j0 && jc <= j1 + 1) {
            const unsigned p = pcol[l] + (unsigned)(jc * sj32);
            FV3_EL(gzb + b, p) = cc[0][l];
            FV3_EL(gzb + b + sk, p) = cc[1][l];
            FV3_EL(pk3b + b, p) = cc[2][l];
            FV3_EL(pk3b + b + sk, p) = cc[3][l];
            FV3_EL(ppb + b, p) = cc[4][l];
            FV3_EL(ppb + b + sk, p) = cc[5][l];
            FV3_EL(wk1 + b, p) = cc[6][l];
          }
        }
      }
      // ---- phase C: the winds of row r-2 (corner values of that row: cp; of its north neighbour: cc; of its east neighbour: lane + 1's cp)
      FV3_LANES(blk, lane, l) {
        PgfCorner pc, pe_, pn;
        pc.g0 = cp[0][l]; pc.g1 = cp[1][l]; pc.k0 = cp[2][l]; pc.k1 = cp[3][l]; pc.q0 = cp[4][l]; pc.q1 = cp[5][l]; pc.w = cp[6][l];
        pn.g0 = cc[0][l]; pn.g1 = cc[1][l]; pn.k0 = cc[2][l]; pn.k1 = cc[3][l]; pn.q0 = cc[4][l]; pn.q1 = cc[5][l]; pn.w = cc[6][l];
        pe_.g0 = FV3_LANE_SHL(1, cp[0], l, lane);
        pe_.g1 = FV3_LANE_SHL(1, cp[1], l, lane);
        pe_.k0 = FV3_LANE_SHL(1, cp[2], l, lane);
        pe_.k1 = FV3_LANE_SHL(1, cp[3], l, lane);
        pe_.q0 = FV3_LANE_SHL(1, cp[4], l, lane);
        pe_.q1 = FV3_LANE_SHL(1, cp[5], l, lane);
        pe_.w = FV3_LANE_SHL(1, cp[6], l, lane);
        const int i = i0 - 2 + lane;
        const bool mine = row_ok && own[l];
        if (mine) {
          const unsigned p = pcol[l] + (unsigned)(jw * sj32);
          const Real wkp = pc.k1 - pc.k0;
          if (i <= nx) {
            const Real du = fv3_div(a.dt, wkp + (pe_.k1 - pe_.k0)) * ((pc.g1 - pe_.g0) * (pe_.k1 - pc.k0) + (pc.g0 - pe_.g1) * (pc.k1 - pe_.k0));
            FV3_EL(a.u + b, p) = (o_u[l] + du + fv3_div(a.dt, pc.w + pe_.w) * ((pc.g1 - pe_.g0) * (pe_.q1 - pc.q0) + (pc.g0 - pe_.g1) * (pc.q1 - pe_.q0))) * o_rx[l];
          }
          if (jw <= ny) {
            const Real dv = fv3_div(a.dt, wkp + (pn.k1 - pn.k0)) * ((pc.g1 - pn.g0) * (pn.k1 - pc.k0) + (pc.g0 - pn.g1) * (pc.k1 - pn.k0));
            FV3_EL(a.v + b, p) = (o_v[l] + dv + fv3_div(a.dt, pc.w + pn.w) * ((pc.g1 - pn.g0) * (pn.q1 - pc.q0) + (pc.g0 - pn.g1) * (pc.q1 - pn.q0))) * o_ry[l];
          }
        }
        if (next_ok) {  // (the winds of row r-1 and their metric terms: consumed in phase C of the next step)
          const unsigned p = pcol[l] + (unsigned)((jw + 1) * sj32);
          o_u[l] = FV3_EL(a.u + b, p);
          o_v[l] = FV3_EL(a.v + b, p);
          o_rx[l] = FV3_EL(rdx + m2, p);
          o_ry[l] = FV3_EL(rdy + m2, p);
        }
      }
    };
    // (trailing steps past r_end: their loads are clamped to r_end and every store is masked by the owned-row tests)
    for (int r = r_beg; r <= r_end; r += 3) {
      step(r, std::integral_constant<int, 0>{});
      step(r + 1, std::integral_constant<int, 1>{});
      step(r + 2, std::integral_constant<int, 2>{});
    }
  });
  };
  // Frame-first passes: the frame pass marches the two row bands of the frame over every strip and the strips that hold the
  // frame's columns over the rows between (whole strips: the march is 60 columns wide whatever it is asked for -- what it
  // computes beyond the frame is left out of the interior pass); the interior pass marches the remaining strips and rows.
  if (pass == 0 || !has_frame) {
    if (pass != 2) march(0, 1, ny + 1);
  } else if (pass == 1) {
    march(0, 1, F);
    march(0, ny + 2 - F, ny + 1);
    march(1, F + 1, ny + 1 - F);
  } else {
    march(2, F + 1, ny + 1 - F);
  }
  if (edges_now) {
    // the winds on the bands the march left out (columns 1, 2 / npx-2 .. npx, rows 1, 2 / npy-2 .. npy of the sub-domains with that
    // tile edge), from the frame corners and the corner column / row the march exported
    const Frame windsF{{Box{1, 2, 1, nyp, 0, nz - 1}, Box{g.npx - 2, g.npx, 1, nyp, 0, nz - 1}, Box{3, g.npx - 3, 1, 2, 0, nz - 1}, Box{3, g.npx - 3, g.npy - 2, g.npy, 0, nz - 1}}};
    launch_frame(c, s, windsF, [=] FV3_HD(int t, int k, int i, int j) {
      int ia, ib, ja, jb;
      pgf_march_range(g, g.flags[t], ia, ib, ja, jb);
      if (i >= ia && i <= ib && j >= ja && j <= jb) return;  // the march's
      winds_at(t, k, i, j);
    });
  }
}
