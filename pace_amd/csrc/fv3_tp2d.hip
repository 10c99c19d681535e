// fv3_tp2d.hip -- Lin-Rood 2-D flux-form transport (fv_tp_2d) and the del-n damping fluxes.
// CPU twin: oracle/fv3_oracle/fvtp2d.py.  [SURVEY A.4, A.4.3; reference operator
// FiniteVolumeTransport, REF examples/notebooks/functions.py:935-951]
//
// Two forms of each operator:
//  * staged (one launch per stage, intermediates in global scratch): the first correct version,
//    kept as the A/B reference (FV3_TP2D_MODE / FV3_DEL6_MODE = staged) and used on the small
//    cube-corner windows of the marching del-n kernel;
//  * marching wave kernels (tp2d_stream, del6_stream): the product path, see the comments there.
// LDS-tiled workgroup variants (64 x 8 / 64 x 16 tiles, 256 threads, per-level and k-walking)
// were measured in between and removed: 13.4 ms / 23 ms per transport against 8 ms for the
// marching form and 18 ms staged (C768, MI355X).
#include <type_traits>

#include "fv3_ops.h"
#include "fv3_ppm.h"

static inline Box clip(Box a, const Box *w) {
  if (!w) return a;
  Box r = a;
  r.i0 = std::max(a.i0, w->i0);
  r.i1 = std::min(a.i1, w->i1);
  r.j0 = std::max(a.j0, w->j0);
  r.j1 = std::min(a.j1, w->j1);
  return r;
}

// Stage-per-launch form (global intermediates).  `win` restricts every launch to a window: used
// for the few tiles next to a cube corner, where the corner-halo remap makes a tile-local
// evaluation awkward; results are valid 4 cells inside the window.
// tail: one more stage run after the chain in the same launch (window mode only; the corner patches copy their
// result into the flux fields there)
struct NoTail {};
template <class Tail>
static void del6_vt_flux_staged_t(fv3_ctx *c, fv3_stream_t s, const Real *q, Real *d2, Real *fx2, Real *fy2, const Deln &dn, bool q_raw, int k0, int k1,
                                  const Box *win_, const Wins *wins_, Tail tail) {
  Wins ws;
  ws.n = 0;
  if (wins_) ws = *wins_;
  else if (win_) {
    ws.n = 1;
    ws.w[0] = *win_;
  }
  const Geo g = c->g;
  const int nm = dn.nord_max;
  const Deln d = dn;
  // d2 = damp * q on (is-1-nord .. ie+1+nord)^2
  const Box box0{-nm, g.nx + 1 + nm, -nm, g.ny + 1 + nm, k0, k1};
  auto st0 = [=] FV3_HD(int t, int k, int i, int j) {
    if (!deln_on(d, k)) return;
    const int n = deln_nord(d, k);
    if (i < -n || i > g.nx + 1 + n || j < -n || j > g.ny + 1 + n) return;
    const long p = t * g.st + k * g.sk + IX(i, j);
    d2[p] = q_raw ? q[p] : deln_damp(d, k) * q[p];
  };
  // first fluxes (copy_corners only when nord > 0)
  const Box box1{1 - nm, g.nx + nm + 1, 1 - nm, g.ny + nm + 1, k0, k1};
  auto st1 = [=] FV3_HD(int t, int k, int i, int j) {
    if (!deln_on(d, k)) return;
    const int n = deln_nord(d, k);
    const int fl = g.flags[t];
    const long b = t * g.st + k * g.sk;
    const long m2 = t * g.st2;
    const Real *dd = d2 + b;
    if (i >= 1 - n && i <= g.nx + n + 1 && j >= 1 - n && j <= g.ny + n) {
      const Real a = n > 0 ? cc<1>(dd, g, fl, i - 1, j) : dd[IX(i - 1, j)];
      const Real e = n > 0 ? cc<1>(dd, g, fl, i, j) : dd[IX(i, j)];
      (fx2 + b)[IX(i, j)] = (g.del6_v + m2)[IX(i, j)] * (a - e);
    }
    if (i >= 1 - n && i <= g.nx + n && j >= 1 - n && j <= g.ny + n + 1) {
      const Real a = n > 0 ? cc<2>(dd, g, fl, i, j - 1) : dd[IX(i, j - 1)];
      const Real e = n > 0 ? cc<2>(dd, g, fl, i, j) : dd[IX(i, j)];
      (fy2 + b)[IX(i, j)] = (g.del6_u + m2)[IX(i, j)] * (a - e);
    }
  };
  auto boxA = [=](int n) { return Box{-(nm - n), g.nx + 1 + (nm - n), -(nm - n), g.ny + 1 + (nm - n), k0, k1}; };
  auto mkA = [=](int n) {
    return [=] FV3_HD(int t, int k, int i, int j) {
      if (!deln_on(d, k)) return;
      const int nord = deln_nord(d, k);
      if (n > nord) return;
      const int nt = nord - n;
      if (i < -nt || i > g.nx + 1 + nt || j < -nt || j > g.ny + 1 + nt) return;
      const long b = t * g.st + k * g.sk;
      const unsigned p = IX(i, j);
      (d2 + b)[p] = ((fx2 + b)[p] - (fx2 + b)[IX(i + 1, j)] + (fy2 + b)[p] - (fy2 + b)[IX(i, j + 1)]) * g.rarea[t * g.st2 + p];
    };
  };
  auto boxB = [=](int n) { return Box{1 - (nm - n), g.nx + (nm - n) + 1, 1 - (nm - n), g.ny + (nm - n) + 1, k0, k1}; };
  auto mkB = [=](int n) {
    return [=] FV3_HD(int t, int k, int i, int j) {
      if (!deln_on(d, k)) return;
      const int nord = deln_nord(d, k);
      if (n > nord) return;
      const int nt = nord - n;
      const int fl = g.flags[t];
      const long b = t * g.st + k * g.sk;
      const long m2 = t * g.st2;
      const Real *dd = d2 + b;
      if (i >= 1 - nt && i <= g.nx + nt + 1 && j >= 1 - nt && j <= g.ny + nt)
        (fx2 + b)[IX(i, j)] = (g.del6_v + m2)[IX(i, j)] * (cc<1>(dd, g, fl, i, j) - cc<1>(dd, g, fl, i - 1, j));
      if (i >= 1 - nt && i <= g.nx + nt && j >= 1 - nt && j <= g.ny + nt + 1)
        (fy2 + b)[IX(i, j)] = (g.del6_u + m2)[IX(i, j)] * (cc<2>(dd, g, fl, i, j) - cc<2>(dd, g, fl, i, j - 1));
    };
  };
  if constexpr (!std::is_same<Tail, NoTail>::value) {
    // window mode with a tail: the whole chain (orders up to 2) and the tail in one launch
    if (ws.n > 0 && nm <= 2) {
      launch_chain(c, s, ws, k0, k1, chain_stage(box0, st0), chain_stage(box1, st1), chain_stage(boxA(1), mkA(1)), chain_stage(boxB(1), mkB(1)),
                   chain_stage(boxA(2), mkA(2)), chain_stage(boxB(2), mkB(2)), tail);
      return;
    }
  }
  launch3w(c, s, box0, ws, st0);
  launch3w(c, s, box1, ws, st1);
  for (int n = 1; n <= nm; ++n) {
    launch3w(c, s, boxA(n), ws, mkA(n));
    launch3w(c, s, boxB(n), ws, mkB(n));
  }
  if constexpr (!std::is_same<Tail, NoTail>::value) launch3w(c, s, tail.nat, ws, tail.f);
}
static void del6_vt_flux_staged(fv3_ctx *c, fv3_stream_t s, const Real *q, Real *d2, Real *fx2, Real *fy2, const Deln &dn, bool q_raw, int k0, int k1,
                                const Box *win_, const Wins *wins_ = nullptr) {
  del6_vt_flux_staged_t(c, s, q, d2, fx2, fy2, dn, q_raw, k0, k1, win_, wins_, NoTail{});
}


// ---------------------------------------------------------------------------------------------
// Marching del-n chain (orders 0..2 = del-2, del-4, del-6; fv_tp_2d / d_sw never ask for more:
// get_column_namelist caps nord_v / nord_w / nord_t at 2).  A wave owns 58 face columns and
// walks j; iteration s of the chain runs s rows behind the row being loaded, its d2 / fx2 / fy2
// values live in registers, i-neighbours are exchanged through LDS lines.  q is read once,
// the final fluxes are written once, no intermediate reaches memory.
//   step r:  d2_0(r);  for s = 1..nord: d2_s(r-s) from the fluxes of iteration s-1;  fy_s(r-s) [own lane]
//            -> exchange d2_s(i-1) -> fx_s(r-s) -> exchange fx_s(i+1) -> kept for d2_(s+1) at the next step
// Faces whose dependency cone touches a cube corner (within nord+1 of it) come out wrong here
// (the corner-halo remap is not a tile-local operation); an 8 x 8 patch per corner is
// recomputed with the staged chain on a small window and overwrites them.
#define D6_OUT 58
#define D6_SEG 64
#define D6_PF 2
#define D6_NMAX 2
#define D6_PATCH FV3_D6_PATCH

static void del6_corner_patches(fv3_ctx *c, fv3_stream_t s, const Real *q, Real *d2, Real *fx2, Real *fy2, const Deln &dn, bool q_raw, int k0, int k1) {
  const Geo g = c->g;
  int any = 0;
  for (int t = 0; t < g.nsub; ++t) any |= g.flags[t];
  Real *tfx = c->scratch[SC_L], *tfy = c->scratch[SC_M];
  const Deln dd = dn;
  const int P = D6_PATCH, M = 4;  // staged results are exact >= 4 points inside an artificial window boundary
  struct Corner {
    int need;
    bool west, south;
  };
  const Corner corners[4] = {{FV3_W | FV3_S, true, true}, {FV3_E | FV3_S, false, true}, {FV3_E | FV3_N, false, false}, {FV3_W | FV3_N, true, false}};
  Wins wins, patches;
  int needs[4];
  wins.n = patches.n = 0;
  for (const Corner &cn : corners) {
    if ((any & cn.need) != cn.need) continue;
    Box w{cn.west ? -3 : g.nx + 1 - P - M, cn.west ? P + M : g.nx + 4, cn.south ? -3 : g.ny + 1 - P - M, cn.south ? P + M : g.ny + 4, k0, k1};
    Box pb{cn.west ? 1 : std::max(g.nx + 2 - P, 1), cn.west ? P : g.nx + 1, cn.south ? 1 : std::max(g.ny + 2 - P, 1), cn.south ? P : g.ny + 1, k0, k1};
    needs[wins.n] = cn.need;
    wins.w[wins.n++] = w;
    patches.w[patches.n++] = pb;
  }
  if (wins.n == 0) return;
  auto run = [&](const Wins &ws, const Wins &ps, const int *nd) {
    Wins psc = ps;
    int n0 = nd[0], n1 = ps.n > 1 ? nd[1] : 0, n2 = ps.n > 2 ? nd[2] : 0, n3 = ps.n > 3 ? nd[3] : 0;
    // (the copy of the patch into the flux fields is the last stage of the same launch; it tests patch membership itself)
    auto copy_patch = [=] FV3_HD(int t, int k, int i, int j) {
      if (!deln_on(dd, k) || deln_nord(dd, k) == 0) return;
      // which corner's patch is this point in (patches handled together are disjoint)
      int need = 0;
      const int nn[4] = {n0, n1, n2, n3};
      for (int w = 0; w < psc.n; ++w)
        if (i >= psc.w[w].i0 && i <= psc.w[w].i1 && j >= psc.w[w].j0 && j <= psc.w[w].j1) need = nn[w];
      if (need == 0 || (g.flags[t] & need) != need) return;
      const long p = t * g.st + k * g.sk + IX(i, j);
      if (j <= g.ny) fx2[p] = tfx[p];
      if (i <= g.nx) fy2[p] = tfy[p];
    };
    del6_vt_flux_staged_t(c, s, q, d2, tfx, tfy, dn, q_raw, k0, k1, nullptr, &ws, chain_stage(Box{1, g.nx + 1, 1, g.ny + 1, k0, k1}, copy_patch));
  };
  if (fv3_wins_disjoint(wins)) {
    run(wins, patches, needs);  // one set of launches for all corners
  } else {                      // tiny sub-domains: windows overlap, one corner at a time
    for (int w = 0; w < wins.n; ++w) {
      Wins one, pone;
      one.n = pone.n = 1;
      one.w[0] = wins.w[w];
      pone.w[0] = patches.w[w];
      run(one, pone, &needs[w]);
    }
  }
}

static void del6_stream(fv3_ctx *c, fv3_stream_t s, const Real *q, Real *d2, Real *fx2, Real *fy2, const Deln &dn, bool q_raw, int k0, int k1, bool edge_strips_only = false) {
  const Geo g = c->g;
  const Deln d = dn;
  const int nk = k1 - k0 + 1;
  const int npair = (nk + 1) / 2;  // a wave walks TWO levels at once: the three metric rows (half of the bytes) are loaded once
  const int nstrip = (g.nx + 1 + D6_OUT - 1) / D6_OUT;
  const int seg = fv3_pick_seg((long)nstrip * ((g.ny + 63) / 64) * g.nsub * npair, 3);
  const int nseg = (g.ny + seg - 1) / seg;
  const int LW = FV3_WAVE + 2;
  const size_t smem = sizeof(Real) * 2 * 2 * (D6_NMAX + 1) * LW;
  const int nx = g.nx, ny = g.ny, nh = g.nh, sj32 = g.sj32, go = g.o;
  const long st = g.st, sk = g.sk, st2 = g.st2;
  const MPtr del6_u = g.del6_u, del6_v = g.del6_v, rarea = g.rarea;
  const unsigned char *gflags = c->g_dev->flags;
  launch_waves<3>(c, s, nstrip, nseg, g.nsub * npair, smem, [=] FV3_HD(const Blk &blk, char *smem_) {
    const int t = blk.bz / npair, ka = k0 + 2 * (blk.bz - t * npair);
    // per-level controls (the two levels of a pair may differ in order / switch: sponge boundary)
    bool act[2];
    int nord[2];
    Real damp[2];
    long b[2];
    int nmax = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = ka + e;
      act[e] = k <= k1 && deln_on(d, k);
      nord[e] = act[e] ? deln_nord(d, k) : 0;
      damp[e] = act[e] ? deln_damp(d, k) : (Real)0;
      b[e] = t * st + (k <= k1 ? k : k1) * sk;
      if (act[e] && nord[e] > nmax) nmax = nord[e];
    }
    if (!act[0] && !act[1]) return;
    const long m2 = t * st2;
    const int i0 = 1 + blk.bx * D6_OUT;
    if (edge_strips_only) {  // only the strips on which tp2d_stream_t evaluates the W / E tile-edge formulas (the others run the chain themselves)
      const int fl = gflags[t];
      if (!(((fl & FV3_W) && i0 <= 3) || ((fl & FV3_E) && i0 + D6_OUT + 1 >= nx))) return;
    }
    const int ja = 1 + blk.by * seg;
    const int jb = blk.by == nseg - 1 ? ny + 1 : ja + seg - 1;
    const int ied = nx + nh, jsd = 1 - nh, jed = ny + nh;
    Real *ld = (Real *)smem_ + 1;                // ld[(e * (NMAX+1) + s) * LW + lane]: d2_s of the lane's cell (read by lane + 1)
    Real *lf = ld + 2 * (D6_NMAX + 1) * LW;      // lf[...]: fx_s of the lane's west face (read by lane - 1)
    const Real *qq0 = q + b[0], *qq1 = q + b[1];
    const MPtr dub = del6_u + m2, dvb = del6_v + m2, rab = rarea + m2;
    struct Row {
      Real q0, q1, du, dv, ra;
    };
    Row pf[D6_PF][FV3_LPT];
    Real du[D6_NMAX + 1][FV3_LPT], dv[D6_NMAX + 1][FV3_LPT], ra[D6_NMAX + 1][FV3_LPT];  // metrics of row r - s
    Real d2p[2][D6_NMAX + 1][FV3_LPT], fxp[2][D6_NMAX + 1][FV3_LPT], fxe[2][D6_NMAX + 1][FV3_LPT], fyp[2][D6_NMAX + 1][FV3_LPT];
    Real d2c[2][D6_NMAX + 1][FV3_LPT], fxc[2][D6_NMAX + 1][FV3_LPT], fyc[2][D6_NMAX + 1][FV3_LPT];
    unsigned pcol[FV3_LPT];
    bool own_x[FV3_LPT], own_y[FV3_LPT];
    int r_beg = ja - 1 - nmax, r_end = jb + nmax;
    if (r_beg < jsd) r_beg = jsd;
    if (r_end > jed) r_end = jed;
    auto load_row = [&](int r, int l) -> Row {
      const unsigned p0 = pcol[l] + (unsigned)(r * sj32);
      Row w;
      w.q0 = qq0[p0];
      w.q1 = qq1[p0];
      w.du = dub[p0];
      w.dv = dvb[p0];
      w.ra = rab[p0];
      return w;
    };
    FV3_LANES(blk, lane, l) {
      const int i = i0 - 3 + lane, ic = i < ied ? i : ied;
      pcol[l] = (unsigned)(go * sj32 + ic + go);
      // (edge strips only: one more x face -- the high face of the strip's last cell belongs to the next strip, which then does not run)
      own_x[l] = i >= i0 && i < i0 + D6_OUT + (edge_strips_only ? 1 : 0) && i <= nx + 1;
      own_y[l] = i >= i0 && i < i0 + D6_OUT && i <= nx;
#pragma unroll
      for (int s_ = 0; s_ <= D6_NMAX; ++s_) {
        du[s_][l] = dv[s_][l] = ra[s_][l] = (Real)0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          d2p[e][s_][l] = fxp[e][s_][l] = fxe[e][s_][l] = fyp[e][s_][l] = d2c[e][s_][l] = fxc[e][s_][l] = fyc[e][s_][l] = (Real)0;
          if (lane == 0) {
            ld[(e * (D6_NMAX + 1) + s_) * LW - 1] = (Real)0;
            lf[(e * (D6_NMAX + 1) + s_) * LW + FV3_WAVE] = (Real)0;
          }
        }
      }
#pragma unroll
      for (int n = 0; n < D6_PF; ++n) pf[n][l] = load_row(r_beg + n < r_end ? r_beg + n : r_end, l);
    }
    for (int r = r_beg; r <= r_end; ++r) {
      const int rn = r + D6_PF < r_end ? r + D6_PF : r_end;
      // ---- phase A: d2 of every iteration on its row, y-fluxes (own lane)
      FV3_LANES(blk, lane, l) {
        const Row cu = pf[0][l];
#pragma unroll
        for (int n = 0; n + 1 < D6_PF; ++n) pf[n][l] = pf[n + 1][l];
        pf[D6_PF - 1][l] = load_row(rn, l);
#pragma unroll
        for (int s_ = D6_NMAX; s_ >= 1; --s_) {
          du[s_][l] = du[s_ - 1][l];
          dv[s_][l] = dv[s_ - 1][l];
          ra[s_][l] = ra[s_ - 1][l];
        }
        du[0][l] = cu.du;
        dv[0][l] = cu.dv;
        ra[0][l] = cu.ra;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (!act[e]) continue;
          const Real qv = e == 0 ? cu.q0 : cu.q1;
          d2c[e][0][l] = q_raw ? qv : damp[e] * qv;
          fyc[e][0][l] = du[0][l] * (d2p[e][0][l] - d2c[e][0][l]);
          ld[e * (D6_NMAX + 1) * LW + lane] = d2c[e][0][l];
#pragma unroll
          for (int s_ = 1; s_ <= D6_NMAX; ++s_) {
            if (s_ <= nord[e]) {
              d2c[e][s_][l] = (fxp[e][s_ - 1][l] - fxe[e][s_ - 1][l] + fyp[e][s_ - 1][l] - fyc[e][s_ - 1][l]) * ra[s_][l];
              fyc[e][s_][l] = du[s_][l] * (d2c[e][s_][l] - d2p[e][s_][l]);
              ld[(e * (D6_NMAX + 1) + s_) * LW + lane] = d2c[e][s_][l];
            }
          }
        }
      }
      blk.wave_sync();
      // ---- phase B: x-fluxes from the west neighbour's d2
      FV3_LANES(blk, lane, l) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (!act[e]) continue;
          fxc[e][0][l] = dv[0][l] * (ld[e * (D6_NMAX + 1) * LW + lane - 1] - d2c[e][0][l]);
          lf[e * (D6_NMAX + 1) * LW + lane] = fxc[e][0][l];
#pragma unroll
          for (int s_ = 1; s_ <= D6_NMAX; ++s_) {
            if (s_ <= nord[e]) {
              fxc[e][s_][l] = dv[s_][l] * (d2c[e][s_][l] - ld[(e * (D6_NMAX + 1) + s_) * LW + lane - 1]);
              lf[(e * (D6_NMAX + 1) + s_) * LW + lane] = fxc[e][s_][l];
            }
          }
        }
      }
      blk.wave_sync();
      // ---- phase C: east neighbour's x-flux for the next step; final fluxes of the last iteration
      FV3_LANES(blk, lane, l) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (!act[e]) continue;
#pragma unroll
          for (int s_ = 0; s_ <= D6_NMAX; ++s_) {
            if (s_ <= nord[e]) {
              fxe[e][s_][l] = lf[(e * (D6_NMAX + 1) + s_) * LW + lane + 1];
              fxp[e][s_][l] = fxc[e][s_][l];
              fyp[e][s_][l] = fyc[e][s_][l];
              d2p[e][s_][l] = d2c[e][s_][l];
            }
          }
          const int jo = r - nord[e];
          const bool fx_row = jo >= ja && jo <= jb && jo <= ny, fy_row = jo >= ja && jo <= jb;
          const unsigned p = pcol[l] + (unsigned)(jo * sj32);
          Real ox = fxc[e][0][l], oy = fyc[e][0][l];
#pragma unroll
          for (int s_ = 1; s_ <= D6_NMAX; ++s_) {
            if (s_ == nord[e]) {
              ox = fxc[e][s_][l];
              oy = fyc[e][s_][l];
            }
          }
          if (fx_row && own_x[l]) (fx2 + b[e])[p] = ox;
          if (fy_row && own_y[l]) (fy2 + b[e])[p] = oy;
        }
      }
      blk.wave_sync();
    }
  });
  if (dn.nord_max > 0) del6_corner_patches(c, s, q, d2, fx2, fy2, dn, q_raw, k0, k1);
}

void del6_vt_flux_edge_strips(fv3_ctx *c, fv3_stream_t s, const Real *q, Real *d2, Real *fx2, Real *fy2, const Deln &dn, bool q_raw, int k0, int k1) {
  if (k0 > k1) return;
  if (dn.nord_max > D6_NMAX) {
    del6_vt_flux_staged(c, s, q, d2, fx2, fy2, dn, q_raw, k0, k1, nullptr);
    return;
  }
  del6_stream(c, s, q, d2, fx2, fy2, dn, q_raw, k0, k1, true);
}

void del6_vt_flux_patches(fv3_ctx *c, fv3_stream_t s, const Real *q, Real *d2, Real *fx2, Real *fy2, const Deln &dn, bool q_raw, int k0, int k1) {
  if (dn.nord_max > 0 && k0 <= k1) del6_corner_patches(c, s, q, d2, fx2, fy2, dn, q_raw, k0, k1);
}

void del6_vt_flux(fv3_ctx *c, fv3_stream_t s, const Real *q, Real *d2, Real *fx2, Real *fy2, const Deln &dn, bool q_raw, int k0, int k1) {
  if (k0 > k1) return;
  // FV3_DEL6_MODE = staged | stream (default): the staged form is the reference / A-B path
  static const char *mode_env = getenv("FV3_DEL6_MODE");
  static const bool staged = mode_env && !strcmp(mode_env, "staged");
  if (staged || dn.nord_max > D6_NMAX)
    del6_vt_flux_staged(c, s, q, d2, fx2, fy2, dn, q_raw, k0, k1, nullptr);
  else
    del6_stream(c, s, q, d2, fx2, fy2, dn, q_raw, k0, k1);
}

static void tp2d_staged(fv3_ctx *c, fv3_stream_t s, const Real *q, const Real *crx, const Real *cry, const Real *xfx, const Real *yfx, Real *fx, Real *fy,
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
      auto M = [&](int s_) { return (g.dya + m2)[IX(i, s_)]; };
      (fy2 + b)[IX(i, j)] = ppm_flux(Q, M, (cry + b)[IX(i, j)], j, (fl & FV3_S) != 0, (fl & FV3_N) != 0, g.npy, hord);
    }
    if (i >= 1 && i <= g.nx + 1) {  // fx2 on j = jsd..jed
      auto Q = [&](int s_) { return cc<1>(qq, g, fl, s_, j); };
      auto M = [&](int s_) { return (g.dxa + m2)[IX(s_, j)]; };
      (fx2 + b)[IX(i, j)] = ppm_flux(Q, M, (crx + b)[IX(i, j)], i, (fl & FV3_W) != 0, (fl & FV3_E) != 0, g.npx, hord);
    }
  });
  // (2) cross-direction updates
  launch3(c, s, Box{isd, ied, jsd, jed, k0, k1}, [=] FV3_HD(int t, int k, int i, int j) {
    const long b = t * g.st + k * g.sk;
    const long m2 = t * g.st2;
    const unsigned p = IX(i, j);
    const Real qa = (q + b)[p] * (g.area + m2)[p];
    if (j >= 1 && j <= g.ny) {
      const unsigned pn = IX(i, j + 1);
      const Real y0 = (yfx + b)[p], y1 = (yfx + b)[pn];
      const Real ra_y = (g.area + m2)[p] + y0 - y1;
      (q_i + b)[p] = (qa + y0 * (fy2 + b)[p] - y1 * (fy2 + b)[pn]) / ra_y;
    }
    if (i >= 1 && i <= g.nx) {
      const unsigned pe = IX(i + 1, j);
      const Real x0 = (xfx + b)[p], x1 = (xfx + b)[pe];
      const Real ra_x = (g.area + m2)[p] + x0 - x1;
      (q_j + b)[p] = (qa + x0 * (fx2 + b)[p] - x1 * (fx2 + b)[pe]) / ra_x;
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
    const unsigned p = IX(i, j);
    const bool on = damped && deln_on(d, k);
    if (j <= g.ny) {
      const Real *qq = q_i + b;
      auto Q = [&](int s_) { return qq[IX(s_, j)]; };
      auto M = [&](int s_) { return (g.dxa + m2)[IX(s_, j)]; };
      const Real f = ppm_flux(Q, M, (crx + b)[p], i, (fl & FV3_W) != 0, (fl & FV3_E) != 0, g.npx, hord);
      Real v = (Real)0.5 * (f + (fx2 + b)[p]) * (mfx ? (mfx + b)[p] : (xfx + b)[p]);
      if (on) {
        if (mass)
          v = v + (Real)0.5 * deln_damp(d, k) * ((mass + b)[IX(i - 1, j)] + (mass + b)[p]) * (dfx + b)[p];
        else
          v = v + (dfx + b)[p];
      }
      (fx + b)[p] = v;
    }
    if (i <= g.nx) {
      const Real *qq = q_j + b;
      auto Q = [&](int s_) { return qq[IX(i, s_)]; };
      auto M = [&](int s_) { return (g.dya + m2)[IX(i, s_)]; };
      const Real f = ppm_flux(Q, M, (cry + b)[p], j, (fl & FV3_S) != 0, (fl & FV3_N) != 0, g.npy, hord);
      Real v = (Real)0.5 * (f + (fy2 + b)[p]) * (mfy ? (mfy + b)[p] : (yfx + b)[p]);
      if (on) {
        if (mass)
          v = v + (Real)0.5 * deln_damp(d, k) * ((mass + b)[IX(i, j - 1)] + (mass + b)[p]) * (dfy + b)[p];
        else
          v = v + (dfy + b)[p];
      }
      (fy + b)[p] = v;
    }
  });
}


// ---------------------------------------------------------------------------------------------
// Marching form.  One wavefront = a strip of 64 columns (58 owned + 3 halo columns a side) of
// one level; the lanes march together along j over a segment of rows.  Per row a lane loads its
// q / crx / xfx / cry / yfx values once, the two y-sweeps (inner flux of q, outer flux of the
// x-advected q_j) run entirely from 4-row register windows and share edge values / cell
// reconstructions between consecutive faces, the two x-sweeps (inner flux of q on the new row,
// outer flux of the y-advected q_i three rows behind) read their i-neighbours from two LDS row
// lines.  Exactly four PPM face evaluations per cell and no intermediate field in memory.
//   step r:  fy_in(r-2) -> q_i(r-3);   fx_in(r), fx_out(r-3) -> fx(r-3);   q_j(r) -> fy_out(r-2) -> fy(r-2)
#define TS_OUT 58
#define TS_SEG 64
#define TS_LINE (FV3_WAVE + 6)
#ifndef TS_WPE
// waves per SIMD the register allocation is sized for.  fp64: two (250 registers); the fp32 build needs 170 and runs three -- the marches are
// bound by the length of a wave's row step, not by a throughput resource, so a third wave is worth 8 - 12 % there
#define TS_WPE (sizeof(Real) == 4 ? 3 : 2)
#endif
#ifndef TS_LDS_ONLY
#define TS_LDS_ONLY 0
#endif
#ifndef TS_PF
#define TS_PF 2  // rows fetched ahead of their use (1 or 2)
#endif

// FEAT: compile-time feature mask (TF_*): every call site gets a kernel that carries only the state,
// pointers and branches it uses (the all-features kernel sat at 256 VGPRs with scratch and 131 spilled SGPRs).
enum { TF_MFX = 1, TF_DAMP = 2, TF_MASS = 4, TF_EPI = 8, TF_AREA = 16, TF_WIND = 32, TF_WFLUX = 64, TF_ACC = 128, TF_ALL = 255, TF_FD = 256 };

// HC: 0 = run-time PPM order; 6 = the order is that constant (see dsw_scalars_t in fv3_tp4.hip).
// FA ("all on", FD instantiations only): the caller's contract for epi->fd != 0 -- every level k0..k1 has its del-n chain switched on,
// and the wind form comes with wind_u_pre / wind_v_pre -- is taken as a compile-time fact on the strips away from the W / E tile
// edges: the chain / damping / epilogue tests fold away, every store of a step is issued (unowned lanes / rows write to the sink:
// fv3_store_sel) and ke(i+1, jf) / ke(i, jr) come from the neighbouring lane / the previous step instead of two more loads.
template <unsigned FEAT, int HC = 0, bool FA = false>
static void tp2d_stream_t(fv3_ctx *c, fv3_stream_t s, const Real *q, const Real *crx, const Real *cry, const Real *xfx, const Real *yfx, Real *fx, Real *fy,
                        const Real *mfx, const Real *mfy, const Real *mass, int hord_, const Deln *dn, int k0, int k1, const TpEpi *epi) {
  const int hord = HC ? HC : hord_;
  static_assert(!FA || (FEAT & TF_FD), "FA is a property of the FD instantiations");
  Real *const trash = c->trash;
  const Geo g = c->g;
  Real *dfx = c->scratch[SC_DN_FX], *dfy = c->scratch[SC_DN_FY];
  const bool damped = dn != nullptr;
  Deln d;
  memset(&d, 0, sizeof(d));
  if (damped) {
    d = *dn;
    if (epi && epi->damp_fx) {  // computed by the caller
      dfx = const_cast<Real *>(epi->damp_fx);
      dfy = const_cast<Real *>(epi->damp_fy);
    } else {
      del6_vt_flux(c, s, q, c->scratch[SC_DN_D2], dfx, dfy, d, mass != nullptr, k0, k1);
    }
  }
  const int nk = k1 - k0 + 1;
  const int nstrip = (g.nx + 1 + TS_OUT - 1) / TS_OUT;
  const int seg = fv3_pick_seg((long)nstrip * ((g.ny + 63) / 64) * g.nsub * nk, 2);
  const int nseg = (g.ny + seg - 1) / seg;
  constexpr int TS_NRING = (FEAT & TF_FD) ? 7 : 0;  // FD: the chain's metric rows + the parked delay lines, own-lane ring [variable][row & 3][lane]
  const size_t smem_lines = (size_t)(2 * TS_LINE + 5 * (FV3_WAVE + 1) + 32);
  const size_t smem = sizeof(Real) * (smem_lines + (size_t)TS_NRING * 4 * FV3_WAVE);
  const bool area_form = epi && epi->area_form;
  const bool fd_on = (FEAT & TF_FD) && epi && epi->fd && !TS_LDS_ONLY;
  const Real *fd_coef = epi ? epi->fd_coef : nullptr;
  const MPtr fd_add = (FEAT & TF_FD) && epi ? (MPtr)epi->fd_add : nullptr;  // 2-D term added to q where it is loaded (absolute = relative vorticity + f0)
  const MPtr d6u = g.del6_u, d6v = g.del6_v;
  const Real *zfx = epi ? epi->zfx : nullptr, *zfy = epi ? epi->zfy : nullptr, *zon = epi ? epi->zon : nullptr;
  Real *epi_out = epi ? epi->out : nullptr;
  const Real *epi_mult = epi ? epi->mult : nullptr;
  const bool wflux = epi ? epi->write_flux : true;
  Real *acc_x = epi ? epi->acc_x : nullptr, *acc_y = epi ? epi->acc_y : nullptr;
  Real *wind_u = epi ? epi->wind_u : nullptr, *wind_v = epi ? epi->wind_v : nullptr;
  const Real *wind_ke = epi ? epi->wind_ke : nullptr;
  const Real *wind_du = epi ? epi->wind_du : nullptr, *wind_dv = epi ? epi->wind_dv : nullptr, *wind_don = epi ? epi->wind_don : nullptr;
  Real *wind_u_pre = epi ? epi->wind_u_pre : nullptr, *wind_v_pre = epi ? epi->wind_v_pre : nullptr;
  const MPtr rarea = g.rarea, gdx = g.dx, gdy = g.dy;
  // The hot loop touches only these scalars; everything the rare paths need (cube-corner remaps,
  // tile-edge metric terms) is read through gp inside those paths, so it does not occupy SGPRs
  // (or spill lanes) across the march.
  const Geo *gp = c->g_dev;
  const int nx = g.nx, ny = g.ny, nh = g.nh, npx = g.npx, npy = g.npy, sj32 = g.sj32, go = g.o;
  const long st = g.st, sk = g.sk, st2 = g.st2;
  const MPtr area = g.area, gdxa = g.dxa;
  // Launch geometry as in fv3_tp4.hip: level-major (KB levels of one (strip, segment) tile are consecutive workgroups of an XCD, so the
  // tile's 2-D metric rows are fetched into that XCD's L2 once per KB levels instead of once per level) when KB > 0; FV3_Q4_KB=0
  // selects the plane-major form (A/B; same values, same time, 15 - 20 % more L2 misses).
  static const int kb_env = getenv("FV3_Q4_KB") ? atoi(getenv("FV3_Q4_KB")) : FV3_Q4_KB_DEFAULT;
  const int KB = kb_env > 0 ? (kb_env < nk ? kb_env : nk) : 0;
  const int nblk = KB ? (nk + KB - 1) / KB : 0;
#if defined(FV3_STAMPS) && !defined(FV3_HOST_EMU)
  unsigned long long *const st_buf = fv3_stamp_buf();
  constexpr unsigned long long st_kid = 2000ull + FEAT + 10000ull * HC;
#endif
  launch_waves<TS_WPE>(c, s, KB ? KB : nstrip, KB ? nstrip * nseg : nseg, KB ? g.nsub * nblk : g.nsub * nk, smem, [=] FV3_HD(const Blk &blk_, char *smem_) {
    Blk blk = blk_;
    int t_, k_;
    if (KB) {
      t_ = blk_.bz / nblk;
      const int kk = (blk_.bz - t_ * nblk) * KB + blk_.bx;
      if (kk >= nk) return;
      k_ = k0 + kk;
      blk.by = blk_.by / nstrip;
      blk.bx = blk_.by - blk.by * nstrip;
    } else {
      t_ = blk.bz / nk;
      k_ = k0 + (blk.bz - t_ * nk);
    }
    // fold the features this instantiation does not have
    constexpr bool C_MFX = FEAT & TF_MFX, C_DAMP = FEAT & TF_DAMP, C_MASS = FEAT & TF_MASS, C_EPI = FEAT & TF_EPI, C_AREA = FEAT & TF_AREA;
    constexpr bool C_WIND = FEAT & TF_WIND, C_WFLUX = FEAT & TF_WFLUX, C_ACC = FEAT & TF_ACC;
    const Real *const mfx_ = C_MFX ? mfx : nullptr, *const mfy_ = C_MFX ? mfy : nullptr, *const mass_ = C_MASS ? mass : nullptr;
    Real *const epi_out_ = C_EPI ? epi_out : nullptr;
    const Real *const epi_mult_ = C_EPI ? epi_mult : nullptr;
    Real *const acc_x_ = C_ACC ? acc_x : nullptr, *const acc_y_ = C_ACC ? acc_y : nullptr;
    Real *const wind_u_ = C_WIND ? wind_u : nullptr, *const wind_v_ = C_WIND ? wind_v : nullptr;
    const Real *const wind_ke_ = C_WIND ? wind_ke : nullptr;
    const Real *const wind_du_ = C_WIND ? wind_du : nullptr, *const wind_dv_ = C_WIND ? wind_dv : nullptr;
    Real *const wind_u_pre_ = C_WIND ? wind_u_pre : nullptr, *const wind_v_pre_ = C_WIND ? wind_v_pre : nullptr;
    const bool wflux_ = C_WFLUX && wflux, area_form_ = C_AREA && area_form;
    const int t = t_, k = k_;
    const int fl = gp->flags[t];
    const long b = t * st + k * sk, m2 = t * st2;
    const int i0 = 1 + blk.bx * TS_OUT;                            // first owned face / cell
    const int ja = 1 + blk.by * seg;                            // first owned row / face
    const int jb = blk.by == nseg - 1 ? ny + 1 : ja + seg - 1;  // last owned face (rows stop at ny)
    const int ied = nx + nh, jsd = 1 - nh, jed = ny + nh;
    Real *lq = (Real *)smem_;         // q on the row being loaded (x-sweep view); index = i - i0 + 6
    Real *lqi = lq + TS_LINE;         // q_i three rows behind
    Real *exp_ = lqi + TS_LINE;       // xfx * fx_in of the lane (read by lane - 1)
    Real *exx = exp_ + FV3_WAVE + 1;  // xfx
    Real *exf = exx + FV3_WAVE + 1;   // final fx of the lane's face (epilogue: read by lane - 1)
    Real *exj = exf + FV3_WAVE + 1;   // xfx(i, r-3) (area-form epilogue)
    Real *exm = exj + FV3_WAVE + 1;   // mass(i, r-3) of the lane (read by lane + 1: the west cell of its face)
    Real *emr = exm + FV3_WAVE + 1;   // tile-edge strips: dxa of the 4 + 4 cells around the W / E edge, rows r..r-3 (ring of 4 x 8)
    Real *ring = (Real *)smem_ + smem_lines;
    enum { RG_DU = 0, RG_DV = 1, RG_RA = 2, RG_CX = 3, RG_XV = 4, RG_AR = 5, RG_FI = 6 };
    auto RG = [&](int var, int r) -> Real * { return ring + (var * 4 + (r & 3)) * FV3_WAVE; };
    const Real *qq = q + b, *crxb = crx + b, *cryb = cry + b, *xfxb = xfx + b, *yfxb = yfx + b;
    const MPtr areab = area + m2;
    const bool W = (fl & FV3_W) && i0 <= 3, E = (fl & FV3_E) && i0 + TS_OUT + 1 >= npx - 1;
    const bool S = fl & FV3_S, N = fl & FV3_N;
    const bool halo_cols = i0 - 3 < 1 || i0 + FV3_WAVE - 4 > nx;
    const bool on = C_DAMP && damped && deln_on(d, k);
    const Real damp = on ? deln_damp(d, k) : (Real)0;
    const int r_end = jb + 3 < jed ? jb + 3 : jed;
    const bool need_mc = mass_ && (on || (epi_out_ && epi_mult_ == mass_));  // mass_(i, r-2): damping weight and / or epilogue multiplier

    // per-lane marching state
    struct Row {  // the inputs of one step, fetched TS_PF steps ahead of their use
      Real qy, cx, xv, ar, cy, yv;
      Real em;  // tile-edge strips, lanes 0..7: dxa(edge cell, r) for the one-sided formulas (prefetched with the row: a load inside the step stalls it)
    };
    // optional inputs consumed at the end of a step (mass_ fluxes, damping fluxes, mass_(i-1, r-3), mass_(i, r-2),
    // epilogue terms): loaded at the top of the same step, AHEAD of the prefetch, so that waiting for them
    // (loads return in order) leaves the prefetched rows in flight
    Real o_mx[FV3_LPT], o_my[FV3_LPT], o_dx[FV3_LPT], o_dy[FV3_LPT], o_mc[FV3_LPT], o_ax[FV3_LPT], o_ay[FV3_LPT];
    Real mb[FV3_LPT];  // mass_(i, r-3) = mass_(i, r-2) of the previous step
    Real fxk[FV3_LPT], fyp[FV3_LPT], era[FV3_LPT], emu[FV3_LPT];  // epilogue: fx(r-3), fy(face r-3), rarea / mult at row r-3
    Real xjr[FV3_LPT], ypp[FV3_LPT];  // xfx(i, r-3), yfx(i, r-3) (area-form epilogue)
    Real zx0[FV3_LPT], zx1[FV3_LPT], zy0[FV3_LPT], zy1[FV3_LPT];  // damping fluxes around the cell (i, r-3)
    const bool zdamp = C_AREA && (FA || (zfx && zon[k] > (Real)1.0e-5));
    const bool wdamp = C_WIND && (FA || (wind_du != nullptr && wind_don[k] > (Real)1.0e-5));
    // FD: the del-n chain of q inside the march (strips away from the W / E tile edges); see dsw_scalars_t in fv3_tp4.hip
    constexpr bool C_FD = (FEAT & TF_FD) != 0;
    const bool fdm = C_FD && (FA || fd_on) && (C_AREA ? zdamp : wdamp) && !(W || E);
    Real *const sink0 = trash + (size_t)(((unsigned)blk_.bx + 61u * (unsigned)blk_.by + 127u * (unsigned)blk_.bz) & (FV3_TRASH_SLOTS - 1)) * FV3_WAVE;
    const Real dcoef = fdm ? fd_coef[k] : (Real)0;
    Real sd0[FV3_LPT], sd1[FV3_LPT], sd2[FV3_LPT], gx0[FV3_LPT], gx1[FV3_LPT], gy0[FV3_LPT], gy1[FV3_LPT], dxd[FV3_LPT], dxn[FV3_LPT], dyf[FV3_LPT], zyp[FV3_LPT], zxo[FV3_LPT];
    Real mdu_n[FV3_LPT], mdv_n[FV3_LPT], mra_n[FV3_LPT];
    constexpr bool C_ADD = C_FD && C_WIND;  // q + fd_add formed where q is consumed (fd_add fetched one step ahead, like the chain's metric rows)
    Real qa_n[FV3_LPT];
    const MPtr addb = C_ADD && fd_add ? fd_add + m2 : nullptr;
    const MPtr d6ub = d6u + m2, d6vb = d6v + m2, rab = rarea + m2;
    const bool c_sw_ = (fl & (FV3_W | FV3_S)) == (FV3_W | FV3_S), c_se = (fl & (FV3_E | FV3_S)) == (FV3_E | FV3_S);
    const bool c_ne = (fl & (FV3_E | FV3_N)) == (FV3_E | FV3_N), c_nw = (fl & (FV3_W | FV3_N)) == (FV3_W | FV3_N);
    const bool pz = fdm && (((c_sw_ || c_nw) && i0 - 3 <= FV3_D6_PATCH) || ((c_se || c_ne) && i0 + FV3_WAVE - 4 >= nx + 2 - FV3_D6_PATCH));
    auto on_patch = [&](int i, int j) -> bool {
      const bool wi = i >= 1 && i <= FV3_D6_PATCH, ei = i >= nx + 2 - FV3_D6_PATCH && i >= 1 && i <= nx + 1;
      const bool sj = j >= 1 && j <= FV3_D6_PATCH, nj = j >= ny + 2 - FV3_D6_PATCH && j >= 1 && j <= ny + 1;
      return (wi && sj && c_sw_) || (ei && sj && c_se) || (ei && nj && c_ne) || (wi && nj && c_nw);
    };
    Real wu[FV3_LPT], wdx[FV3_LPT], wkf[FV3_LPT], wke[FV3_LPT], wv[FV3_LPT], wdy[FV3_LPT], wkr[FV3_LPT];  // wind epilogue inputs
    Real sqx[FV3_LPT], sqi[FV3_LPT], sxv[FV3_LPT], smb[FV3_LPT];  // strips away from the W / E tile edges: what the neighbouring lanes read (wavefront shuffles)
    Real wdu[FV3_LPT], wdv[FV3_LPT];  // vorticity-damping increments of u (face r-2) / v (row r-3)
    Row nxt[FV3_LPT], nx2[FV3_LPT], cur[FV3_LPT];
    Real a1[FV3_LPT], a2[FV3_LPT], a3[FV3_LPT];  // area of rows r-1, r-2, r-3 (a delay line instead of a second load of the metric)
    Real w2[FV3_LPT], w3[FV3_LPT], w4[FV3_LPT], w5[FV3_LPT], al_q[FV3_LPT];  // q rows r-3..r, al(r-2)
    Real v2[FV3_LPT], v3[FV3_LPT], v4[FV3_LPT], v5[FV3_LPT], al_v[FV3_LPT];  // q_j likewise
    PpmCell cq[FV3_LPT], cv[FV3_LPT];                                        // reconstructed cell r-3 of q / q_j
    Real p_prev[FV3_LPT], y_prev[FV3_LPT];                                   // yfx * fy_in and yfx of face r-3
    Real fi1[FV3_LPT], fi2[FV3_LPT], fi3[FV3_LPT];                           // fx_in of rows r-1, r-2, r-3
    Real cx1[FV3_LPT], cx2[FV3_LPT], cx3[FV3_LPT], xv1[FV3_LPT], xv2[FV3_LPT], xv3[FV3_LPT];
    Real fyin[FV3_LPT], px[FV3_LPT];
    unsigned pcol[FV3_LPT];  // in-plane offset of (ic, 0)
    bool own_x[FV3_LPT], own_y[FV3_LPT];

    const MPtr dxab0 = gdxa + m2;
    auto load_row = [&](int r, int l, int lane) -> Row {
      const int rf = r - 2 < jsd ? jsd : r - 2;  // face / row of the y-sweeps (clamped while the windows fill)
      const unsigned p0 = pcol[l] + (unsigned)(r * sj32), pf = pcol[l] + (unsigned)(rf * sj32);
      Row w;
      w.qy = qq[p0];
      w.cx = crxb[p0];
      w.xv = xfxb[p0];
      w.ar = areab[p0];
      w.cy = cryb[pf];
      w.yv = yfxb[pf];
      w.em = (Real)1;
      if ((W || E) && lane < 8) {
        const bool have = lane < 4 ? W : E;
        const int sc = lane < 4 ? lane - 1 : npx - 2 + (lane - 4);  // columns -1..2 (W) / npx-2..npx+1 (E)
        if (have) w.em = dxab0[(unsigned)((r + go) * sj32 + sc + go)];
      }
      return w;
    };
    FV3_LANES(blk, lane, l) {
      const int i = i0 - 3 + lane, ic = i < ied ? i : ied;
      pcol[l] = (unsigned)(go * sj32 + ic + go);
      own_x[l] = i >= i0 && i < i0 + TS_OUT && i <= nx + 1;
      own_y[l] = i >= i0 && i < i0 + TS_OUT && i <= nx;
      w2[l] = w3[l] = w4[l] = w5[l] = al_q[l] = v2[l] = v3[l] = v4[l] = v5[l] = al_v[l] = (Real)0;
      a1[l] = a2[l] = a3[l] = (Real)1;  // (warm-up steps: outputs masked, keep the divisions finite)
      cq[l] = cv[l] = PpmCell{(Real)0, (Real)0, (Real)0, false};
      wu[l] = wdx[l] = wkf[l] = wke[l] = wv[l] = wdy[l] = wkr[l] = wdu[l] = wdv[l] = (Real)0;
      sqx[l] = sqi[l] = sxv[l] = smb[l] = (Real)0;
      fxk[l] = fyp[l] = era[l] = emu[l] = o_mx[l] = o_my[l] = o_dx[l] = o_dy[l] = o_mc[l] = o_ax[l] = o_ay[l] = (Real)0;
      if (lane == 0) exf[FV3_WAVE] = exj[FV3_WAVE] = (Real)0;
      xjr[l] = ypp[l] = zx0[l] = zx1[l] = zy0[l] = zy1[l] = (Real)0;
      sd0[l] = sd1[l] = sd2[l] = gx0[l] = gx1[l] = gy0[l] = gy1[l] = dxd[l] = dxn[l] = dyf[l] = zyp[l] = zxo[l] = mdu_n[l] = mdv_n[l] = mra_n[l] = (Real)0;
      qa_n[l] = (Real)0;
      if constexpr (C_ADD) {
        if (addb) qa_n[l] = addb[pcol[l] + (unsigned)((ja - 3) * sj32)];
      }
      if constexpr (C_FD) {
        if (!(W || E))
          for (int v = 0; v < TS_NRING; ++v)
            for (int q_ = 0; q_ < 4; ++q_) RG(v, q_)[lane] = v == RG_AR ? (Real)1 : (Real)0;
        if (fdm) {
          const unsigned pm = pcol[l] + (unsigned)((ja - 3) * sj32);
          mdu_n[l] = d6ub[pm];
          mdv_n[l] = d6vb[pm];
          mra_n[l] = rab[pm];
        }
      }
      mb[l] = p_prev[l] = y_prev[l] = fi1[l] = fi2[l] = fi3[l] = cx1[l] = cx2[l] = cx3[l] = xv1[l] = xv2[l] = xv3[l] = (Real)0;
      if (lane < 3) lq[lane] = lqi[lane] = lq[FV3_WAVE + 3 + lane] = lqi[FV3_WAVE + 3 + lane] = (Real)0;
      if (lane == 0) exp_[FV3_WAVE] = exx[FV3_WAVE] = (Real)0;
      nxt[l] = load_row(ja - 3, l, lane);
      if (TS_PF == 2) nx2[l] = load_row(ja - 2 < r_end ? ja - 2 : r_end, l, lane);
      if (lane < 32) emr[lane] = (Real)1;
      exm[lane] = (Real)0;
    }

    // XE: this strip reaches a cube-tile edge in x (one-sided PPM formulas among its faces)
    FV3_STAMP_STATE;
    auto march = [&](auto xe_tag) {
      constexpr bool XE = decltype(xe_tag)::value;
      constexpr bool LX = XE || TS_LDS_ONLY;  // neighbour reads through the LDS lines (tile-edge strips; TS_LDS_ONLY: everywhere, the A/B form)
      constexpr bool UST = FA && !XE;         // unconditional stores, derived ke values (strips away from the W / E tile edges)
      auto step = [&](int r) {
        FV3_STAMP(0);
        const int r3 = r - 3 < jsd ? jsd : r - 3;
        const int rn = r + TS_PF < r_end ? r + TS_PF : r_end;
        const int sy = r - 1;  // cell whose low edge value the y-windows complete at this step
        const bool y_edge = (S && sy >= 0 && sy <= 2) || (N && sy >= npy - 1 && sy <= npy + 1);
        const bool corner_row = halo_cols && (r < 1 || r > ny);
        // ---- phase 1: prefetch row r+2; inner y-flux at face r-2, q_i at row r-3
        FV3_LANES(blk, lane, l) {
          {
            const int rf = r - 2 < jsd ? jsd : r - 2;
            const unsigned p3 = pcol[l] + (unsigned)(r3 * sj32), pf = pcol[l] + (unsigned)(rf * sj32);
            if (mfx_) {
              o_mx[l] = (mfx_ + b)[p3];
              o_my[l] = (mfy_ + b)[pf];
            }
            if (acc_x_) {  // accumulated fluxes: fetched here, not at the += (a load consumed at once stalls the step for a full memory latency)
              o_ax[l] = (acc_x_ + b)[p3];
              o_ay[l] = (acc_y_ + b)[pf];
            }
            if (on) {
              o_dx[l] = (dfx + b)[p3];
              o_dy[l] = (dfy + b)[pf];
            }
            if (need_mc) o_mc[l] = (mass_ + b)[pf];
            if (epi_out_) {
              if (!(C_FD && fdm)) era[l] = (rarea + m2)[p3];
              if (epi_mult_ && epi_mult_ != mass_) emu[l] = (epi_mult_ + b)[p3];
              if (zdamp && !(C_FD && fdm)) {
                zx0[l] = (zfx + b)[p3];
                zx1[l] = (zfx + b)[p3 + 1];
                zy0[l] = (zfy + b)[p3];
                zy1[l] = (zfy + b)[p3 + (unsigned)sj32];
              }
            }
            if constexpr (C_FD && !XE) {
              if (fdm) {
                // the chain's metric terms: row r (fetched during the previous step) goes into the ring, row r+1 is requested
                const Real du_c = mdu_n[l], dv_c = mdv_n[l], ra_c = mra_n[l];
                const int r1 = r + 1 < r_end ? r + 1 : r_end;
                const unsigned pm = pcol[l] + (unsigned)(r1 * sj32);
                mdu_n[l] = d6ub[pm];
                mdv_n[l] = d6vb[pm];
                mra_n[l] = rab[pm];
                RG(RG_DU, r)[lane] = du_c;
                RG(RG_DV, r)[lane] = dv_c;
                RG(RG_RA, r)[lane] = ra_c;
                era[l] = RG(RG_RA, r - 3)[lane];
              }
            }
            if (wind_u_) {
              wu[l] = (wind_u_ + b)[pf];
              wdx[l] = (gdx + m2)[pf];
              if constexpr (UST) wkr[l] = wkf[l];  // ke(i, jr) = the ke(i, jf) of the previous step (jr = jf - 1)
              wkf[l] = (wind_ke_ + b)[pf];      // ke(i, jf) -- also ke(i, jr + 1)
              if constexpr (!UST) wke[l] = (wind_ke_ + b)[pf + 1];  // ke(i + 1, jf)  (UST: the neighbouring lane's wkf, below)
              wv[l] = (wind_v_ + b)[p3];
              wdy[l] = (gdy + m2)[p3];
              if constexpr (!UST) wkr[l] = (wind_ke_ + b)[p3];      // ke(i, jr)
              if (wdamp && !(C_FD && fdm)) {
                wdu[l] = (wind_du_ + b)[pf];
                wdv[l] = (wind_dv_ + b)[p3];
              }
            }
          }
          cur[l] = nxt[l];
          if (TS_PF == 2) {
            nxt[l] = nx2[l];
            nx2[l] = load_row(rn, l, lane);
          } else {
            nxt[l] = load_row(rn, l, lane);
          }
          FV3_STAMP(1);
          FV3_STAMP_USE(cur[l].yv);
          FV3_STAMP_USE(cur[l].cy);
          FV3_STAMP_USE(cur[l].cx);
          FV3_STAMP_USE(cur[l].xv);
          FV3_STAMP_USE(cur[l].ar);
          FV3_STAMP_USE(cur[l].qy);
          FV3_STAMP(2);
          if (XE && lane < 8) emr[(r & 3) * 8 + lane] = cur[l].em;
          const Real qraw = cur[l].qy;  // (the del-n chain runs on the field itself: no added term, no corner remap)
          Real qy = qraw, qx = qy;
          if constexpr (C_ADD) {
            if (addb) {
              qy = qx = qraw + qa_n[l];  // (row r's term, requested during the previous step)
              cur[l].qy = qy;
              const int r1 = r + 1 < r_end ? r + 1 : r_end;
              qa_n[l] = addb[pcol[l] + (unsigned)(r1 * sj32)];
            }
          }
          if (corner_row) {  // the two sweeps see the cube-corner cells through different remaps (rare, not prefetched)
            const int i = i0 - 3 + lane, ic = i < ied ? i : ied;
            const int rc = r < jed ? r : jed;  // (trailing steps of the unrolled march)
            const unsigned iy = cc_index<2>(*gp, fl, ic, rc), ix = cc_index<1>(*gp, fl, ic, rc);
            qy = qq[iy];
            qx = qq[ix];
            if constexpr (C_ADD) {
              if (addb) {
                qy = qy + addb[iy];
                qx = qx + addb[ix];
              }
            }
            FV3_LANDED(qy);
            FV3_LANDED(qx);
            cur[l].qy = qy;
          }
          if constexpr (C_FD && !XE) {
            if (fdm) {
              // ---- del-n chain, own-lane part: d2 of iteration s on row r-s, its y flux at face r-s (del6_stream phase A)
              const Real du0 = RG(RG_DU, r)[lane], du1 = RG(RG_DU, r - 1)[lane], du2 = RG(RG_DU, r - 2)[lane];
              const Real ra1 = RG(RG_RA, r - 1)[lane], ra2 = RG(RG_RA, r - 2)[lane];
              const Real d0c = dcoef * qraw;
              const Real fyc0 = du0 * (sd0[l] - d0c);
              const Real gxe0 = FV3_LANE_SHL(1, gx0, l, lane), gxe1 = FV3_LANE_SHL(1, gx1, l, lane);
              const Real d2c1 = (gx0[l] - gxe0 + gy0[l] - fyc0) * ra1;
              const Real fyc1 = du1 * (d2c1 - sd1[l]);
              const Real d2c2 = (gx1[l] - gxe1 + gy1[l] - fyc1) * ra2;
              dyf[l] = du2 * (d2c2 - sd2[l]);
              gy0[l] = fyc0;
              gy1[l] = fyc1;
              sd0[l] = d0c;
              sd1[l] = d2c1;
              sd2[l] = d2c2;
            }
          }
          w2[l] = w3[l];
          w3[l] = w4[l];
          w4[l] = w5[l];
          w5[l] = qy;
          Real al_new;
          if (y_edge) {
            const MPtr dyab = gp->dya + m2;
            auto My = [&](int s_) { return dyab[pcol[l] + (unsigned)(s_ * sj32)]; };
            al_new = ppm_al_win(w2[l], w3[l], w4[l], w5[l], My, sy, S, N, npy);
            FV3_LANDED(al_new);
          } else {
            al_new = PPM_P1 * (w3[l] + w4[l]) + PPM_P2 * (w2[l] + w5[l]);
          }
          const PpmCell co = ppm_cell(al_q[l], al_new, w3[l], hord);
          al_q[l] = al_new;
          fyin[l] = ppm_face(cq[l], co, cur[l].cy);
          cq[l] = co;
          const Real yv = cur[l].yv;
          const Real pn = yv * fyin[l];
          constexpr bool PARK = C_FD && !XE;  // the pure delay lines of the wave live in the LDS ring (makes room for the chain)
          const Real ar3 = PARK ? RG(RG_AR, r - 3)[lane] : a3[l];  // area(i, r-3): the row loaded three steps ago
          const Real qi = (w2[l] * ar3 + p_prev[l] - pn) / (ar3 + y_prev[l] - yv);
          p_prev[l] = pn;
          ypp[l] = y_prev[l];
          y_prev[l] = yv;
          if constexpr (LX) {
            lq[3 + lane] = qx;
            lqi[3 + lane] = qi;
          } else {
            sqx[l] = qx;
            sqi[l] = qi;
          }
        }
        if constexpr (LX) blk.wave_sync();
        FV3_STAMP(3);
        // ---- phase 2: inner x-flux on row r, outer x-flux on row r-3, fx(row r-3)
        const int jr = r - 3;
        const bool fx_row = jr >= ja && jr <= jb && jr <= ny;
        FV3_LANES(blk, lane, l) {
          const Real cx = cur[l].cx, xv = cur[l].xv;
          Real fxin, fxout;
          constexpr bool PARK = C_FD && !XE;
          if constexpr (PARK) {
            cx3[l] = RG(RG_CX, r - 3)[lane];
            xv3[l] = RG(RG_XV, r - 3)[lane];
            fi3[l] = RG(RG_FI, r - 3)[lane];
          }
          if constexpr (C_FD && !XE) {
            if (fdm) {
              // ---- del-n chain, x fluxes (del6_stream phase B); the damping fluxes around the cell (i, r-3) for the epilogue
              const Real dv0 = RG(RG_DV, r)[lane], dv1 = RG(RG_DV, r - 1)[lane], dv2 = RG(RG_DV, r - 2)[lane];
              const Real w0 = FV3_LANE_SHR(1, sd0, l, lane), w1 = FV3_LANE_SHR(1, sd1, l, lane), w2_ = FV3_LANE_SHR(1, sd2, l, lane);
              gx0[l] = dv0 * (w0 - sd0[l]);
              gx1[l] = dv1 * (sd1[l] - w1);
              dxn[l] = dv2 * (sd2[l] - w2_);
              Real ox = dxd[l], oy = dyf[l];  // x flux of (i, r-3), y flux of face (i, r-2)
              if (pz) {
                const int i = i0 - 3 + lane, jr_ = r - 3, jf_ = r - 2;
                const Real *pfx = C_AREA ? zfx : wind_dv_, *pfy = C_AREA ? zfy : wind_du_;  // the staged chain's fluxes on the patches
                if (jr_ >= 1 && jr_ <= ny && on_patch(i, jr_)) ox = (pfx + b)[pcol[l] + (unsigned)(jr_ * sj32)];
                if (i <= nx && on_patch(i, jf_)) oy = (pfy + b)[pcol[l] + (unsigned)(jf_ * sj32)];
                FV3_LANDED(ox);
                FV3_LANDED(oy);
              }
              if constexpr (C_AREA) {
                zx0[l] = ox;
                zxo[l] = ox;
                zy0[l] = zyp[l];
                zy1[l] = oy;
              }
              if constexpr (C_WIND) {
                // the vorticity-damping increments of v (row r-3) / u (face r-2); the damping-heat kernel reads them again: stored
                wdv[l] = ox;
                wdu[l] = oy;
                const int jr_ = r - 3, jf_ = r - 2;
                if constexpr (UST) {
                  fv3_store_sel((const_cast<Real *>(wind_dv_) + b) + (pcol[l] + (unsigned)(jr_ * sj32)), sink0 + lane, jr_ >= ja && jr_ <= jb && jr_ <= ny && own_x[l], ox);
                  fv3_store_sel((const_cast<Real *>(wind_du_) + b) + (pcol[l] + (unsigned)(jf_ * sj32)), sink0 + lane, jf_ >= ja && jf_ <= jb && own_y[l], oy);
                } else {
                  if (jr_ >= ja && jr_ <= jb && jr_ <= ny && own_x[l]) (const_cast<Real *>(wind_dv_) + b)[pcol[l] + (unsigned)(jr_ * sj32)] = ox;
                  if (jf_ >= ja && jf_ <= jb && own_y[l]) (const_cast<Real *>(wind_du_) + b)[pcol[l] + (unsigned)(jf_ * sj32)] = oy;
                }
              }
              dxd[l] = dxn[l];
            }
          }
          if (XE) {
            const int i = i0 - 3 + lane;
            auto EI = [&](int s_) { return s_ <= 2 ? s_ + 1 : s_ - (npx - 2) + 4; };  // ring column of an edge cell
            auto Qx = [&](int s_) { return lq[s_ - i0 + 6]; };
            auto Mx = [&](int s_) { return emr[(r & 3) * 8 + EI(s_)]; };
            fxin = ppm_flux(Qx, Mx, cx, i, W, E, npx, hord);
            auto Qi = [&](int s_) { return lqi[s_ - i0 + 6]; };
            auto Mx3 = [&](int s_) { return emr[((r - 3) & 3) * 8 + EI(s_)]; };
            fxout = ppm_flux(Qi, Mx3, cx3[l], i, W, E, npx, hord);
          } else if constexpr (LX) {
            const Real *a = lq + lane, *bq = lqi + lane;
            fxin = ppm_flux_int(a[0], a[1], a[2], a[3], a[4], a[5], cx, hord);
            fxout = ppm_flux_int(bq[0], bq[1], bq[2], bq[3], bq[4], bq[5], cx3[l], hord);
          } else {
            // the six cells around the face from the neighbouring lanes' registers (DPP wave_shr / wave_shl: no LDS line, no ordering point)
            const Real a0 = FV3_LANE_SHR(3, sqx, l, lane), a1_ = FV3_LANE_SHR(2, sqx, l, lane), a2_ = FV3_LANE_SHR(1, sqx, l, lane), a3_ = sqx[l],
                       a4 = FV3_LANE_SHL(1, sqx, l, lane), a5 = FV3_LANE_SHL(2, sqx, l, lane);
            const Real b0 = FV3_LANE_SHR(3, sqi, l, lane), b1 = FV3_LANE_SHR(2, sqi, l, lane), b2 = FV3_LANE_SHR(1, sqi, l, lane), b3 = sqi[l],
                       b4 = FV3_LANE_SHL(1, sqi, l, lane), b5 = FV3_LANE_SHL(2, sqi, l, lane);
            fxin = ppm_flux_int(a0, a1_, a2_, a3_, a4, a5, cx, hord);
            fxout = ppm_flux_int(b0, b1, b2, b3, b4, b5, cx3[l], hord);
          }
          {
            Real v = (Real)0.5 * (fxout + fi3[l]) * (mfx_ ? o_mx[l] : xv3[l]);
            Real mwest;  // mass of the west cell = the neighbouring lane's mb (read outside the selections: a shuffle needs every lane)
            if constexpr (LX)
              mwest = lane > 0 ? exm[lane - 1] : (Real)0;
            else
              mwest = FV3_LANE_SHR(1, smb, l, lane);
            if (on) v = mass_ ? v + (Real)0.5 * damp * (mwest + mb[l]) * o_dx[l] : v + o_dx[l];
            if (wflux_ && fx_row && own_x[l]) {
              const unsigned p = pcol[l] + (unsigned)(jr * sj32);  // own lanes: ic == i
              (fx + b)[p] = v;
              if (acc_x_) (acc_x_ + b)[p] = o_ax[l] + v;
            }
            if constexpr (UST && C_WIND) {
              const unsigned p = pcol[l] + (unsigned)(jr * sj32);
              const bool ok = fx_row && own_x[l];
              const Real vn = wv[l] * wdy[l] + wkr[l] - wkf[l] - v;
              fv3_store_sel((wind_v_pre_ + b) + p, sink0 + lane, ok, vn);
              fv3_store_sel((wind_v_ + b) + p, sink0 + lane, ok, vn - wdv[l]);
            } else if (wind_v_ && fx_row && own_x[l]) {
              const unsigned p = pcol[l] + (unsigned)(jr * sj32);
              const Real vn = wv[l] * wdy[l] + wkr[l] - wkf[l] - v;
              if (wind_v_pre_) (wind_v_pre_ + b)[p] = vn;
              (wind_v_ + b)[p] = wdamp ? vn - wdv[l] : vn;
            }
            if (epi_out_) {
              fxk[l] = v;
              if constexpr (LX) exf[lane] = v;
              if (area_form_) {
                xjr[l] = xv3[l];
                if constexpr (LX) exj[lane] = xv3[l];
              }
            }
          }
          if constexpr (PARK) {
            RG(RG_FI, r)[lane] = fxin;
            RG(RG_CX, r)[lane] = cx;
            RG(RG_XV, r)[lane] = xv;
          } else {
            fi3[l] = fi2[l];
            fi2[l] = fi1[l];
            fi1[l] = fxin;
            cx3[l] = cx2[l];
            cx2[l] = cx1[l];
            cx1[l] = cx;
            xv3[l] = xv2[l];
            xv2[l] = xv1[l];
            xv1[l] = xv;
          }
          px[l] = xv * fxin;
          if constexpr (LX) {
            exp_[lane] = px[l];
            exx[lane] = xv;
          } else {
            sxv[l] = xv;
          }
        }
        if constexpr (LX) blk.wave_sync();
        FV3_STAMP(4);
        // ---- phase 3: q_j on row r, outer y-flux at face r-2, fy(face r-2)
        const int jf = r - 2;
        const bool fy_row = jf >= ja && jf <= jb;
        FV3_LANES(blk, lane, l) {
          Real p1, x1, fxe = (Real)0, xje = (Real)0;  // the east neighbour's values
          if constexpr (LX) {
            p1 = exp_[lane + 1];
            x1 = exx[lane + 1];
            if (epi_out_) {
              fxe = exf[lane + 1];
              if (area_form_) xje = exj[lane + 1];
            }
          } else {
            p1 = FV3_LANE_SHL(1, px, l, lane);
            x1 = FV3_LANE_SHL(1, sxv, l, lane);
            if (epi_out_) {
              fxe = FV3_LANE_SHL(1, fxk, l, lane);
              if (area_form_) xje = FV3_LANE_SHL(1, xjr, l, lane);
            }
          }
          if constexpr (C_FD && C_AREA && !XE) {
            if (fdm) {
              zx1[l] = FV3_LANE_SHL(1, zxo, l, lane);
              zyp[l] = zy1[l];
            }
          }
          const Real ar = cur[l].ar;
          const Real qj = (cur[l].qy * ar + px[l] - p1) / (ar + cur[l].xv - x1);
          v2[l] = v3[l];
          v3[l] = v4[l];
          v4[l] = v5[l];
          v5[l] = qj;
          Real al_new;
          if (y_edge) {
            const MPtr dyab = gp->dya + m2;
            auto My = [&](int s_) { return dyab[pcol[l] + (unsigned)(s_ * sj32)]; };
            al_new = ppm_al_win(v2[l], v3[l], v4[l], v5[l], My, sy, S, N, npy);
            FV3_LANDED(al_new);
          } else {
            al_new = PPM_P1 * (v3[l] + v4[l]) + PPM_P2 * (v2[l] + v5[l]);
          }
          const PpmCell co = ppm_cell(al_v[l], al_new, v3[l], hord);
          al_v[l] = al_new;
          const Real fyout = ppm_face(cv[l], co, cur[l].cy);
          cv[l] = co;
          {
            Real v = (Real)0.5 * (fyout + fyin[l]) * (mfy_ ? o_my[l] : cur[l].yv);
            if (on) v = mass_ ? v + (Real)0.5 * damp * (mb[l] + o_mc[l]) * o_dy[l] : v + o_dy[l];
            if (wflux_ && fy_row && own_y[l]) {
              const unsigned p = pcol[l] + (unsigned)(jf * sj32);
              (fy + b)[p] = v;
              if (acc_y_) (acc_y_ + b)[p] = o_ay[l] + v;
            }
            if constexpr (UST && C_WIND) {
              const unsigned p = pcol[l] + (unsigned)(jf * sj32);
              const bool ok = fy_row && own_y[l];
              const Real ke_e = FV3_LANE_SHL(1, wkf, l, lane);  // ke(i + 1, jf): the neighbouring lane loaded it as its ke(i, jf)
              const Real un = wu[l] * wdx[l] + wkf[l] - ke_e + v;
              fv3_store_sel((wind_u_pre_ + b) + p, sink0 + lane, ok, un);
              fv3_store_sel((wind_u_ + b) + p, sink0 + lane, ok, un + wdu[l]);
            } else if (wind_u_ && fy_row && own_y[l]) {
              const unsigned p = pcol[l] + (unsigned)(jf * sj32);
              const Real un = wu[l] * wdx[l] + wkf[l] - wke[l] + v;
              if (wind_u_pre_) (wind_u_pre_ + b)[p] = un;
              (wind_u_ + b)[p] = wdamp ? un + wdu[l] : un;
            }
            if (epi_out_) {
              // flux-form update of the cell (i, r-3): its west / south fluxes are fxk / fyp, east from lane + 1, north = v
              if constexpr (UST && C_AREA) {
                const Real qc = w2[l];  // q(i, r-3)
                const Real ar_ = RG(RG_AR, r - 3)[lane];
                const Real ra_x = ar_ + xjr[l] - xje, ra_y = ar_ + ypp[l] - cur[l].yv;
                Real z = (qc * ar_ + fxk[l] - fxe + fyp[l] - v) / (ra_x + ra_y - ar_);
                z = z + (zx0[l] - zx1[l] + zy0[l] - zy1[l]) * era[l];
                fv3_store_sel((epi_out_ + b) + (pcol[l] + (unsigned)(jr * sj32)), sink0 + lane, fx_row && own_y[l], z);
              } else if (fx_row && own_y[l]) {
                const Real qc = w2[l];  // q(i, r-3)
                const Real mu = epi_mult_ ? (epi_mult_ == mass_ ? mb[l] : emu[l]) : (Real)1;
                if (area_form_) {
                  const Real ar_ = (C_FD && !XE) ? RG(RG_AR, r - 3)[lane] : a3[l];
                  const Real ra_x = ar_ + xjr[l] - xje, ra_y = ar_ + ypp[l] - cur[l].yv;
                  Real z = (qc * ar_ + fxk[l] - fxe + fyp[l] - v) / (ra_x + ra_y - ar_);
                  if (zdamp) z = z + (zx0[l] - zx1[l] + zy0[l] - zy1[l]) * era[l];
                  (epi_out_ + b)[pcol[l] + (unsigned)(jr * sj32)] = z;
                } else {
                  const Real dv_ = (fxk[l] - fxe + fyp[l] - v) * era[l];
                  (epi_out_ + b)[pcol[l] + (unsigned)(jr * sj32)] = epi_mult_ ? mu * qc + dv_ : qc + dv_;
                }
              }
              fyp[l] = v;
            }
          }
          mb[l] = o_mc[l];
          if constexpr (LX) {
            if (on && mass_) exm[lane] = mb[l];
          } else {
            smb[l] = mb[l];
          }
          if constexpr (C_FD && !XE) {
            RG(RG_AR, r)[lane] = cur[l].ar;
          } else {
            a3[l] = a2[l];
            a2[l] = a1[l];
            a1[l] = cur[l].ar;
          }
        }
        if constexpr (LX) blk.wave_sync();
        FV3_STAMP(5);
      };
      // The prefetched rows rotate through TS_PF + 1 register sets (cur <- nxt <- nx2).  Rolled, the rotation is
      // register copies at the loop head, and a copy of a row still in flight is a full vmcnt(0) drain every
      // step; unrolled by the rotation period the copies become renames (interior strips only: code size).
      // Measured on MI355X: 1.5 % -- with every load of the march redirected to one cached row the kernel is
      // still 6 of its 8.3 ms, i.e. the march is bound by its own issue + LDS-exchange latency at 2 waves / SIMD,
      // not by HBM latency (prefetching the optional inputs as well, at one or two steps, changed nothing).
      // (the FA instantiation of the wind form: rolled too -- unrolled it spilled 15 - 17 registers: d_sw 53.1 -> 52.1 ms rolled; the
      //  area form has the registers and is faster unrolled: update_dz_d 11.15 ms against 11.6 rolled -- same-box A/B, round 4)
      constexpr bool ROLLED = XE || (FA && C_WIND);
      if constexpr (ROLLED) {
        for (int r = ja - 3; r <= r_end; ++r) step(r);
      } else {
        // (trailing steps past r_end: their loads are clamped to r_end and every store is masked by the
        // owned-row tests, so they only keep the trip count a multiple of the period)
        for (int r = ja - 3; r <= r_end; r += TS_PF + 1) {
          step(r);
          step(r + 1);
          if (TS_PF == 2) step(r + 2);
        }
      }
    };
    if (W || E)
      march(std::true_type{});
    else
      march(std::false_type{});
#if defined(FV3_STAMPS) && !defined(FV3_HOST_EMU)
    FV3_STAMP_FLUSH(st_buf, st_kid, blk.tid);
#endif
  });
}

// Will the FD forms of tp2d (TpEpi::fd != 0, FA contract) run the round-5 march of fv3_tp2x.hip?  It serves every tile of the sub-domains, so the
// callers then compute the chain's fluxes only on the cube-corner patches (del6_vt_flux_patches) instead of on the whole tile-edge strips.
bool tp2d_fd_lean(const fv3_ctx *c, int hord, int k0, int k1) {
  // (the contract of the FD forms -- chain of order 2 switched on at every level of the call -- checked on the host tables)
  for (int k = k0; k <= k1; ++k)
    if (k >= (int)c->nord_v_h.size() || !(c->nord_v_h[k] == 2 && c->damp_vt_h[k] > 1.0e-5)) return false;
  static const bool hc_off = getenv("FV3_HORD_CONST") && getenv("FV3_HORD_CONST")[0] == '0';
  static const bool fa_off = getenv("FV3_TP2D_FA") && getenv("FV3_TP2D_FA")[0] == '0';
  const char *me = getenv("FV3_TP2D_MARCH");  // (read per call: the parity test flips it)
  return hord == 6 && !hc_off && !fa_off && !TS_LDS_ONLY && !(me && !strcmp(me, "old"));
}

static void tp2d_stream(fv3_ctx *c, fv3_stream_t s, const Real *q, const Real *crx, const Real *cry, const Real *xfx, const Real *yfx, Real *fx, Real *fy,
                        const Real *mfx, const Real *mfy, const Real *mass, int hord, const Deln *dn, int k0, int k1, const TpEpi *epi) {
  unsigned m = 0;
  if (mfx) m |= TF_MFX;
  if (dn) m |= TF_DAMP;
  if (mass) m |= TF_MASS;
  if (!epi || epi->write_flux) m |= TF_WFLUX;
  if (epi) {
    if (epi->out) m |= TF_EPI;
    if (epi->area_form) m |= TF_AREA;
    if (epi->wind_u) m |= TF_WIND;
    if (epi->acc_x) m |= TF_ACC;
    if (epi->fd && ((epi->area_form && epi->zfx) || (epi->wind_u && epi->wind_du))) m |= TF_FD;
  }
#define TP_CASE(F)                                                                                          \
  case (F):                                                                                                 \
    tp2d_stream_t<(F)>(c, s, q, crx, cry, xfx, yfx, fx, fy, mfx, mfy, mass, hord, dn, k0, k1, epi); \
    return;
  // the two big launches of the acoustic sub-step with the PPM order as a constant (reference default 6; FV3_HORD_CONST=0: A/B)
  static const bool hc_off = getenv("FV3_HORD_CONST") && getenv("FV3_HORD_CONST")[0] == '0';
  // (FA: what the two callers of the FD forms guarantee -- fv3_update_dz_d / fv3_d_sw_out pass only levels whose chain is on)
  static const bool fa_off_env = getenv("FV3_TP2D_FA") && getenv("FV3_TP2D_FA")[0] == '0';
  if (hord == 6 && !hc_off && !TS_LDS_ONLY) {
    // Round 5: the FA forms run the march of fv3_tp2x.hip (every tile: it evaluates the W / E one-sided formulas in its lanes and the cube-corner
    // remaps / patch fluxes in its general steps).  FV3_TP2D_MARCH=old: the round-4 kernel (A/B; read per call).
    // (FA is the callers' contract "the chain is switched on and of order 2 on every level of this call"; it is CHECKED here on the host
    //  tables before it is taken as a compile-time fact -- a caller that passes an undamped level gets the general form)
    bool fa_ok = true;
    for (int k = k0; k <= k1 && k < (int)c->nord_v_h.size(); ++k) fa_ok = fa_ok && c->nord_v_h[k] == 2 && c->damp_vt_h[k] > 1.0e-5;
    const bool fa_off = fa_off_env || !fa_ok;
    const bool sx_on = tp2d_fd_lean(c, hord, k0, k1) && epi && epi->fd_coef;
    if (m == (TF_EPI | TF_AREA | TF_FD)) {
      if (!fa_off && epi->area_form && epi->zfx && epi->out) {
        if (sx_on)
          tp2d_single_march(c, s, 2, q, crx, cry, xfx, yfx, k0, k1, epi);
        else
          tp2d_stream_t<(TF_EPI | TF_AREA | TF_FD), 6, true>(c, s, q, crx, cry, xfx, yfx, fx, fy, mfx, mfy, mass, hord, dn, k0, k1, epi);
      } else
        tp2d_stream_t<(TF_EPI | TF_AREA | TF_FD), 6>(c, s, q, crx, cry, xfx, yfx, fx, fy, mfx, mfy, mass, hord, dn, k0, k1, epi);
      return;
    }
    if (m == (TF_WIND | TF_FD)) {
      if (!fa_off && epi->wind_du && epi->wind_u_pre && epi->wind_v_pre) {
        if (sx_on && epi->fd_add)
          tp2d_single_march(c, s, 1, q, crx, cry, xfx, yfx, k0, k1, epi, epi->heat);
        else
          tp2d_stream_t<(TF_WIND | TF_FD), 6, true>(c, s, q, crx, cry, xfx, yfx, fx, fy, mfx, mfy, mass, hord, dn, k0, k1, epi);
      } else
        tp2d_stream_t<(TF_WIND | TF_FD), 6>(c, s, q, crx, cry, xfx, yfx, fx, fy, mfx, mfy, mass, hord, dn, k0, k1, epi);
      return;
    }
  }
  switch (m) {
    TP_CASE(TF_WFLUX)                                             // plain transport (C entry)
    TP_CASE(TF_DAMP | TF_WFLUX)                                   // damped (C entry)
    TP_CASE(TF_MFX | TF_MASS | TF_DAMP | TF_WFLUX)                // mass-flux weighted + damped (C entry)
    TP_CASE(TF_DAMP | TF_EPI | TF_WFLUX | TF_ACC)                 // d_sw: delp
    TP_CASE(TF_MFX | TF_EPI)                                      // d_sw: w
    TP_CASE(TF_MFX | TF_MASS | TF_DAMP | TF_EPI)                  // d_sw: q_con, pt
    TP_CASE(TF_WIND)                                              // d_sw: absolute vorticity + wind update
    TP_CASE(TF_EPI | TF_AREA)                                     // update_dz_d: interface heights
    TP_CASE(TF_EPI | TF_AREA | TF_FD)                             // update_dz_d: interface heights, their del-n chain inside the march
    TP_CASE(TF_WIND | TF_FD)                                      // d_sw: vorticity (q + f0 formed on load) + wind update, the vorticity's del-n chain inside the march
    default:
      tp2d_stream_t<TF_ALL>(c, s, q, crx, cry, xfx, yfx, fx, fy, mfx, mfy, mass, hord, dn, k0, k1, epi);
  }
#undef TP_CASE
}

void tp2d(fv3_ctx *c, fv3_stream_t s, const Real *q, const Real *crx, const Real *cry, const Real *xfx, const Real *yfx, Real *fx, Real *fy,
          const Real *mfx, const Real *mfy, const Real *mass, int hord, const Deln *dn, int k0, int k1, const TpEpi *epi) {
  // FV3_TP2D_MODE = staged | stream (default): the staged form is the reference / A-B path
  static const char *mode_env = getenv("FV3_TP2D_MODE");
  static const bool staged = mode_env && !strcmp(mode_env, "staged");
  if (k0 > k1) return;
  if (!staged) {
    tp2d_stream(c, s, q, crx, cry, xfx, yfx, fx, fy, mfx, mfy, mass, hord, dn, k0, k1, epi);
    return;
  }
  tp2d_staged(c, s, q, crx, cry, xfx, yfx, fx, fy, mfx, mfy, mass, hord, dn, k0, k1);
  if (epi) {  // the non-marching forms always store the fluxes; apply the update in a second pass
    const Geo g = c->g;
    Real *out = epi->out;
    const Real *mult = epi->mult;
    Real *ax = epi->acc_x, *ay = epi->acc_y;
    const bool aform = epi->area_form;
    const Real *zfx_ = epi->zfx, *zfy_ = epi->zfy, *zon_ = epi->zon;
    Real *wu_ = epi->wind_u, *wv_ = epi->wind_v;
    const Real *wk_ = epi->wind_ke;
    const Real *wdu_ = epi->wind_du, *wdv_ = epi->wind_dv, *wdon_ = epi->wind_don;
    Real *wup_ = epi->wind_u_pre, *wvp_ = epi->wind_v_pre;
    launch3(c, s, Box{1, g.nx + 1, 1, g.ny + 1, k0, k1}, [=] FV3_HD(int t, int k, int i, int j) {
      const long b = t * g.st + k * g.sk;
      const unsigned p = IX(i, j);
      if (wu_) {
        const long m2 = t * g.st2;
        const bool wd = wdu_ && wdon_[k] > (Real)1.0e-5;
        if (i <= g.nx) {
          const Real un = (wu_ + b)[p] * (g.dx + m2)[p] + (wk_ + b)[p] - (wk_ + b)[IX(i + 1, j)] + (fy + b)[p];
          if (wup_) (wup_ + b)[p] = un;
          (wu_ + b)[p] = wd ? un + (wdu_ + b)[p] : un;
        }
        if (j <= g.ny) {
          const Real vn = (wv_ + b)[p] * (g.dy + m2)[p] + (wk_ + b)[p] - (wk_ + b)[IX(i, j + 1)] - (fx + b)[p];
          if (wvp_) (wvp_ + b)[p] = vn;
          (wv_ + b)[p] = wd ? vn - (wdv_ + b)[p] : vn;
        }
      }
      if (out && i <= g.nx && j <= g.ny) {
        const unsigned pe_ = IX(i + 1, j), pn = IX(i, j + 1);
        if (aform) {
          const Real ar = g.area[t * g.st2 + p];
          const Real ra_x = ar + (xfx + b)[p] - (xfx + b)[pe_], ra_y = ar + (yfx + b)[p] - (yfx + b)[pn];
          Real z = ((q + b)[p] * ar + (fx + b)[p] - (fx + b)[pe_] + (fy + b)[p] - (fy + b)[pn]) / (ra_x + ra_y - ar);
          if (zfx_ && zon_[k] > (Real)1.0e-5) z = z + ((zfx_ + b)[p] - (zfx_ + b)[pe_] + (zfy_ + b)[p] - (zfy_ + b)[pn]) * g.rarea[t * g.st2 + p];
          (out + b)[p] = z;
        } else {
          const Real dv_ = ((fx + b)[p] - (fx + b)[pe_] + (fy + b)[p] - (fy + b)[pn]) * g.rarea[t * g.st2 + p];
          (out + b)[p] = mult ? (mult + b)[p] * (q + b)[p] + dv_ : (q + b)[p] + dv_;
        }
      }
      if (ax && j <= g.ny) (ax + b)[p] += (fx + b)[p];
      if (ay && i <= g.nx) (ay + b)[p] += (fy + b)[p];
    });
  }
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
  if (hord == 6 && fv3_alt("smt5_lim_fac")) hord = 7;  // (FV3_ALT: see fv3_ppm.h)
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
