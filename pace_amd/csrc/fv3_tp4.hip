// fv3_tp4.hip -- d_sw's four scalar transports (air mass delp, vertical velocity w, condensate q_con, potential
// temperature pt) fused into ONE marching wave kernel.  CPU twin: oracle/fv3_oracle/d_sw.py (the four fv_tp_2d calls
// and the divisions by the new air mass of d_sw_levels).  [SURVEY A.3.2 - A.3.4, A.4]
//
// Why: as four launches of tp2d_stream_t the transports re-read the Courant numbers, the area fluxes and the cell areas
// four times, the air-mass fluxes go out to memory to come back three times as the mass fluxes of the other tracers, the
// flux-form updates go out as delp * q fields and a fifth kernel divides them by the new air mass: 141 GB + 29 GB of the
// 352 GB one d_sw call moved at C768 (PMC, round 1) for 8 fields of algorithmic traffic.  Here a wave carries the four
// tracers of its strip together:
//   * crx / cry / xfx / yfx / area are loaded once per row for the four tracers (and the two denominators of the
//     cross-direction updates are shared);
//   * the final air-mass flux of a face is used in the same lane, in the same step, as the mass flux of w / q_con / pt
//     (a pointwise dependency) and only leaves the wave as the mfx / mfy accumulation;
//   * the old air mass is the damping weight and the epilogue multiplier of q_con / pt: it is already in the wave as the
//     delp tracer's own window;
//   * the epilogue of a cell forms delp_new, (delp * q + div) / delp_new for the three tracers, w's del-n increment and
//     the heat it dissipates -- the post-transport kernel of round 1 is gone.
// One wave per SIMD (about 440 VGPRs): the four tracers are four independent dependency chains, which is the
// instruction-level parallelism the single-tracer march lacks at two waves per SIMD, and every LDS exchange
// (one ordering point) now serves four rows.  The arithmetic of each tracer is expression for expression the one of
// tp2d_stream_t (same operation order: bitwise equal results, checked by tests/test_gpu_invariants.py and the A/B switch
// FV3_DSW_SCALARS=separate).
#include "fv3_ops.h"
#include "fv3_ppm.h"

#define Q4_OUT 58
#define Q4_LINE (FV3_WAVE + 6)
#define Q4_PF 2
#define Q4_NT 4  // tracer slots: 0 = delp, 1 = w, 2 = q_con, 3 = pt

void dsw_scalars_stream(fv3_ctx *c, fv3_stream_t s, const DswScalars &a_) {
  const Geo g = c->g;
  const DswScalars a = a_;
  const int nk = g.nz;
  const int nstrip = (g.nx + 1 + Q4_OUT - 1) / Q4_OUT;
  const int seg = fv3_pick_seg((long)nstrip * ((g.ny + 63) / 64) * g.nsub * nk, 1);
  const int nseg = (g.ny + seg - 1) / seg;
  // LDS: per tracer the two x-sweep row lines (q on the new row, q_i three rows behind), xfx * fx_in and the final x flux;
  // shared: xfx, the old air mass of row r-3, the tile-edge dxa ring
  const size_t smem = sizeof(Real) * (Q4_NT * (2 * Q4_LINE + 2 * (FV3_WAVE + 1)) + 2 * (FV3_WAVE + 1) + 32);
  const Geo *gp = c->g_dev;
  const int nx = g.nx, ny = g.ny, nh = g.nh, npx = g.npx, npy = g.npy, sj32 = g.sj32, go = g.o;
  const long st = g.st, sk = g.sk, st2 = g.st2;
  const MPtr area = g.area, gdxa = g.dxa, rarea = g.rarea;
  const Real *damp_w_k = g.damp_w, *ke_bg_k = g.ke_bg;
  launch_waves<1>(c, s, nstrip, nseg, g.nsub * nk, smem, [=] FV3_HD(const Blk &blk, char *smem_) {
    const int t = blk.bz / nk, k = blk.bz - t * nk;
    const int fl = gp->flags[t];
    const long b = t * st + k * sk, m2 = t * st2;
    const int i0 = 1 + blk.bx * Q4_OUT;
    const int ja = 1 + blk.by * seg;
    const int jb = blk.by == nseg - 1 ? ny + 1 : ja + seg - 1;
    const int ied = nx + nh, jsd = 1 - nh, jed = ny + nh;
    Real *lq[Q4_NT], *lqi[Q4_NT], *exp_[Q4_NT], *exf[Q4_NT];
    {
      Real *p = (Real *)smem_;
#pragma unroll
      for (int n = 0; n < Q4_NT; ++n) {
        lq[n] = p;
        lqi[n] = p + Q4_LINE;
        exp_[n] = p + 2 * Q4_LINE;
        exf[n] = p + 2 * Q4_LINE + FV3_WAVE + 1;
        p += 2 * Q4_LINE + 2 * (FV3_WAVE + 1);
      }
    }
    Real *exx = (Real *)smem_ + Q4_NT * (2 * Q4_LINE + 2 * (FV3_WAVE + 1));  // xfx of the lane's face (read by lane - 1)
    Real *exm = exx + FV3_WAVE + 1;                                          // old delp(i, r-3) of the lane (read by lane + 1)
    Real *emr = exm + FV3_WAVE + 1;                                          // tile-edge strips: dxa ring (4 rows x 8 cells)
    const Real *qin[Q4_NT] = {a.delp + b, a.w + b, a.q_con + b, a.pt + b};
    const int hord[Q4_NT] = {a.hord_dp, a.hord_vt, a.hord_dp, a.hord_tm};
    const Real *crxb = a.crx + b, *cryb = a.cry + b, *xfxb = a.xfx + b, *yfxb = a.yfx + b;
    const MPtr areab = area + m2;
    const bool W = (fl & FV3_W) && i0 <= 3, E = (fl & FV3_E) && i0 + Q4_OUT + 1 >= npx - 1;
    const bool S = fl & FV3_S, N = fl & FV3_N;
    const bool halo_cols = i0 - 3 < 1 || i0 + FV3_WAVE - 4 > nx;
    const bool on_vt = deln_on(a.dn_vt, k), on_t = deln_on(a.dn_t, k);
    const Real damp_vt = on_vt ? deln_damp(a.dn_vt, k) : (Real)0, damp_t = on_t ? deln_damp(a.dn_t, k) : (Real)0;
    const bool on_w = damp_w_k[k] > (Real)1.0e-5;
    const Real dd8 = ke_bg_k[k] * fabs(a.dt);
    const int r_end = jb + 3 < jed ? jb + 3 : jed;

    struct Row {
      Real qy[Q4_NT], cx, xv, ar, cy, yv, em;
    };
    // inputs consumed at the end of a step, loaded at its top ahead of the prefetch (loads return in order: waiting for
    // them leaves the prefetched rows in flight)
    Real o_ax[FV3_LPT], o_ay[FV3_LPT];                  // accumulated mass fluxes
    Real o_dx[Q4_NT][FV3_LPT], o_dy[Q4_NT][FV3_LPT];    // damping fluxes (slot 1 unused: w's enter as an increment)
    Real era[FV3_LPT];                                  // rarea(i, r-3)
    Real zx0[FV3_LPT], zx1[FV3_LPT], zy0[FV3_LPT], zy1[FV3_LPT];  // w's damping fluxes around the cell (i, r-3)
    Row nxt[FV3_LPT], nx2[FV3_LPT], cur[FV3_LPT];
    Real a1[FV3_LPT], a2[FV3_LPT], a3[FV3_LPT];
    Real w2[Q4_NT][FV3_LPT], w3[Q4_NT][FV3_LPT], w4[Q4_NT][FV3_LPT], w5[Q4_NT][FV3_LPT], al_q[Q4_NT][FV3_LPT];
    Real v2[Q4_NT][FV3_LPT], v3[Q4_NT][FV3_LPT], v4[Q4_NT][FV3_LPT], v5[Q4_NT][FV3_LPT], al_v[Q4_NT][FV3_LPT];
    PpmCell cq[Q4_NT][FV3_LPT], cv[Q4_NT][FV3_LPT];
    Real p_prev[Q4_NT][FV3_LPT], y_prev[FV3_LPT];
    Real fi1[Q4_NT][FV3_LPT], fi2[Q4_NT][FV3_LPT], fi3[Q4_NT][FV3_LPT];
    Real cx1[FV3_LPT], cx2[FV3_LPT], cx3[FV3_LPT], xv1[FV3_LPT], xv2[FV3_LPT], xv3[FV3_LPT];
    Real fyin[Q4_NT][FV3_LPT], px[Q4_NT][FV3_LPT];
    Real fxk[Q4_NT][FV3_LPT], fyp[Q4_NT][FV3_LPT];  // x flux of the west face / y flux of the south face of cell (i, r-3)
    unsigned pcol[FV3_LPT];
    bool own_x[FV3_LPT], own_y[FV3_LPT];

    const MPtr dxab0 = gdxa + m2;
    auto load_row = [&](int r, int l, int lane) -> Row {
      const int rf = r - 2 < jsd ? jsd : r - 2;
      const unsigned p0 = pcol[l] + (unsigned)(r * sj32), pf = pcol[l] + (unsigned)(rf * sj32);
      Row w;
#pragma unroll
      for (int n = 0; n < Q4_NT; ++n) w.qy[n] = qin[n][p0];
      w.cx = crxb[p0];
      w.xv = xfxb[p0];
      w.ar = areab[p0];
      w.cy = cryb[pf];
      w.yv = yfxb[pf];
      w.em = (Real)1;
      if ((W || E) && lane < 8) {
        const bool have = lane < 4 ? W : E;
        const int sc = lane < 4 ? lane - 1 : npx - 2 + (lane - 4);
        if (have) w.em = dxab0[(unsigned)((r + go) * sj32 + sc + go)];
      }
      return w;
    };
    FV3_LANES(blk, lane, l) {
      const int i = i0 - 3 + lane, ic = i < ied ? i : ied;
      pcol[l] = (unsigned)(go * sj32 + ic + go);
      own_x[l] = i >= i0 && i < i0 + Q4_OUT && i <= nx + 1;
      own_y[l] = i >= i0 && i < i0 + Q4_OUT && i <= nx;
      a1[l] = a2[l] = a3[l] = (Real)1;
      o_ax[l] = o_ay[l] = era[l] = zx0[l] = zx1[l] = zy0[l] = zy1[l] = y_prev[l] = (Real)0;
      cx1[l] = cx2[l] = cx3[l] = xv1[l] = xv2[l] = xv3[l] = (Real)0;
#pragma unroll
      for (int n = 0; n < Q4_NT; ++n) {
        w2[n][l] = w3[n][l] = w4[n][l] = w5[n][l] = al_q[n][l] = v2[n][l] = v3[n][l] = v4[n][l] = v5[n][l] = al_v[n][l] = (Real)0;
        cq[n][l] = cv[n][l] = PpmCell{(Real)0, (Real)0, (Real)0, false};
        p_prev[n][l] = fi1[n][l] = fi2[n][l] = fi3[n][l] = fyin[n][l] = px[n][l] = fxk[n][l] = fyp[n][l] = o_dx[n][l] = o_dy[n][l] = (Real)0;
        if (lane == 0) exf[n][FV3_WAVE] = exp_[n][FV3_WAVE] = (Real)0;
        if (lane < 3) lq[n][lane] = lqi[n][lane] = lq[n][FV3_WAVE + 3 + lane] = lqi[n][FV3_WAVE + 3 + lane] = (Real)0;
      }
      if (lane == 0) exx[FV3_WAVE] = (Real)0;
      nxt[l] = load_row(ja - 3, l, lane);
      nx2[l] = load_row(ja - 2 < r_end ? ja - 2 : r_end, l, lane);
      if (lane < 32) emr[lane] = (Real)1;
      exm[lane] = (Real)0;
    }

    auto march = [&](auto xe_tag) {
      constexpr bool XE = decltype(xe_tag)::value;
      auto step = [&](int r) {
        const int r3 = r - 3 < jsd ? jsd : r - 3;
        const int rn = r + Q4_PF < r_end ? r + Q4_PF : r_end;
        const int sy = r - 1;
        const bool y_edge = (S && sy >= 0 && sy <= 2) || (N && sy >= npy - 1 && sy <= npy + 1);
        const bool corner_row = halo_cols && (r < 1 || r > ny);
        // ---- phase 1: prefetch row r+2; inner y-fluxes at face r-2, q_i at row r-3 (four tracers)
        FV3_LANES(blk, lane, l) {
          {
            const int rf = r - 2 < jsd ? jsd : r - 2;
            const unsigned p3 = pcol[l] + (unsigned)(r3 * sj32), pf = pcol[l] + (unsigned)(rf * sj32);
            o_ax[l] = (a.mfx + b)[p3];
            o_ay[l] = (a.mfy + b)[pf];
            if (on_vt) {
              o_dx[0][l] = (a.dpx + b)[p3];
              o_dy[0][l] = (a.dpy + b)[pf];
              o_dx[3][l] = (a.dtx + b)[p3];
              o_dy[3][l] = (a.dty + b)[pf];
            }
            if (on_t) {
              o_dx[2][l] = (a.dqx + b)[p3];
              o_dy[2][l] = (a.dqy + b)[pf];
            }
            era[l] = (rarea + m2)[p3];
            if (on_w) {
              zx0[l] = (a.dwx + b)[p3];
              zx1[l] = (a.dwx + b)[p3 + 1];
              zy0[l] = (a.dwy + b)[p3];
              zy1[l] = (a.dwy + b)[p3 + (unsigned)sj32];
            }
          }
          cur[l] = nxt[l];
          nxt[l] = nx2[l];
          nx2[l] = load_row(rn, l, lane);
          if (XE && lane < 8) emr[(r & 3) * 8 + lane] = cur[l].em;
          const Real yv = cur[l].yv;
          const Real ar3 = a3[l];
          const Real den_y = ar3 + y_prev[l] - yv;
#pragma unroll
          for (int n = 0; n < Q4_NT; ++n) {
            Real qy = cur[l].qy[n], qx = qy;
            if (corner_row) {
              const int i = i0 - 3 + lane, ic = i < ied ? i : ied;
              const int rc = r < jed ? r : jed;
              qy = cc<2>(qin[n], *gp, fl, ic, rc);
              qx = cc<1>(qin[n], *gp, fl, ic, rc);
              cur[l].qy[n] = qy;
            }
            w2[n][l] = w3[n][l];
            w3[n][l] = w4[n][l];
            w4[n][l] = w5[n][l];
            w5[n][l] = qy;
            Real al_new;
            if (y_edge) {
              const MPtr dyab = gp->dya + m2;
              auto My = [&](int s_) { return dyab[pcol[l] + (unsigned)(s_ * sj32)]; };
              al_new = ppm_al_win(w2[n][l], w3[n][l], w4[n][l], w5[n][l], My, sy, S, N, npy);
            } else {
              al_new = PPM_P1 * (w3[n][l] + w4[n][l]) + PPM_P2 * (w2[n][l] + w5[n][l]);
            }
            const PpmCell co = ppm_cell(al_q[n][l], al_new, w3[n][l], hord[n]);
            al_q[n][l] = al_new;
            fyin[n][l] = ppm_face(cq[n][l], co, cur[l].cy);
            cq[n][l] = co;
            const Real pn = yv * fyin[n][l];
            const Real qi = (w2[n][l] * ar3 + p_prev[n][l] - pn) / den_y;
            p_prev[n][l] = pn;
            lq[n][3 + lane] = qx;
            lqi[n][3 + lane] = qi;
          }
          y_prev[l] = yv;
          exm[lane] = w2[0][l];  // old air mass of the cell (i, r-3)
        }
        blk.wave_sync();
        // ---- phase 2: inner x-fluxes on row r, outer x-fluxes on row r-3, final x fluxes of row r-3
        const int jr = r - 3;
        const bool fx_row = jr >= ja && jr <= jb && jr <= ny;
        FV3_LANES(blk, lane, l) {
          const Real cx = cur[l].cx, xv = cur[l].xv;
          Real fxin[Q4_NT], fxout[Q4_NT];
#pragma unroll
          for (int n = 0; n < Q4_NT; ++n) {
            if (XE) {
              const int i = i0 - 3 + lane;
              auto EI = [&](int s_) { return s_ <= 2 ? s_ + 1 : s_ - (npx - 2) + 4; };
              auto Qx = [&](int s_) { return lq[n][s_ - i0 + 6]; };
              auto Mx = [&](int s_) { return emr[(r & 3) * 8 + EI(s_)]; };
              fxin[n] = ppm_flux(Qx, Mx, cx, i, W, E, npx, hord[n]);
              auto Qi = [&](int s_) { return lqi[n][s_ - i0 + 6]; };
              auto Mx3 = [&](int s_) { return emr[((r - 3) & 3) * 8 + EI(s_)]; };
              fxout[n] = ppm_flux(Qi, Mx3, cx3[l], i, W, E, npx, hord[n]);
            } else {
              const Real *aq = lq[n] + lane, *bq = lqi[n] + lane;
              fxin[n] = ppm_flux_int(aq[0], aq[1], aq[2], aq[3], aq[4], aq[5], cx, hord[n]);
              fxout[n] = ppm_flux_int(bq[0], bq[1], bq[2], bq[3], bq[4], bq[5], cx3[l], hord[n]);
            }
          }
          const Real mb = w2[0][l];                                       // old delp(i, r-3)
          const Real mw = (lane > 0 ? exm[lane - 1] : (Real)0) + mb;      // + old delp(i-1, r-3)
          // air mass: area-flux weighted, plain damping flux
          Real vm = (Real)0.5 * (fxout[0] + fi3[0][l]) * xv3[l];
          if (on_vt) vm = vm + o_dx[0][l];
          if (fx_row && own_x[l]) (a.mfx + b)[pcol[l] + (unsigned)(jr * sj32)] = o_ax[l] + vm;
          // the three tracers riding on the air-mass flux
          Real vw = (Real)0.5 * (fxout[1] + fi3[1][l]) * vm;
          Real vq = (Real)0.5 * (fxout[2] + fi3[2][l]) * vm;
          if (on_t) vq = vq + (Real)0.5 * damp_t * mw * o_dx[2][l];
          Real vp = (Real)0.5 * (fxout[3] + fi3[3][l]) * vm;
          if (on_vt) vp = vp + (Real)0.5 * damp_vt * mw * o_dx[3][l];
          const Real vx[Q4_NT] = {vm, vw, vq, vp};
#pragma unroll
          for (int n = 0; n < Q4_NT; ++n) {
            fxk[n][l] = vx[n];
            exf[n][lane] = vx[n];
            fi3[n][l] = fi2[n][l];
            fi2[n][l] = fi1[n][l];
            fi1[n][l] = fxin[n];
            px[n][l] = xv * fxin[n];
            exp_[n][lane] = px[n][l];
          }
          cx3[l] = cx2[l];
          cx2[l] = cx1[l];
          cx1[l] = cx;
          xv3[l] = xv2[l];
          xv2[l] = xv1[l];
          xv1[l] = xv;
          exx[lane] = xv;
        }
        blk.wave_sync();
        // ---- phase 3: q_j on row r, outer y-fluxes at face r-2, final y fluxes, the cell update of (i, r-3)
        const int jf = r - 2;
        const bool fy_row = jf >= ja && jf <= jb;
        FV3_LANES(blk, lane, l) {
          const Real x1 = exx[lane + 1];
          const Real ar = cur[l].ar;
          const Real den_x = ar + cur[l].xv - x1;
          Real fyout[Q4_NT];
#pragma unroll
          for (int n = 0; n < Q4_NT; ++n) {
            const Real p1 = exp_[n][lane + 1];
            const Real qj = (cur[l].qy[n] * ar + px[n][l] - p1) / den_x;
            v2[n][l] = v3[n][l];
            v3[n][l] = v4[n][l];
            v4[n][l] = v5[n][l];
            v5[n][l] = qj;
            Real al_new;
            if (y_edge) {
              const MPtr dyab = gp->dya + m2;
              auto My = [&](int s_) { return dyab[pcol[l] + (unsigned)(s_ * sj32)]; };
              al_new = ppm_al_win(v2[n][l], v3[n][l], v4[n][l], v5[n][l], My, sy, S, N, npy);
            } else {
              al_new = PPM_P1 * (v3[n][l] + v4[n][l]) + PPM_P2 * (v2[n][l] + v5[n][l]);
            }
            const PpmCell co = ppm_cell(al_v[n][l], al_new, v3[n][l], hord[n]);
            al_v[n][l] = al_new;
            fyout[n] = ppm_face(cv[n][l], co, cur[l].cy);
            cv[n][l] = co;
          }
          const Real mb = w2[0][l], mc = w3[0][l];  // old delp(i, r-3), old delp(i, r-2)
          Real vm = (Real)0.5 * (fyout[0] + fyin[0][l]) * cur[l].yv;
          if (on_vt) vm = vm + o_dy[0][l];
          if (fy_row && own_y[l]) (a.mfy + b)[pcol[l] + (unsigned)(jf * sj32)] = o_ay[l] + vm;
          Real vw = (Real)0.5 * (fyout[1] + fyin[1][l]) * vm;
          Real vq = (Real)0.5 * (fyout[2] + fyin[2][l]) * vm;
          if (on_t) vq = vq + (Real)0.5 * damp_t * (mb + mc) * o_dy[2][l];
          Real vp = (Real)0.5 * (fyout[3] + fyin[3][l]) * vm;
          if (on_vt) vp = vp + (Real)0.5 * damp_vt * (mb + mc) * o_dy[3][l];
          const Real vy[Q4_NT] = {vm, vw, vq, vp};
          if (fx_row && own_y[l]) {
            // flux-form updates of the cell (i, r-3): west / south fluxes fxk / fyp, east from lane + 1, north = vy
            const unsigned p = pcol[l] + (unsigned)(jr * sj32);
            Real up[Q4_NT];
#pragma unroll
            for (int n = 0; n < Q4_NT; ++n) {
              const Real dv_ = (fxk[n][l] - exf[n][lane + 1] + fyp[n][l] - vy[n]) * era[l];
              up[n] = n == 0 ? w2[0][l] + dv_ : mb * w2[n][l] + dv_;
            }
            const Real dpn = up[0];
            Real wn = up[1] / dpn, hs = (Real)0;
            if (on_w) {
              const Real dwv = (zx0[l] - zx1[l] + zy0[l] - zy1[l]) * era[l];
              hs = dd8 - dwv * (w2[1][l] + (Real)0.5 * dwv);
              wn = wn + dwv;
            }
            (a.o_delp + b)[p] = dpn;
            (a.o_pt + b)[p] = up[3] / dpn;
            (a.o_w + b)[p] = wn;
            (a.heat + b)[p] = hs;
            (a.o_q_con + b)[p] = up[2] / dpn;
          }
#pragma unroll
          for (int n = 0; n < Q4_NT; ++n) fyp[n][l] = vy[n];
          a3[l] = a2[l];
          a2[l] = a1[l];
          a1[l] = cur[l].ar;
        }
        blk.wave_sync();
      };
      if constexpr (XE) {
        for (int r = ja - 3; r <= r_end; ++r) step(r);
      } else {
        for (int r = ja - 3; r <= r_end; r += Q4_PF + 1) {
          step(r);
          step(r + 1);
          step(r + 2);
        }
      }
    };
    if (W || E)
      march(std::true_type{});
    else
      march(std::false_type{});
  });
}
