// fv3_wind.hip -- d_sw's wind-branch stage kernels as ONE march (round 6): cell-mean relative vorticity, corner kinetic energy (xtp_u / ytp_v),
// the nord-fold divergence-damping iteration, the 4th-order corner interpolation of the vorticity and the Smagorinsky-type damping that consumes the
// three of them.  Staged form (fv3_dsw.hip: vort_cells, ke_stream, divdamp_stream, a2b_ord4_t + epilogue): five launches that move 44 GB at C768 L79 to
// produce three fields -- u / v are read by the vorticity launch and again by the KE march, wk is written, read back by the interpolation, the iterated
// divergence is written and read back, ke is written, then read, modified and written again.  Here a wave reads u, v, uc, vc and the divergence of its
// strip once and writes wk, ke (damping included) and the corner damping field once: 5 + 3 field passes instead of 11 + 7.
// Same expressions in the same order as the staged kernels (bitwise equal: FV3_DSW_WINDSTAGE=staged is the A/B switch,
// tests/test_parity.py::test_fused_wind_stage_is_bitwise_the_staged_kernels); CPU twin: oracle/fv3_oracle/d_sw.py (d_sw_levels: vorticity, ke, divergence
// damping), a2b_ord4.py.  [SURVEY A.3.5 - A.3.7, A.13; reference DGridShallowWaterLagrangianDynamics / DivergenceDamping / AGrid2BGridFourthOrder]
//
// Row alignment.  Step r loads v, u, dx of row r, dy / 1/area of row r-1, everything of the KE at row jf = r-2 (uc, vc, cosa, 1/sina, 1/dx, 1/dy) and the
// divergence row r - 2 + nord with its three metric rows.  It then has
//   * the v window r-3 .. r                      -> ytp_v at face jf (the recurrences of ke_stream),
//   * u of rows r, r-1, r-2                      -> xtp_u on row jf (reconstruction shared between neighbouring lanes), wk of cell row r-1,
//   * the wk window r-4 .. r-1                   -> the corner interpolation at corner row jf (the recurrences of a2b_ord4_t),
//   * iteration n of the damping chain on row r - 2 + nord - n (windows of divdamp_stream) -> the iterated divergence of corner row jf,
// i.e. every ingredient of corner (i, jf): ke = 0.5 (vb ytp_v + ub xtp_u) + vd, vd = damp2 * divg + dd8 * divg_iterated.
//
// What stays outside: the three outermost corner rows / columns next to a cube-tile edge (other formulas: one per-point launch afterwards that evaluates
// ke_point + a2b_point + the damping there; the march EXPORTS the iterated divergence on those corners), the 8 x 8 corners next to a cube corner of the
// damping chain (cube-corner terms / halo remaps are not tile-local: the staged chain on a private copy, run BEFORE the march, which reads the patch
// values where its own chain is wrong), the sponge levels (no chain, the absolute-vorticity field of the round-4 transport).
#include <type_traits>

#include "fv3_a2b.h"
#include "fv3_ops.h"
#include "fv3_ppm.h"
#include "fv3_march.h"

namespace {

#define WS_OUT 58
#ifndef WS_WPE
#define WS_WPE 2
#endif
#define WS_NMAX 3
#ifndef WS_PF3
#define WS_PF3 2  // steps the 3-D rows are requested ahead (1: with the metric rows -- ten registers less, for the three-waves-per-SIMD experiment)
#endif

// HC: 0 = the PPM order of xtp_u / ytp_v is a run-time value; 6 = the constant 6 (the reference configurations): the limiter test folds to one comparison
template <int HC>
void wind_stage_march_t(fv3_ctx *c, fv3_stream_t s, const WindStage &a) {
  const Geo g = c->g;
  const int k0 = a.k0, nk = a.k1 - a.k0 + 1;
  if (nk <= 0) return;
  const int nx = g.nx, ny = g.ny, nh = g.nh, npx = g.npx, npy = g.npy, sj32 = g.sj32, go = g.o;
  const int nstrip = (nx + 1 + WS_OUT - 1) / WS_OUT;
  const int seg = fv3_pick_seg((long)nstrip * ((ny + 64) / 64) * g.nsub * nk, WS_WPE);
  const int nseg = (ny + 1 + seg - 1) / seg;
  const long st = g.st, sk = g.sk, st2 = g.st2;
  const Geo *gp = c->g_dev;
  const unsigned char *gflags = c->g_dev->flags;
  const MPtr m_cosa = g.cosa, m_rsina = g.rsina, m_rdx = g.rdx, m_rdy = g.rdy, m_dx = g.dx, m_dy = g.dy, m_ra = g.rarea;
  const MPtr m_du = g.divg_u, m_dv = g.divg_v, m_rac = g.rarea_c;
  const Real *const fu = a.u, *const fv = a.v, *const fuc = a.uc, *const fvc = a.vc, *const fdg = a.divgd;
  Real *const fke = a.ke, *const fvd = a.vdamp, *const fwk = a.wk, *const fdn = a.dnew;
  const int *const nord_k = g.nord;
  const Real *const d2_k = g.d2_divg, *const dd8_k = a.dd8;
  const Real dt = a.dt, dddmp = a.dddmp, da_min_c = g.da_min_c;
  const int hord = HC ? HC : a.hord;
  const bool store_dn = a.store_dn;
  Real *const fus = a.u_side, *const fvs = a.v_side;
  const int side_seg = (a.u_side && a.v_side) ? a.side_seg : 0;
  static const int kb_env = getenv("FV3_KE_KB") ? atoi(getenv("FV3_KE_KB")) : 16;
  const int KB = kb_env > 0 ? (kb_env < nk ? kb_env : nk) : 0;
  const int nblk = KB ? (nk + KB - 1) / KB : 0;
  launch_waves<WS_WPE>(c, s, KB ? KB : nstrip, KB ? nstrip * nseg : nseg, KB ? g.nsub * nblk : g.nsub * nk, 0, [=] FV3_HD(const Blk &blk_, char *) {
    int t, k, bx, by;
    if (KB) {
      t = blk_.bz / nblk;
      const int kk = (blk_.bz - t * nblk) * KB + blk_.bx;
      if (kk >= nk) return;
      k = k0 + kk;
      by = blk_.by / nstrip;
      bx = blk_.by - by * nstrip;
    } else {
      t = blk_.bz / nk;
      k = k0 + (blk_.bz - t * nk);
      bx = blk_.bx;
      by = blk_.by;
    }
    const int nord = nord_k[k];
    if (nord <= 0 || nord > WS_NMAX) return;  // (the caller passes levels that run the chain; guarded all the same)
    const int fl = gflags[t];
    const long b = t * st + k * sk, m2 = t * st2;
    const bool W = fl & FV3_W, E = fl & FV3_E, S = fl & FV3_S, N = fl & FV3_N;
    // corners that take the interior formulas of all three stages (KE: 4 from a tile edge; a2b: 3; the chain: anywhere off the patches)
    const int ia = W ? 4 : 1, ib = E ? npx - 3 : nx + 1, jA = S ? 4 : 1, jB = N ? npy - 3 : ny + 1;
    const int i0 = 1 + bx * WS_OUT;
    const int ja = 1 + by * seg, jb = ja + seg - 1 < ny + 1 ? ja + seg - 1 : ny + 1;
    const bool seg_first = by == 0, seg_last = by == nseg - 1, strip_first = bx == 0, strip_last = bx == nstrip - 1;
    const int isd = 1 - nh, ied = nx + nh, jsd = 1 - nh, jed = ny + nh;
    const int imax = nx + nh + 1, rmin = jsd, rmax = ny + nh + 1;  // last column / first and last row of the allocation
    // cube corners of this sub-domain whose damping-chain patch (WS_PATCH^2 corners) reaches into this tile
    const bool c_ll = W && S, c_hl = E && S, c_hh = E && N, c_lh = W && N;
    const int P = WS_PATCH;
    const bool patch_cols = ((c_ll || c_lh) && i0 - 3 <= P) || ((c_hl || c_hh) && i0 + FV3_WAVE - 4 >= nx + 2 - P);
    auto on_patch = [&](int ic_, int jc_) -> bool {
      const bool llo = ic_ >= 1 && ic_ <= P, lhi = ic_ >= nx + 2 - P && ic_ <= nx + 1;
      const bool mlo = jc_ >= 1 && jc_ <= P, mhi = jc_ >= ny + 2 - P && jc_ <= ny + 1;
      return (llo && mlo && c_ll) || (lhi && mlo && c_hl) || (lhi && mhi && c_hh) || (llo && mhi && c_lh);
    };
    // uniform bases
    const Real *const ub_ = fu + b, *const vb_ = fv + b, *const ucb = fuc + b, *const vcb = fvc + b, *const dgb = fdg + b;
    Real *const keb = fke + b, *const vdb = fvd + b, *const wkb_ = fwk + b, *const dnb = fdn + b;
    Real *const usb = side_seg > 0 ? fus + b : nullptr, *const vsb = side_seg > 0 ? fvs + b : nullptr;
    const Real *const cob = (const Real *)m_cosa + m2, *const rsb = (const Real *)m_rsina + m2, *const rdxb = (const Real *)m_rdx + m2, *const rdyb = (const Real *)m_rdy + m2;
    const Real *const dxb = (const Real *)m_dx + m2, *const dyb = (const Real *)m_dy + m2, *const rab = (const Real *)m_ra + m2;
    const Real *const dub = (const Real *)m_du + m2, *const dvb = (const Real *)m_dv + m2, *const racb = (const Real *)m_rac + m2;
    const unsigned rowB = (unsigned)sj32 * (unsigned)sizeof(Real);
    const Real dt5 = (Real)0.5 * dt, adt = fabs(dt);
    const Real d2k = d2_k[k], dd8 = dd8_k[k];

    // what a step consumes.  The 3-D rows are requested two steps ahead, the 2-D metric rows (shared by the sixteen levels of a tile through the XCD's L2) one
    // step ahead; three register sets each, rotated by the step's static index.
    struct Row3 {
      Real v, u;       // row r
      Real ucc, vcc;   // row jf = r - 2
      Real d;          // row r - 2 + nord (the damping chain's input)
    };
    struct RowM {
      Real dxr;              // row r
      Real dyc, rac;         // row r - 1 (the vorticity's cell row)
      Real co, rs, rx, ry;   // row jf
      Real du, dv, rc;       // row r - 2 + nord
    };
    Row3 R[3][FV3_LPT];
    RowM M[3][FV3_LPT];
    unsigned pcolB[FV3_LPT];
    bool own_e[FV3_LPT], own_c[FV3_LPT], own_w[FV3_LPT], own_us[FV3_LPT], own_vs[FV3_LPT];
    // KE (ke_stream; xtp_u with the reconstruction shared between neighbouring lanes as in fv3_tp4x.hip)
    Real w2[FV3_LPT], w3[FV3_LPT], w4[FV3_LPT], w5[FV3_LPT], al_v[FV3_LPT], uc_prev[FV3_LPT], ry_prev[FV3_LPT];
    PpmCell cv[FV3_LPT];
    Real u1[FV3_LPT], u2[FV3_LPT], a_prev[FV3_LPT];  // u of rows r-1, r-2; u * dx of row r-1
    Real s_vcc[FV3_LPT], s_ucs[FV3_LPT], s_co[FV3_LPT], s_rs[FV3_LPT], s_ry[FV3_LPT], s_rx[FV3_LPT], s_du[FV3_LPT], s_u[FV3_LPT];
    Real s_ubv[FV3_LPT], s_vbv[FV3_LPT], s_vfl[FV3_LPT], s_e[FV3_LPT], s_a0[FV3_LPT], s_a1[FV3_LPT], s_ra[FV3_LPT];
    Real s_al[FV3_LPT], s_bl[FV3_LPT], s_br[FV3_LPT], s_kev[FV3_LPT];
    bool s_sm[FV3_LPT];
    // corner interpolation (a2b_ord4_t)
    Real qa[FV3_LPT], qb[FV3_LPT], qc[FV3_LPT], qd[FV3_LPT], x0[FV3_LPT], x1[FV3_LPT], x2[FV3_LPT], x3[FV3_LPT], s_ly[FV3_LPT];
    // damping chain (divdamp_stream): iteration n works on row rho = (loaded row) - n; of its input it keeps rows rho (wb) and rho + 1 (wc) -- the i-neighbours of
    // row rho are wavefront shuffles at the point of use --, its own uc of row rho - 1, and the metric rows of the last four loaded rows
    Real wb[WS_NMAX][FV3_LPT], wc[WS_NMAX][FV3_LPT], ucp[WS_NMAX][FV3_LPT], newrow[WS_NMAX + 1][FV3_LPT], wa0[FV3_LPT];
    Real mdu[WS_NMAX + 1][FV3_LPT], mdv[WS_NMAX + 1][FV3_LPT], mra[WS_NMAX + 1][FV3_LPT];
    Real s_dpc[FV3_LPT], s_dn[FV3_LPT];

    auto clampr = [&](int r) -> int { return r < rmin ? rmin : (r > rmax ? rmax : r); };
    auto load3 = [&](int r, int l, auto gen_tag) -> Row3 {
      constexpr bool GEN = decltype(gen_tag)::value;
      const unsigned p0 = pcolB[l] + (unsigned)(GEN ? clampr(r) : r) * rowB, p2 = pcolB[l] + (unsigned)(GEN ? clampr(r - 2) : r - 2) * rowB;
      const unsigned pd = pcolB[l] + (unsigned)(GEN ? clampr(r - 2 + nord) : r - 2 + nord) * rowB;
      Row3 w;
      w.v = px_ld3(vb_, p0);
      w.u = px_ld3(ub_, p0);
      w.ucc = px_ld3(ucb, p2);
      w.vcc = px_ld3(vcb, p2);
      w.d = px_ld3(dgb, pd);
      return w;
    };
    auto loadm = [&](int r, int l, auto gen_tag) -> RowM {
      constexpr bool GEN = decltype(gen_tag)::value;
      const unsigned p0 = pcolB[l] + (unsigned)(GEN ? clampr(r) : r) * rowB, p1 = pcolB[l] + (unsigned)(GEN ? clampr(r - 1) : r - 1) * rowB;
      const unsigned p2 = pcolB[l] + (unsigned)(GEN ? clampr(r - 2) : r - 2) * rowB, pd = pcolB[l] + (unsigned)(GEN ? clampr(r - 2 + nord) : r - 2 + nord) * rowB;
      RowM w;
      w.dxr = px_ld(dxb, p0);
      w.dyc = px_ld(dyb, p1);
      w.rac = px_ld(rab, p1);
      w.co = px_ld(cob, p2);
      w.rs = px_ld(rsb, p2);
      w.rx = px_ld(rdxb, p2);
      w.ry = px_ld(rdyb, p2);
      w.du = px_ld(dub, pd);
      w.dv = px_ld(dvb, pd);
      w.rc = px_ld(racb, pd);
      return w;
    };

    // first step: the v window needs rows ja-3 ..; the chain's final row jf = ja needs its input from row ja - nord on, i.e. step ja + 2 - 2 nord
    int r_first = ja - 3;
    if (ja + 2 - 2 * nord < r_first) r_first = ja + 2 - 2 * nord;
    // last step: corner row jb at step jb + 2; the last segment also owns the wk rows up to jed (row jed at step jed + 1)
    int r_last = jb + 2;
    if (seg_last && jed + 1 > r_last) r_last = jed + 1;
    // side copies (see WindStage): the next row >= ja of the form 1 + m * side_seg, m >= 1
    int next_us = 1 << 30;
    if (side_seg > 0) {
      int m = (ja - 1 + side_seg - 1) / side_seg;
      if (m < 1) m = 1;
      next_us = 1 + m * side_seg;
    }

    FV3_LANES(blk_, lane, l) {
      const int i = i0 - 3 + lane, ic = i < imax ? i : imax;
      pcolB[l] = (unsigned)(go * sj32 + go + ic) * (unsigned)sizeof(Real);
      own_c[l] = i >= i0 && i < i0 + WS_OUT && i <= nx + 1;                    // a corner column of this strip
      own_e[l] = own_c[l] && i >= ia && i <= ib;                               // ... that takes the interior formulas
      own_w[l] = (i >= i0 && i < i0 + WS_OUT && i <= ied) || (strip_first && i >= isd && i < i0) || (strip_last && i >= i0 + WS_OUT && i <= ied);  // a wk column of this strip
      own_us[l] = own_c[l] && i <= nx;          // side copies: u on the compute columns of this strip ...
      own_vs[l] = i == i0 && bx > 0;            // ... v on the first column of every strip but the first
      w2[l] = w3[l] = w4[l] = w5[l] = al_v[l] = uc_prev[l] = ry_prev[l] = u1[l] = u2[l] = a_prev[l] = (Real)0;
      cv[l] = PpmCell{(Real)0, (Real)0, (Real)0, false};
      s_vcc[l] = s_ucs[l] = s_co[l] = s_rs[l] = s_ry[l] = s_rx[l] = s_du[l] = s_u[l] = (Real)0;
      s_ubv[l] = s_vbv[l] = s_vfl[l] = s_e[l] = s_a0[l] = s_a1[l] = s_ra[l] = s_al[l] = s_bl[l] = s_br[l] = s_kev[l] = s_ly[l] = (Real)0;
      s_sm[l] = false;
      qa[l] = qb[l] = qc[l] = qd[l] = x0[l] = x1[l] = x2[l] = x3[l] = (Real)0;
      s_dpc[l] = s_dn[l] = wa0[l] = (Real)0;
#pragma unroll
      for (int n = 0; n < WS_NMAX; ++n) wb[n][l] = wc[n][l] = ucp[n][l] = (Real)0;
#pragma unroll
      for (int n = 0; n <= WS_NMAX; ++n) mdu[n][l] = mdv[n][l] = mra[n][l] = newrow[n][l] = (Real)0;
      R[0][l] = load3(r_first, l, std::true_type{});
      R[1][l] = load3(r_first + 1, l, std::true_type{});
      R[2][l] = R[0][l];
      M[0][l] = loadm(r_first, l, std::true_type{});
      M[1][l] = M[2][l] = M[0][l];
    }

    // ---- step r, Q = (r - r_first) mod 3.  GEN: with the row-ownership tests, the load clamps and the cube-corner patch reads; the rows between run the form
    //      without them (every row test true, no patch corner in reach).
    auto step = [&](const int r_, auto q_tag, auto gen_tag) {
      constexpr int Q = decltype(q_tag)::value, Q1 = (Q + 1) % 3, Q2 = (Q + 2) % 3;
      constexpr bool GEN = decltype(gen_tag)::value;
      int r = r_;
      PX_OPAQUE_S(r);
      const int jf = r - 2;                                                                // the corner row of this step
      const bool row_e = !GEN || (jf >= ja && jf <= jb && jf >= jA && jf <= jB);           // ... takes the interior formulas: ke + damping stored
      const bool row_c = !GEN || (jf >= ja && jf <= jb);                                   // ... is this segment's
      const int jw = r - 1;                                                                // the wk cell row of this step
      const bool row_w = !GEN || (jw >= ja && jw <= jb && jw <= jed) || (seg_first && jw >= jsd && jw < ja) || (seg_last && jw > jb && jw <= jed);
      const bool prow = GEN && patch_cols && jf >= 1 && ((jf <= P && (c_ll || c_hl)) || (jf >= ny + 2 - P && jf <= ny + 1 && (c_hh || c_lh)));
      const bool side_now = side_seg > 0 && r >= ja && r <= jb && r >= 1 && r <= ny;  // this step's row r of u / v is one the side copies may want
      const bool side_u = side_now && r == next_us;
      if (side_seg > 0 && r >= next_us) next_us += side_seg;
      // ---- phase 1a (own lane): the requests; what the neighbouring lanes will read of this step's rows
      FV3_LANES(blk_, lane, l) {
        if (WS_PF3 == 2)
          R[Q2][l] = load3(r + 2, l, gen_tag);
        else
          R[Q1][l] = load3(r + 1, l, gen_tag);
        M[Q1][l] = loadm(r + 1, l, gen_tag);
        const Row3 cu = R[Q][l];
        const RowM cm = M[Q][l];
        if (side_now) {  // (wave-uniform; rows of this segment only: every row belongs to one segment)
          const unsigned ps = pcolB[l] + (unsigned)r * rowB;
          if (side_u && own_us[l]) *fv3_at(usb, ps) = cu.u;
          if (own_vs[l]) *fv3_at(vsb, ps) = cu.v;
        }
        s_vcc[l] = cu.vcc;
        s_ucs[l] = uc_prev[l] + cu.ucc;
        uc_prev[l] = px_move(cu.ucc);
        s_co[l] = cm.co;
        s_rs[l] = cm.rs;
        s_ry[l] = cm.ry;
        s_rx[l] = cm.rx;
        s_u[l] = u2[l];  // u of row jf = r - 2
        // vorticity of cell (i, r-1): rarea * (u dx - (u dx)[j+1] - v dy + (v dy)[i+1])
        const Real a1 = cu.u * cm.dxr;
        s_a0[l] = a_prev[l];
        s_a1[l] = a1;
        a_prev[l] = a1;
        u2[l] = u1[l];
        u1[l] = px_move(cu.u);
        // ytp_v: v window rows r-3 .. r
        w2[l] = w3[l];
        w3[l] = w4[l];
        w4[l] = w5[l];
        w5[l] = px_move(cu.v);
        s_e[l] = w4[l] * cm.dyc;  // (w4 = v of row r - 1)
        s_ra[l] = cm.rac;
        // damping chain: the metric delay lines, the loaded row
#pragma unroll
        for (int n = WS_NMAX; n >= 1; --n) {
          mdu[n][l] = mdu[n - 1][l];
          mdv[n][l] = mdv[n - 1][l];
          mra[n][l] = mra[n - 1][l];
        }
        mdu[0][l] = px_move(cm.du);  // (moved out of the load's register: a value that outlives its step would be copied at the back edge while a younger load is in flight)
        mdv[0][l] = px_move(cm.dv);
        mra[0][l] = px_move(cm.rc);
        newrow[0][l] = px_move(cu.d);
      }
      // ---- phase 1b: ytp_v at face jf (al(r-1), cell r-2, the face between cells r-3 and r-2), the contravariant corner winds, the edge value at the low face of
      //      the lane's u cell; the vorticity of the cell row, its window, the y-interpolated corner row
      FV3_LANES(blk_, lane, l) {
        const Real al_new = PPM_P1 * (w3[l] + w4[l]) + PPM_P2 * (w2[l] + w5[l]);
        const PpmCell co = ppm_cell(al_v[l], al_new, w3[l], hord);
        al_v[l] = al_new;
        const Real vcs = FV3_LANE_SHR(1, s_vcc, l, lane) + s_vcc[l];  // vc(i-1, jf) + vc(i, jf)
        const Real vbv = dt5 * (vcs - s_ucs[l] * s_co[l]) * s_rs[l];
        const Real ubv = dt5 * (s_ucs[l] - vcs * s_co[l]) * s_rs[l];
        s_vfl[l] = ppm_face_cfl(cv[l], co, vbv, ry_prev[l], s_ry[l]);
        cv[l] = co;
        ry_prev[l] = s_ry[l];
        s_vbv[l] = vbv;
        s_ubv[l] = ubv;
        s_al[l] = PPM_P1 * (FV3_LANE_SHR(1, s_u, l, lane) + s_u[l]) + PPM_P2 * (FV3_LANE_SHR(2, s_u, l, lane) + FV3_LANE_SHL(1, s_u, l, lane));
        const Real e1 = FV3_LANE_SHL(1, s_e, l, lane);
        const Real wkv = s_ra[l] * (s_a0[l] - s_a1[l] - s_e[l] + e1);
        if (row_w && own_w[l]) *fv3_at(wkb_, pcolB[l] + (unsigned)jw * rowB) = wkv;
        qa[l] = qb[l];
        qb[l] = qc[l];
        qc[l] = qd[l];
        qd[l] = wkv;
        s_ly[l] = A2B_B2 * (qa[l] + qd[l]) + A2B_B1 * (qb[l] + qc[l]);  // corner row jf
      }
      PX_FENCE();
      // ---- phase 2: the lane's u cell, then xtp_u at the lane's corner (the face between the previous lane's cell and the lane's own) and the kinetic energy
      FV3_LANES(blk_, lane, l) {
        const PpmCell ci = ppm_cell(s_al[l], FV3_LANE_SHL(1, s_al, l, lane), s_u[l], hord);
        s_bl[l] = ci.bl;
        s_br[l] = ci.br;
        s_sm[l] = ci.sm;
      }
      FV3_LANES(blk_, lane, l) {
        const PpmCell cmn{FV3_LANE_SHR(1, s_bl, l, lane), FV3_LANE_SHR(1, s_br, l, lane), FV3_LANE_SHR(1, s_u, l, lane), FV3_LANE_SHR1_FLAG(s_sm, l, lane)};
        const PpmCell c0{s_bl[l], s_br[l], s_u[l], s_sm[l]};
        const Real ufl = ppm_face_cfl(cmn, c0, s_ubv[l], FV3_LANE_SHR(1, s_rx, l, lane), s_rx[l]);
        s_kev[l] = (Real)0.5 * (s_vbv[l] * s_vfl[l] + s_ubv[l] * ufl);
      }
      PX_FENCE();
      // ---- phase 3: the chain's iterations
      FV3_LANES(blk_, lane, l) {  // the un-iterated divergence of the chain's final row: nord rows behind the loaded one
        s_dpc[l] = nord == 1 ? wc[0][l] : nord == 2 ? wb[0][l] : wa0[l];
        wa0[l] = wb[0][l];
      }
#pragma unroll
      for (int n = 1; n <= WS_NMAX; ++n) {
        if (n <= nord) {
          FV3_LANES(blk_, lane, l) {  // the row iteration n-1 produced at this step enters the window of iteration n
            wb[n - 1][l] = wc[n - 1][l];
            wc[n - 1][l] = newrow[n - 1][l];
          }
          FV3_LANES(blk_, lane, l) {  // iteration n on row rho (window row wb), metrics of that row = delay n
            const Real bC = wb[n - 1][l], bL = FV3_LANE_SHR(1, wb[n - 1], l, lane), bR = FV3_LANE_SHL(1, wb[n - 1], l, lane);
            const Real ucc = (wc[n - 1][l] - bC) * mdv[n][l];                       // uc(i, rho)
            const Real vcm = (bC - bL) * FV3_LANE_SHR(1, mdu[n], l, lane);          // vc(i-1, rho)
            const Real vcc = (bR - bC) * mdu[n][l];                                 // vc(i, rho)
            const Real dn_ = (ucp[n - 1][l] - ucc + vcm - vcc) * mra[n][l];
            ucp[n - 1][l] = ucc;
            newrow[n][l] = dn_;
            if (n == nord) s_dn[l] = dn_;  // (not newrow[nord]: a run-time index sends the array through scratch memory -- and its wait is vmcnt(0))
          }
        }
      }
      PX_FENCE();
      // ---- phase 4: the corner interpolation and the damping of corner (i, jf)
      FV3_LANES(blk_, lane, l) {
        x0[l] = x1[l];
        x1[l] = x2[l];
        x2[l] = x3[l];
        x3[l] = A2B_B2 * (FV3_LANE_SHR(2, qd, l, lane) + FV3_LANE_SHL(1, qd, l, lane)) + A2B_B1 * (FV3_LANE_SHR(1, qd, l, lane) + qd[l]);
      }
      FV3_LANES(blk_, lane, l) {
        const Real qxx = A2B_A2 * (x0[l] + x3[l]) + A2B_A1 * (x1[l] + x2[l]);
        const Real qyy = A2B_A2 * (FV3_LANE_SHR(2, s_ly, l, lane) + FV3_LANE_SHL(1, s_ly, l, lane)) + A2B_A1 * (FV3_LANE_SHR(1, s_ly, l, lane) + s_ly[l]);
        const Real wkbv = (Real)0.5 * (qxx + qyy);
        Real dn_ = s_dn[l];
        bool onp = false;
        if (GEN && prow) {  // a corner of the chain's cube-corner patch: the staged chain's value (rare: waited for inside the branch)
          int lane_o = lane;
          FV3_LAUNDER(lane_o);
          onp = on_patch(i0 - 3 + lane_o, jf);
          if (onp && own_c[l]) dn_ = px_ld(dnb, pcolB[l] + (unsigned)jf * rowB);
          FV3_LANDED(dn_);
        }
        const unsigned p = pcolB[l] + (unsigned)jf * rowB;
        if (row_e && own_e[l]) {
          const Real dpc = s_dpc[l];
          Real vo = (Real)0;
          if (dddmp >= (Real)1.0e-5) vo = adt * sqrt(dpc * dpc + wkbv * wkbv);
          const Real damp2 = da_min_c * fv3_max(d2k, fv3_min((Real)0.20, dddmp * vo));
          const Real vd = damp2 * dpc + dd8 * dn_;
          FV3_ST_NT(*fv3_at(vdb, p), vd);
          *fv3_at(keb, p) = s_kev[l] + vd;
          if (store_dn) *fv3_at(dnb, p) = dn_;
        } else if (row_c && own_c[l] && !onp) {
          *fv3_at(dnb, p) = dn_;  // a tile-edge corner of this tile: the per-point launch that follows needs the iterated divergence
        }
      }
      PX_FENCE();
    };

    // ---- the march: general triples while the windows fill / next to the S tile edge and corner patch, branch-free triples, general triples to the end
    int r_lo = (ja > jA ? ja : jA) + 2;                     // first step whose corner row takes the interior formulas ...
    if (r_lo < r_first + 6) r_lo = r_first + 6;             // ... with every window filled from real rows
    if (patch_cols && (c_ll || c_hl) && r_lo < P + 3) r_lo = P + 3;  // ... and no S corner patch in reach (jf > P)
    int r_hi = jb + 1 < jB + 2 ? jb + 1 : jB + 2;           // last step that owns its corner row AND its wk row and takes the interior formulas ...
    if (r_hi > rmax - 2) r_hi = rmax - 2;                   // ... and requests no row past the allocation
    if (r_hi > rmax - nord - 1) r_hi = rmax - nord - 1;
    if (patch_cols && (c_hh || c_lh) && r_hi > ny + 3 - P) r_hi = ny + 3 - P;  // ... and meets no N corner patch (jf < ny + 2 - P)
    int r = r_first;
#pragma clang loop unroll(disable)
    for (int part = 0; part < 2; ++part) {  // (one copy of the general triple in the code: head and tail are two trips of this loop)
      const int stop = part == 0 ? r_lo - 1 : r_last;
#pragma clang loop unroll(disable)
      for (; r <= stop; r += 3) {
        step(r, std::integral_constant<int, 0>{}, std::true_type{});
        step(r + 1, std::integral_constant<int, 1>{}, std::true_type{});
        step(r + 2, std::integral_constant<int, 2>{}, std::true_type{});
      }
      if (part == 0) {
#pragma clang loop unroll(disable)
        for (; r + 2 <= r_hi; r += 3) {
          step(r, std::integral_constant<int, 0>{}, std::false_type{});
          step(r + 1, std::integral_constant<int, 1>{}, std::false_type{});
          step(r + 2, std::integral_constant<int, 2>{}, std::false_type{});
        }
      }
    }
    (void)gp;
  });
}

}  // namespace

void wind_stage_march(fv3_ctx *c, fv3_stream_t s, const WindStage &a) {
  static const bool hc_off = getenv("FV3_HORD_CONST") && getenv("FV3_HORD_CONST")[0] == '0';
  if (a.hord == 6 && !hc_off)
    wind_stage_march_t<6>(c, s, a);
  else
    wind_stage_march_t<0>(c, s, a);
}
