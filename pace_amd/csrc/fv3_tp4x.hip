// fv3_tp4x.hip -- d_sw's two-tracer scalar marches (delp + w, q_con + pt), round-5 form.  Same operator, same expressions in the same
// order as dsw_scalars_t<ROLE, Q4_INTERIOR, false, true> of fv3_tp4.hip (bitwise equal: FV3_DSW_MARCH=old is the A/B switch, and
// tests/test_parity.py::test_pair_march_is_bitwise_the_round4_march compares them); CPU twin: oracle/fv3_oracle/d_sw.py (the four fv_tp_2d
// calls and the divisions by the new air mass of d_sw_levels).  [SURVEY A.3.2 - A.3.4, A.4]
//
// Why a second form.  Round 4 measured the round-4 march from inside: 57 800 VALU instructions per wave of which 33 500 were
// fp64 arithmetic -- the rest were loop-carried register copies of the rolled march (~6 000), 64-bit address arithmetic (~4 000), reloads
// of spilled SGPRs (~4 000: 106 live scalars, most of them the state of rare paths) and DPP moves (7 300).  This file is the march
// with those removed by construction:
//   * NO RARE PATH IN THE BRANCH-FREE ROW STEP.  What is rare -- the one-sided PPM formulas on the three rows either side of a S / N tile edge, the rows a
//     segment does not own while its windows fill and drain, and, on the (strip, segment) tiles that touch a cube corner of their sub-domain (1 tile of 28
//     at C768 layout 2 x 2), the corner-halo remaps and the faces whose del-n fluxes come from the staged chain's 8 x 8 patches -- lives in a GENERAL form of
//     the step (template argument GEN) that runs the first six (fifteen next to a S corner patch) and the last few rows of a segment; the rows between run
//     the form without a single branch, ownership test or clamp.  The W / E one-sided formulas are evaluated in the lanes of a tile-edge strip (px_al_edge).
//     The march therefore serves EVERY tile of the levels that run their chains inside it (the round-4 kernels of fv3_tp4.hip only serve the sponge levels,
//     other PPM orders and the A/B switch).
//   * STATIC ROTATION.  The march is unrolled by three: the three register sets of the rows in flight, the optional inputs and the
//     del-n metric rows fetched one step ahead rotate by the step's static index Q instead of being copied, and the own-lane LDS ring
//     of the delay lines has three slots addressed by Q too (immediate offsets: no address arithmetic).
//   * SCALAR-BASE ADDRESSING.  Every access is (uniform field base) + (32-bit byte offset of the lane's column and row); the six row
//     offsets of a step are one vector add each.
//   * ONE RECIPROCAL PER DENOMINATOR.  The two tracers of a wave divide by the same cross-direction areas (and q_con / pt by the same
//     new air mass): the refined reciprocal (rcp + two Newton steps, the part of the fp64 division sequence that depends on the
//     denominator only) is formed once and each quotient takes three more instructions -- operation for operation what fv3_div does per
//     quotient, hence the same bits (fv3_math.h).
//   * the PPM order is the constant 6 (the reference configurations; any other order takes the round-4 kernel).
#include <type_traits>

#include "fv3_ops.h"
#include "fv3_math.h"
#define FV3_MARCH_ST(lhs, val) FV3_ST_NT(lhs, val)
#include "fv3_ppm.h"
#include "fv3_march.h"

namespace {

#define PX_OUT 58
#define PX_ORD 6
enum { PX_AIR = 1, PX_TRC = 2, PX_BOTH = 3 };
#ifndef PX_WPE
#define PX_WPE (sizeof(Real) == 4 ? 3 : 2)  // (fp32: 168 registers)
#endif

// MODE PX_AIR / PX_TRC: one role per launch (the round-5 form).  PX_BOTH (round 6, device only): workgroups of TWO waves on the same (strip, segment, level) tile --
// wave 0 runs the delp + w role, wave 1 the q_con + pt role one row step behind it -- and what the second role used to read back from memory is handed over
// through LDS: the air-mass fluxes of the sub-step (written by one march and read by the other: 9.4 GB at C768), the old air mass, and the five rows both
// roles read (Courant numbers, area fluxes, cell areas: 11.7 GB).  One workgroup barrier per row step (`s_waitcnt lgkmcnt(0); s_barrier`: the prefetched rows
// stay in flight); the hand-over block has three slots addressed by the step's static index like the rings.  Same expressions on the same values: the same bits.
template <int MODE>
void pair_march_t(fv3_ctx *c, fv3_stream_t s, const DswScalars &a, int k_lo, int k_hi) {
  const Geo g = c->g;
  const int nk = k_hi - k_lo + 1;
  if (nk <= 0) return;
  const int nL = g.nx, nM = g.ny, nh = g.nh, go = g.o, sj32 = g.sj32;
  // (launch geometry of dsw_scalars_t<ROLE, Q4_INTERIOR>: the two kernels split the same tiles between them)
  const int nstrip = (nL + 1 + PX_OUT - 1) / PX_OUT;
  const int seg = fv3_pick_seg((long)nstrip * ((nM + 63) / 64) * g.nsub * nk, PX_WPE);
  const int nseg = (nM + seg - 1) / seg;
  static const int kb_env = getenv("FV3_Q4_KB") ? atoi(getenv("FV3_Q4_KB")) : FV3_Q4_KB_DEFAULT;
  const int KB = kb_env > 0 ? (kb_env < nk ? kb_env : nk) : 0;
  const int nblk = KB ? (nk + KB - 1) / KB : 0;
  const long st = g.st, sk = g.sk, st2 = g.st2;
  const unsigned char *gflags = c->g_dev->flags;
  const MPtr garea = g.area, grarea = g.rarea, gd6L = g.del6_v, gd6M = g.del6_u, gdya = g.dya, gdxa = g.dxa;
  const Geo *gp = c->g_dev;
  const Real *const crL = a.crx, *const crM = a.cry, *const afL = a.xfx, *const afM = a.yfx;
  Real *const accL = a.mfx, *const accM = a.mfy;          // AIR: accumulated air-mass fluxes (read + written)
  const bool acc_first_ = a.acc_first && a.zeros;
  const Real *const zerosb = a.zeros;
  Real *const flL = a.fx, *const flM = a.fy;              // air-mass fluxes of the sub-step: AIR writes, TRC reads (uncoupled form)
  const Real *const oldm = a.delp;                        // TRC: the old air mass
  Real *const heat = a.heat;
  // per-level coefficients: d2 of iteration 0 = c0 * q (delp, w); the mass-weighted damping of q_con / pt
  const Deln dn_vt = a.dn_vt, dn_t = a.dn_t, dn_w = a.dn_w;
  const Real *ke_bg_k = g.ke_bg;
  const Real adt = fabs(a.dt);
  constexpr int NRING = 9;  // ring variables: del-n metric rows (M faces, L faces, 1 / area), Courant number, area flux, area, inner L flux x 2, cell width (tile-edge strips)
  enum { RG_DU = 0, RG_DV = 1, RG_RA = 2, RG_CX = 3, RG_XV = 4, RG_AR = 5, RG_FI = 6, RG_MX = 8 };
  const size_t smem = sizeof(Real) * (size_t)NRING * 3 * FV3_WAVE;
  // hand-over block of the coupled form: [slot 3][variable 8][lane 64]
  enum { XC_CX = 0, XC_XV = 1, XC_AR = 2, XC_CY = 3, XC_YV = 4, XC_FX = 5, XC_FY = 6, XC_OM = 7, NXCH = 8 };
  auto body = [=] FV3_HD(auto role_tag, auto cpl_tag, const Blk &blk_, char *smem_, Real *xch) {
    constexpr bool AIR = decltype(role_tag)::value == PX_AIR;
    constexpr bool CPL = decltype(cpl_tag)::value;      // coupled: this wave shares its tile with the wave of the other role
    constexpr bool XIN = CPL && !AIR;                   // ... and takes the shared rows / the air-mass fluxes / the old air mass from the hand-over block
    (void)xch;
    // the staged chain's damping fluxes on the cube-corner patches, per tracer slot (L faces, M faces)
    const Real *const pL0 = AIR ? a.dpx : a.dqx, *const pL1 = AIR ? a.dwx : a.dtx, *const pM0 = AIR ? a.dpy : a.dqy, *const pM1 = AIR ? a.dwy : a.dty;
    // what a wave reads / writes (the role folds the unused ones away)
    const Real *const q0f = AIR ? a.delp : a.q_con, *const q1f = AIR ? a.w : a.pt;
    Real *const out0 = AIR ? a.o_delp : a.o_q_con, *const out1 = AIR ? a.o_w : a.o_pt;
    const bool acc_first = AIR && acc_first_;
    auto XC = [&](int var, int slot) -> Real * { return xch + (slot * NXCH + var) * FV3_WAVE; };
    int t, k, bx, by;
    if (KB) {
      t = blk_.bz / nblk;
      const int kk = (blk_.bz - t * nblk) * KB + blk_.bx;
      if (kk >= nk) return;
      k = k_lo + kk;
      by = blk_.by / nstrip;
      bx = blk_.by - by * nstrip;
    } else {
      t = blk_.bz / nk;
      k = k_lo + (blk_.bz - t * nk);
      bx = blk_.bx;
      by = blk_.by;
    }
    const int fl = gflags[t];
    const int l0 = 1 + bx * PX_OUT;
    const int ca = 1 + by * seg, fa = ca;
    const int fb = by == nseg - 1 ? nM + 1 : fa + seg - 1;
    const int cb = fb < nM ? fb : nM;
    const int Led = nL + nh, Msd = 1 - nh, Med = nM + nh, npM = nM + 1;
    const int r_end = fb + 3 < Med ? fb + 3 : Med;
    const bool Mlo = fl & FV3_S, Mhi = fl & FV3_N;
    // xe: this strip has W / E one-sided formulas among its L faces (evaluated in the lanes: px_al_edge)
    const bool Llo = (fl & FV3_W) && l0 <= 3, Lhi = (fl & FV3_E) && l0 + PX_OUT + 1 >= nL;
    const bool xe = Llo || Lhi;
    // Cube corners (general steps only): corner-halo remaps of the rows outside 1 .. nM, the staged chain's fluxes on the corner patches
    const bool halo_cols = l0 - 3 < 1 || l0 + FV3_WAVE - 4 > nL;
    const bool c_ll = (fl & (FV3_W | FV3_S)) == (FV3_W | FV3_S), c_hl = (fl & (FV3_E | FV3_S)) == (FV3_E | FV3_S);
    const bool c_hh = (fl & (FV3_E | FV3_N)) == (FV3_E | FV3_N), c_lh = (fl & (FV3_W | FV3_N)) == (FV3_W | FV3_N);
    const bool pz = ((c_ll || c_lh) && l0 - 3 <= FV3_D6_PATCH) || ((c_hl || c_hh) && l0 + FV3_WAVE - 4 >= nL + 2 - FV3_D6_PATCH);
    auto on_patch = [&](int lc, int m) -> bool {
      const bool llo = lc >= 1 && lc <= FV3_D6_PATCH, lhi = lc >= nL + 2 - FV3_D6_PATCH && lc >= 1 && lc <= nL + 1;
      const bool mlo = m >= 1 && m <= FV3_D6_PATCH, mhi = m >= nM + 2 - FV3_D6_PATCH && m >= 1 && m <= nM + 1;
      return (llo && mlo && c_ll) || (lhi && mlo && c_hl) || (lhi && mhi && c_hh) || (llo && mhi && c_lh);
    };
    const long b = t * st + k * sk, m2 = t * st2;
    // uniform bases
    const Real *const q0b = q0f + b, *const q1b = q1f + b, *const crLb = crL + b, *const crMb = crM + b, *const afLb = afL + b, *const afMb = afM + b;
    const Real *const dxab = (const Real *)gdxa + m2;
    const Real *const areab = (const Real *)garea + m2, *const rab = (const Real *)grarea + m2, *const d6Lb = (const Real *)gd6L + m2, *const d6Mb = (const Real *)gd6M + m2;
    Real *const accLb = accL + b, *const accMb = accM + b, *const flLb = flL + b, *const flMb = flM + b;
    // (first sub-step of a call inside the sequencer: the accumulators hold nothing -- the "old value" is read from one plane of zeros, which stays in L2)
    const Real *const accLb_ld = acc_first ? zerosb : accLb, *const accMb_ld = acc_first ? zerosb : accMb;
    const Real *const oldmb = oldm + b;
    Real *const out0b = out0 + b, *const out1b = out1 + b, *const heatb = heat + b;
    const Real damp_vt = deln_damp(dn_vt, k), damp_t = deln_damp(dn_t, k);
    const Real c0 = AIR ? deln_damp(dn_vt, k) : (Real)1, c1 = AIR ? deln_damp(dn_w, k) : (Real)1;
    const Real dd8 = ke_bg_k[k] * adt;
    const unsigned rowB = (unsigned)sj32 * (unsigned)sizeof(Real);
    Real *const ring = (Real *)smem_;
    auto RG = [&](int var, int slot) -> Real * { return ring + (var * 3 + slot) * FV3_WAVE; };

    // ---- per-lane state
    struct Row {
      Real q0, q1, cx, xv, ar, cy, yv;
    };
    struct Met {  // del-n metric terms of a row, requested one step ahead
      Real du, dv, ra;
      Real mx;  // tile-edge strips: the cell width dxa of the lane's column
    };
    Row R[3][FV3_LPT];
    // inputs a step consumes late, requested one step ahead.  AIR: accumulated L / M fluxes;  TRC: air-mass L / M fluxes, old air mass of (lc, r-2)
    Real Ox[3][FV3_LPT], Oy[3][FV3_LPT], Om[3][FV3_LPT];
    Met MN[3][FV3_LPT];
    unsigned pcolB[FV3_LPT];  // byte offset of (lc, M coordinate 0) inside a plane
    bool own_x[FV3_LPT], own_y[FV3_LPT];
    // PPM windows / cells of q (rows r-3 .. r) and of the L-advected q (likewise), per tracer
    Real w2[2][FV3_LPT], w3[2][FV3_LPT], w4[2][FV3_LPT], al_q[2][FV3_LPT];
    Real v2[2][FV3_LPT], v3[2][FV3_LPT], v4[2][FV3_LPT], al_v[2][FV3_LPT];
    PpmCell cq[2][FV3_LPT], cv[2][FV3_LPT];
    Real p_prev[2][FV3_LPT], y_prev[FV3_LPT];
    Real fyin[2][FV3_LPT], px[2][FV3_LPT], fxk[2][FV3_LPT], fyp[2][FV3_LPT];
    Real sqx[2][FV3_LPT], sqi[2][FV3_LPT], smb[FV3_LPT], sxv[FV3_LPT];
    Real mbk[FV3_LPT], fyp_air[FV3_LPT];
    // del-n chains (see dsw_scalars_t)
    Real sd0[2][FV3_LPT], sd1[2][FV3_LPT], sd2[2][FV3_LPT], gx0[2][FV3_LPT], gx1[2][FV3_LPT], gy0[2][FV3_LPT], gy1[2][FV3_LPT];
    Real dxd[2][FV3_LPT], dyf[2][FV3_LPT], zyp[FV3_LPT], zxo[FV3_LPT];
    // values handed from phase to phase inside a step
    Real s_al[2][2][FV3_LPT], s_bl[2][2][FV3_LPT], s_br[2][2][FV3_LPT];  // L sweeps (inner, outer) x tracers: edge value at the low face of the lane's cell, the cell's bl / br ...
    bool s_sm[2][2][FV3_LPT];                                              // ... and limiter flag
    int akind[FV3_LPT];                   // tile-edge strips: the edge-value form of the lane's cell (px_al_edge)
    Real h_mx0[FV3_LPT], h_mx3[FV3_LPT];  // ... and dxa of the lane's column on rows r / r-3
    Real h_era[FV3_LPT], h_ody[2][FV3_LPT], h_q5[2][FV3_LPT], h_mc[FV3_LPT];
    Real h_zx0[FV3_LPT], h_zy0[FV3_LPT], h_zy1[FV3_LPT];

    // loads.  GEN: rows clamped like the round-4 kernel clamps them while the windows fill / past the last row
    auto load_row = [&](int r, int l, auto gen_tag) -> Row {
      constexpr bool GEN = decltype(gen_tag)::value;
      const int rf = GEN ? (r - 2 < Msd ? Msd : r - 2) : r - 2;
      const unsigned p0 = pcolB[l] + (unsigned)r * rowB, pf = pcolB[l] + (unsigned)rf * rowB;
      Row w;
      w.q0 = px_ld3(q0b, p0);
      w.q1 = px_ld3(q1b, p0);
      if constexpr (XIN) {  // (the other role's wave hands these over: Row's slots are filled from the hand-over block at the top of the step)
        w.cx = w.xv = w.ar = w.cy = w.yv = (Real)0;
        (void)pf;
      } else {
        w.cx = px_ld3(crLb, p0);
        w.xv = px_ld3(afLb, p0);
        w.ar = px_ld(areab, p0);
        w.cy = px_ld3(crMb, pf);
        w.yv = px_ld3(afMb, pf);
      }
      return w;
    };
    auto load_opt = [&](int q, int r, int l, auto gen_tag) {  // what step r consumes: row r-3, face r-2 (into set q)
      constexpr bool GEN = decltype(gen_tag)::value;
      const int r3 = GEN ? (r - 3 < Msd ? Msd : r - 3) : r - 3, rf = GEN ? (r - 2 < Msd ? Msd : r - 2) : r - 2;
      const unsigned p3 = pcolB[l] + (unsigned)r3 * rowB, pf = pcolB[l] + (unsigned)rf * rowB;
      if constexpr (AIR) {
        Ox[q][l] = px_ld3(accLb_ld, p3);
        Oy[q][l] = px_ld3(accMb_ld, pf);
      } else if constexpr (!XIN) {
        Ox[q][l] = px_ld3(flLb, p3);
        Oy[q][l] = px_ld3(flMb, pf);
        Om[q][l] = px_ld3(oldmb, pf);
      } else {
        (void)p3;
        (void)pf;
      }
    };
    auto load_met = [&](int r, int l) -> Met {
      const unsigned pm = pcolB[l] + (unsigned)r * rowB;
      Met m;
      m.du = px_ld(d6Mb, pm);
      m.dv = px_ld(d6Lb, pm);
      m.ra = px_ld(rab, pm);
      m.mx = (Real)1;
      if (xe) m.mx = px_ld(dxab, pm);
      return m;
    };

    const int r0 = ca - 3;
    FV3_LANES(blk_, lane, l) {
      const int lc = l0 - 3 + lane, lcc = lc < Led ? lc : Led;
      pcolB[l] = (unsigned)(go * sj32 + go + lcc) * (unsigned)sizeof(Real);
      own_x[l] = lc >= l0 && lc < l0 + PX_OUT && lc <= nL + 1;
      own_y[l] = lc >= l0 && lc < l0 + PX_OUT && lc <= nL;
      akind[l] = 0;
      if (Llo && lc >= 0 && lc <= 2) akind[l] = 1 + lc;
      if (Lhi && lc >= nL && lc <= nL + 2) akind[l] = 1 + (lc - nL);
      h_mx0[l] = h_mx3[l] = (Real)1;
      y_prev[l] = smb[l] = sxv[l] = mbk[l] = fyp_air[l] = zyp[l] = zxo[l] = (Real)0;
      h_era[l] = h_mc[l] = h_zx0[l] = h_zy0[l] = h_zy1[l] = (Real)0;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        w2[n][l] = w3[n][l] = w4[n][l] = al_q[n][l] = v2[n][l] = v3[n][l] = v4[n][l] = al_v[n][l] = (Real)0;
        cq[n][l] = cv[n][l] = PpmCell{(Real)0, (Real)0, (Real)0, false};
        p_prev[n][l] = fyin[n][l] = px[n][l] = fxk[n][l] = fyp[n][l] = sqx[n][l] = sqi[n][l] = (Real)0;
        sd0[n][l] = sd1[n][l] = sd2[n][l] = gx0[n][l] = gx1[n][l] = gy0[n][l] = gy1[n][l] = dxd[n][l] = dyf[n][l] = (Real)0;
        h_ody[n][l] = h_q5[n][l] = (Real)0;
        for (int w = 0; w < 2; ++w) {
          s_al[w][n][l] = s_bl[w][n][l] = s_br[w][n][l] = (Real)0;
          s_sm[w][n][l] = false;
        }
      }
      for (int v = 0; v < NRING; ++v)
        for (int q = 0; q < 3; ++q) RG(v, q)[lane] = (v == RG_AR || v == RG_MX) ? (Real)1 : (Real)0;  // (warm-up steps: outputs masked, keep the divisions finite)
      // step r0 (Q = 0) consumes R[0] = row r0, O[0], MN[0]; row r0 + 1 is in flight in R[1]
      MN[0][l] = load_met(r0, l);
#pragma unroll
      for (int q = 0; q < 3; ++q) Ox[q][l] = Oy[q][l] = Om[q][l] = (Real)0;
      load_opt(0, r0, l, std::true_type{});
      R[0][l] = load_row(r0, l, std::true_type{});
      R[1][l] = load_row(r0 + 1 < r_end ? r0 + 1 : r_end, l, std::true_type{});
      MN[1][l] = MN[2][l] = MN[0][l];
      R[2][l] = R[0][l];
#if PX_ABL == 2
      R[2][l] = load_row(r0 + 2 < r_end ? r0 + 2 : r_end, l, std::true_type{});
      load_opt(1, r0 + 1 < r_end ? r0 + 1 : r_end, l, std::true_type{});
      load_opt(2, r0 + 2 < r_end ? r0 + 2 : r_end, l, std::true_type{});
      MN[1][l] = load_met(r0 + 1 < r_end ? r0 + 1 : r_end, l);
      MN[2][l] = load_met(r0 + 2 < r_end ? r0 + 2 : r_end, l);
#endif
    }

    // ---- step r, Q = (r - r0) mod 3.  GEN: with the row-ownership tests, the load clamps and the S / N tile-edge formulas
    auto step = [&](const int r_, auto q_tag, auto gen_tag) {
      constexpr int Q = decltype(q_tag)::value, Q1 = (Q + 1) % 3, Q2 = (Q + 2) % 3;
      constexpr bool GEN = decltype(gen_tag)::value;
      // (the row index is made opaque: seen as an induction variable, every access stream gets a 64-bit vector pointer of its own that is
      //  bumped per step -- 2 registers and a 64-bit add per stream -- instead of (uniform base) + (32-bit offset))
      int r = r_;
      PX_OPAQUE_S(r);
      // ring slots: rows r and r-3 share slot Q (r-3 is read before r is written), r-1 -> Q2, r-2 -> Q1
      const int sy = r - 1;  // cell whose low edge value the M windows complete at this step
      const bool m_edge = GEN && ((Mlo && sy >= 0 && sy <= 2) || (Mhi && sy >= npM - 1 && sy <= npM + 1));
      const int jr = r - 3, jf = r - 2;
      const bool fx_row = !GEN || (jr >= ca && jr <= cb), fy_row = !GEN || (jf >= fa && jf <= fb);
      if constexpr (CPL) {
        if (GEN && XIN && r_ < r0) {  // (the lagging wave has no step in the first row step of its group)
          blk_.group_sync();
          return;
        }
        FV3_LANES(blk_, lane, l) {
          if constexpr (XIN) {  // what the other role's wave left for this step: its slot Q
            R[Q][l].cx = XC(XC_CX, Q)[lane];
            R[Q][l].xv = XC(XC_XV, Q)[lane];
            R[Q][l].ar = XC(XC_AR, Q)[lane];
            R[Q][l].cy = XC(XC_CY, Q)[lane];
            R[Q][l].yv = XC(XC_YV, Q)[lane];
            Ox[Q][l] = XC(XC_FX, Q)[lane];
            Oy[Q][l] = XC(XC_FY, Q)[lane];
            Om[Q][l] = XC(XC_OM, Q)[lane];
          } else {              // the five rows both roles read
            XC(XC_CX, Q)[lane] = R[Q][l].cx;
            XC(XC_XV, Q)[lane] = R[Q][l].xv;
            XC(XC_AR, Q)[lane] = R[Q][l].ar;
            XC(XC_CY, Q)[lane] = R[Q][l].cy;
            XC(XC_YV, Q)[lane] = R[Q][l].yv;
          }
        }
      }
#if PX_ABL == 1
      FV3_LANES(blk_, lane, l) {
        const Met mc_ = MN[Q][l];
        const Row cu = R[Q][l];
        const Real ox = Ox[Q][l], oy = Oy[Q][l], om = AIR ? (Real)0 : Om[Q][l];
        {
          const int r1 = GEN ? (r + 1 < r_end ? r + 1 : r_end) : r + 1, rn = GEN ? (r + 2 < r_end ? r + 2 : r_end) : r + 2;
          load_opt(Q1, r1, l, gen_tag);
          MN[Q1][l] = load_met(r1, l);
          R[Q2][l] = load_row(rn, l, gen_tag);
        }
        const Real sum = ((cu.q0 + cu.q1) + (cu.cx + cu.xv)) + ((cu.ar + cu.cy) + (cu.yv + ox)) + ((oy + om) + (mc_.du + mc_.dv)) + mc_.ra;
        if (fx_row && own_x[l] && AIR) {
          const unsigned p = pcolB[l] + (unsigned)jr * rowB;
          FV3_MARCH_ST(*fv3_at(accLb, p), sum);
          FV3_MARCH_ST(*fv3_at(flLb, p), sum);
        }
        if (fy_row && own_y[l] && AIR) {
          const unsigned p = pcolB[l] + (unsigned)jf * rowB;
          FV3_MARCH_ST(*fv3_at(accMb, p), sum);
          FV3_MARCH_ST(*fv3_at(flMb, p), sum);
        }
        if (fx_row && own_y[l]) {
          const unsigned p = pcolB[l] + (unsigned)jr * rowB;
          FV3_MARCH_ST(*fv3_at(out0b, p), sum);
          FV3_MARCH_ST(*fv3_at(out1b, p), sum);
          if constexpr (AIR) FV3_MARCH_ST(*fv3_at(heatb, p), sum);
        }
      }
      if (PX_ABL == 1) return;
#endif
      // ---- phase 1: requests of the next steps; inner M fluxes at face r-2, the M-advected q at row r-3; del-n chains, own-lane part
      FV3_LANES(blk_, lane, l) {
        const Met mc_ = MN[Q][l];
        {
          const int r1 = GEN ? (r + 1 < r_end ? r + 1 : r_end) : r + 1, rn = GEN ? (r + 2 < r_end ? r + 2 : r_end) : r + 2;
#if PX_ABL == 2
          (void)r1;
          (void)rn;
          PX_KEEP(Ox[Q1][l]); PX_KEEP(Oy[Q1][l]); PX_KEEP(Om[Q1][l]);
          PX_KEEP(MN[Q1][l].du); PX_KEEP(MN[Q1][l].dv); PX_KEEP(MN[Q1][l].ra);
          PX_KEEP(R[Q2][l].q0); PX_KEEP(R[Q2][l].q1); PX_KEEP(R[Q2][l].cx); PX_KEEP(R[Q2][l].xv); PX_KEEP(R[Q2][l].ar); PX_KEEP(R[Q2][l].cy); PX_KEEP(R[Q2][l].yv);
#else
          load_opt(Q1, r1, l, gen_tag);
          MN[Q1][l] = load_met(r1, l);
          R[Q2][l] = load_row(rn, l, gen_tag);
#endif
        }
        const Row cu = R[Q][l];
        const Real era = RG(RG_RA, Q)[lane];
        const Real ar3 = RG(RG_AR, Q)[lane];
        const Real du1 = RG(RG_DU, Q2)[lane], du2 = RG(RG_DU, Q1)[lane], ra1 = RG(RG_RA, Q2)[lane], ra2 = RG(RG_RA, Q1)[lane];
        RG(RG_DU, Q)[lane] = mc_.du;
        RG(RG_DV, Q)[lane] = mc_.dv;
        RG(RG_RA, Q)[lane] = mc_.ra;
        const Real du0 = mc_.du;
        h_era[l] = era;
        if (xe) {
          h_mx3[l] = RG(RG_MX, Q)[lane];
          RG(RG_MX, Q)[lane] = mc_.mx;
          h_mx0[l] = mc_.mx;
        }
        const Real yv = px_move(cu.yv);
        const Real den_y = ar3 + y_prev[l] - yv;
        const Real rden_y = px_rcp(den_y);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const Real qraw = px_move(n == 0 ? cu.q0 : cu.q1);
          Real qy = qraw, qx = qraw;
          if (GEN && halo_cols && (r < 1 || r > nM)) {  // the two sweeps see the cube-corner cells through different remaps (rare, not prefetched)
            int lane_o = lane;
            FV3_LAUNDER(lane_o);
            const int lc = l0 - 3 + lane_o, lcc = lc < Led ? lc : Led;
            const int rc = r < Med ? r : Med;
            const Real *qf = (n == 0 ? q0f : q1f) + b;
            qy = cc<2>(qf, *gp, fl, lcc, rc);
            qx = cc<1>(qf, *gp, fl, lcc, rc);
            FV3_LANDED(qy);
            FV3_LANDED(qx);
          }
          {  // del-n chain (on the field itself: the corner-halo remaps belong to the patches): d2 of iteration s on row r-s, its M flux at face r-s
            const Real cf = n == 0 ? c0 : c1;
            const Real d0c = AIR ? cf * qraw : qraw;
            const Real fyc0 = du0 * (sd0[n][l] - d0c);
            const Real gxe0 = FV3_LANE_SHL(1, gx0[n], l, lane), gxe1 = FV3_LANE_SHL(1, gx1[n], l, lane);
            const Real d2c1 = (gx0[n][l] - gxe0 + gy0[n][l] - fyc0) * ra1;
            const Real fyc1 = du1 * (d2c1 - sd1[n][l]);
            const Real d2c2 = (gx1[n][l] - gxe1 + gy1[n][l] - fyc1) * ra2;
            dyf[n][l] = du2 * (d2c2 - sd2[n][l]);
            gy0[n][l] = fyc0;
            gy1[n][l] = fyc1;
            sd0[n][l] = d0c;
            sd1[n][l] = d2c1;
            sd2[n][l] = d2c2;
          }
          const Real a_ = w2[n][l], b_ = w3[n][l], c_ = w4[n][l], d_ = qy;  // q of rows r-3 .. r
          Real al_new;
          if (m_edge) {
            const Real *mmb = (const Real *)gdya + m2;
            auto My = [&](int s_) { return px_ld(mmb, pcolB[l] + (unsigned)s_ * rowB); };
            al_new = ppm_al_win(a_, b_, c_, d_, My, sy, Mlo, Mhi, npM);
            FV3_LANDED(al_new);
          } else {
            al_new = PPM_P1 * (b_ + c_) + PPM_P2 * (a_ + d_);
          }
          const PpmCell co = ppm_cell(al_q[n][l], al_new, b_, PX_ORD);
          al_q[n][l] = al_new;
          const Real fyi = ppm_face(cq[n][l], co, cu.cy);
          fyin[n][l] = fyi;
          cq[n][l] = co;
          const Real pn = yv * fyi;
          const Real qi = px_quot(a_ * ar3 + p_prev[n][l] - pn, den_y, rden_y);
          p_prev[n][l] = pn;
          sqx[n][l] = qx;
          sqi[n][l] = qi;
          h_q5[n][l] = qy;
          PX_FENCE_T();
        }
        y_prev[l] = yv;
        if constexpr (AIR) {
          smb[l] = w2[0][l];
        } else {
          smb[l] = mbk[l];
          h_mc[l] = px_move(Om[Q][l]);
        }
      }
      PX_FENCE();
      // ---- phase 2: inner L fluxes on row r, outer L fluxes on row r-3, final L fluxes of row r-3; del-n chains, L fluxes.
      //      The L sweeps share their reconstruction between neighbouring lanes: a lane forms the edge value at the low face of ITS cell
      //      and that cell's limited profile once (ppm_flux_int forms three edge values and two cells per face, two of the edge values
      //      and one cell being the neighbouring lane's too); the high edge value is the next lane's low one, the upwind-side cell of
      //      the lane's face the previous lane's -- wavefront shuffles of results instead of operands, the limiter flag as a shifted
      //      lane mask (a scalar instruction).  Same expressions on the same operands: the same bits.
      //      (2a, 2b, 2c are one stretch of code on the device; the host emulation needs the neighbour's value complete before it is read)
      FV3_LANES(blk_, lane, l) {  // 2a: edge value at the low face of the lane's cell, both sweeps
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          s_al[0][n][l] = PPM_P1 * (FV3_LANE_SHR(1, sqx[n], l, lane) + sqx[n][l]) + PPM_P2 * (FV3_LANE_SHR(2, sqx[n], l, lane) + FV3_LANE_SHL(1, sqx[n], l, lane));
          s_al[1][n][l] = PPM_P1 * (FV3_LANE_SHR(1, sqi[n], l, lane) + sqi[n][l]) + PPM_P2 * (FV3_LANE_SHR(2, sqi[n], l, lane) + FV3_LANE_SHL(1, sqi[n], l, lane));
        }
        if (xe) {
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            s_al[0][n][l] = px_al_edge(s_al[0][n][l], akind[l], FV3_LANE_SHR(2, sqx[n], l, lane), FV3_LANE_SHR(1, sqx[n], l, lane), sqx[n][l], FV3_LANE_SHL(1, sqx[n], l, lane),
                                       FV3_LANE_SHR(2, h_mx0, l, lane), FV3_LANE_SHR(1, h_mx0, l, lane), h_mx0[l], FV3_LANE_SHL(1, h_mx0, l, lane));
            s_al[1][n][l] = px_al_edge(s_al[1][n][l], akind[l], FV3_LANE_SHR(2, sqi[n], l, lane), FV3_LANE_SHR(1, sqi[n], l, lane), sqi[n][l], FV3_LANE_SHL(1, sqi[n], l, lane),
                                       FV3_LANE_SHR(2, h_mx3, l, lane), FV3_LANE_SHR(1, h_mx3, l, lane), h_mx3[l], FV3_LANE_SHL(1, h_mx3, l, lane));
          }
        }
      }
      FV3_LANES(blk_, lane, l) {  // 2b: the lane's cell
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const PpmCell ci = ppm_cell(s_al[0][n][l], FV3_LANE_SHL(1, s_al[0][n], l, lane), sqx[n][l], PX_ORD);
          const PpmCell co = ppm_cell(s_al[1][n][l], FV3_LANE_SHL(1, s_al[1][n], l, lane), sqi[n][l], PX_ORD);
          s_bl[0][n][l] = ci.bl;
          s_br[0][n][l] = ci.br;
          s_sm[0][n][l] = ci.sm;
          s_bl[1][n][l] = co.bl;
          s_br[1][n][l] = co.br;
          s_sm[1][n][l] = co.sm;
        }
      }
      FV3_LANES(blk_, lane, l) {  // 2c: the faces
        const Row cu = R[Q][l];
        const Real cx = cu.cx, xv = cu.xv;
        const Real cx3 = RG(RG_CX, Q)[lane], xv3 = RG(RG_XV, Q)[lane];
        const Real dv0 = RG(RG_DV, Q)[lane], dv1 = RG(RG_DV, Q2)[lane], dv2 = RG(RG_DV, Q1)[lane];
        Real o_dx[2], o_dy[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const Real e0 = FV3_LANE_SHR(1, sd0[n], l, lane), e1 = FV3_LANE_SHR(1, sd1[n], l, lane), e2 = FV3_LANE_SHR(1, sd2[n], l, lane);
          gx0[n][l] = dv0 * (e0 - sd0[n][l]);
          gx1[n][l] = dv1 * (sd1[n][l] - e1);
          o_dx[n] = dxd[n][l];  // final L flux of row r-3 (formed at the previous step)
          o_dy[n] = dyf[n][l];
          dxd[n][l] = dv2 * (sd2[n][l] - e2);  // ... of row r-2
        }
        if (GEN && pz) {  // faces on a cube-corner patch: the staged chain's fluxes
          int lane_o = lane;
          FV3_LAUNDER(lane_o);
          const int lc = l0 - 3 + lane_o;
          const bool px_ = jr >= 1 && jr <= nM && on_patch(lc, jr), py_ = lc <= nL && on_patch(lc, jf);
          if (px_) {
            o_dx[0] = px_ld(pL0 + b, pcolB[l] + (unsigned)jr * rowB);
            o_dx[1] = px_ld(pL1 + b, pcolB[l] + (unsigned)jr * rowB);
          }
          if (py_) {
            o_dy[0] = px_ld(pM0 + b, pcolB[l] + (unsigned)jf * rowB);
            o_dy[1] = px_ld(pM1 + b, pcolB[l] + (unsigned)jf * rowB);
          }
          FV3_LANDED(o_dx[0]);
          FV3_LANDED(o_dx[1]);
          FV3_LANDED(o_dy[0]);
          FV3_LANDED(o_dy[1]);
        }
        h_ody[0][l] = o_dy[0];
        h_ody[1][l] = o_dy[1];
        if constexpr (AIR) {
          h_zx0[l] = o_dx[1];
          h_zy0[l] = zyp[l];
          h_zy1[l] = o_dy[1];
          zxo[l] = o_dx[1];
        }
        const Real mb = AIR ? w2[0][l] : mbk[l];  // old air mass of (lc, r-3)
        const Real mw = FV3_LANE_SHR(1, smb, l, lane) + mb;
        Real vm = Ox[Q][l];  // TRC: the stored air-mass flux of the face
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const Real fi3 = RG(RG_FI + n, Q)[lane];
          Real ff[2];
#pragma unroll
          for (int w = 0; w < 2; ++w) {  // w = 0: inner flux (q on row r), 1: outer flux (the M-advected q on row r-3)
            const Real qv = w == 0 ? sqx[n][l] : sqi[n][l];
            const PpmCell cm{FV3_LANE_SHR(1, s_bl[w][n], l, lane), FV3_LANE_SHR(1, s_br[w][n], l, lane), w == 0 ? FV3_LANE_SHR(1, sqx[n], l, lane) : FV3_LANE_SHR(1, sqi[n], l, lane),
                             FV3_LANE_SHR1_FLAG(s_sm[w][n], l, lane)};
            const PpmCell c0{s_bl[w][n][l], s_br[w][n][l], qv, s_sm[w][n][l]};
            ff[w] = ppm_face(cm, c0, w == 0 ? cx : cx3);
          }
          const Real fxin = ff[0], fxout = ff[1];
          Real v;
          if (AIR && n == 0) {  // air mass: area-flux weighted, plain damping flux
            v = (Real)0.5 * (fxout + fi3) * xv3;
            v = v + o_dx[n];
            if (fx_row && own_x[l]) {
              const unsigned p = pcolB[l] + (unsigned)jr * rowB;
              FV3_MARCH_ST(*fv3_at(accLb, p), Ox[Q][l] + v);
              if constexpr (!CPL) FV3_MARCH_ST(*fv3_at(flLb, p), v);
            }
            if constexpr (CPL) XC(XC_FX, Q)[lane] = v;  // (every lane: the other role reads its own and the next lane's)
            vm = v;
          } else {  // riding on the air-mass flux; q_con / pt with the mass-weighted damping flux
            v = (Real)0.5 * (fxout + fi3) * vm;
            if constexpr (!AIR) v = v + (Real)0.5 * (n == 0 ? damp_t : damp_vt) * mw * o_dx[n];
          }
          fxk[n][l] = v;
          RG(RG_FI + n, Q)[lane] = fxin;
          px[n][l] = xv * fxin;
          PX_FENCE_T();
        }
        RG(RG_CX, Q)[lane] = cx;
        RG(RG_XV, Q)[lane] = xv;
        sxv[l] = xv;
      }
      PX_FENCE();
      // ---- phase 3: the L-advected q on row r, outer M fluxes at face r-2, final M fluxes, the cell update of (lc, r-3)
      FV3_LANES(blk_, lane, l) {
        const Row cu = R[Q][l];
        const Real x1 = FV3_LANE_SHL(1, sxv, l, lane);
        const Real ar = cu.ar;
        const Real den_x = ar + cu.xv - x1;
        const Real rden_x = px_rcp(den_x);
        const Real era = h_era[l];
        const Real mb = AIR ? w2[0][l] : mbk[l], mc = AIR ? w3[0][l] : h_mc[l];  // old air mass of (lc, r-3), (lc, r-2)
        Real vy[2];
        Real vm = Oy[Q][l];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const Real p1 = FV3_LANE_SHL(1, px[n], l, lane);
          const Real qj = px_quot(h_q5[n][l] * ar + px[n][l] - p1, den_x, rden_x);
          const Real a_ = v2[n][l], b_ = v3[n][l], c_ = v4[n][l], d_ = qj;
          Real al_new;
          if (m_edge) {
            const Real *mmb = (const Real *)gdya + m2;
            auto My = [&](int s_) { return px_ld(mmb, pcolB[l] + (unsigned)s_ * rowB); };
            al_new = ppm_al_win(a_, b_, c_, d_, My, sy, Mlo, Mhi, npM);
            FV3_LANDED(al_new);
          } else {
            al_new = PPM_P1 * (b_ + c_) + PPM_P2 * (a_ + d_);
          }
          const PpmCell co = ppm_cell(al_v[n][l], al_new, b_, PX_ORD);
          al_v[n][l] = al_new;
          const Real fyout = ppm_face(cv[n][l], co, cu.cy);
          cv[n][l] = co;
          v2[n][l] = b_;
          v3[n][l] = c_;
          v4[n][l] = d_;
          Real v;
          if (AIR && n == 0) {
            v = (Real)0.5 * (fyout + fyin[n][l]) * cu.yv;
            v = v + h_ody[n][l];
            if (fy_row && own_y[l]) {
              const unsigned p = pcolB[l] + (unsigned)jf * rowB;
              FV3_MARCH_ST(*fv3_at(accMb, p), Oy[Q][l] + v);
              if constexpr (!CPL) FV3_MARCH_ST(*fv3_at(flMb, p), v);
            }
            if constexpr (CPL) {
              XC(XC_FY, Q)[lane] = v;
              XC(XC_OM, Q)[lane] = w3[0][l];  // the old air mass of (lc, r-2)
            }
            vm = v;
          } else {
            v = (Real)0.5 * (fyout + fyin[n][l]) * vm;
            if constexpr (!AIR) v = v + (Real)0.5 * (n == 0 ? damp_t : damp_vt) * (mb + mc) * h_ody[n][l];
          }
          vy[n] = v;
          PX_FENCE_T();
        }
        Real fxe[2];  // (read outside the branch below: a shuffle needs the source lane active)
        fxe[0] = FV3_LANE_SHL(1, fxk[0], l, lane);
        fxe[1] = FV3_LANE_SHL(1, fxk[1], l, lane);
        Real fe_air = (Real)0, zx1 = (Real)0;
        if constexpr (AIR)
          zx1 = FV3_LANE_SHL(1, zxo, l, lane);
        else
          fe_air = FV3_LANE_SHL(1, Ox[Q], l, lane);  // the air-mass flux through the high L face of the cell: the neighbouring lane's
        if (fx_row && own_y[l]) {
          const unsigned p = pcolB[l] + (unsigned)jr * rowB;
          Real up[2];
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const Real dv_ = (fxk[n][l] - fxe[n] + fyp[n][l] - vy[n]) * era;
            up[n] = (AIR && n == 0) ? w2[n][l] + dv_ : mb * w2[n][l] + dv_;
          }
          if constexpr (AIR) {
            const Real dpn = up[0];
            FV3_MARCH_ST(*fv3_at(out0b, p), dpn);
            Real wn = px_quot(up[1], dpn, px_rcp(dpn));
            const Real dwv = (h_zx0[l] - zx1 + h_zy0[l] - h_zy1[l]) * era;  // (x terms first)
            const Real hs = dd8 - dwv * (w2[1][l] + (Real)0.5 * dwv);
            wn = wn + dwv;
            FV3_MARCH_ST(*fv3_at(out1b, p), wn);
            FV3_MARCH_ST(*fv3_at(heatb, p), hs);
          } else {
            const Real dpn = mb + (Ox[Q][l] - fe_air + fyp_air[l] - Oy[Q][l]) * era;  // the new air mass, as the delp + w march formed it
            const Real rdpn = px_rcp(dpn);
            FV3_MARCH_ST(*fv3_at(out0b, p), px_quot(up[0], dpn, rdpn));
            FV3_MARCH_ST(*fv3_at(out1b, p), px_quot(up[1], dpn, rdpn));
          }
        }
        if constexpr (!AIR) {
          mbk[l] = h_mc[l];
          fyp_air[l] = Oy[Q][l];
        } else {
          zyp[l] = h_zy1[l];
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          fyp[n][l] = vy[n];
          w2[n][l] = w3[n][l];
          w3[n][l] = w4[n][l];
          w4[n][l] = h_q5[n][l];
        }
        RG(RG_AR, Q)[lane] = ar;
      }
      PX_FENCE();
      if constexpr (CPL) blk_.group_sync();  // one barrier per row step of the group (LDS traffic complete; the vector-memory counter is NOT drained)
    };
    // a row step of the GROUP: the leading wave (delp + w) at row r, the lagging wave (q_con + pt) at row r - 1 with the rotation index that row has
    auto gstep = [&](const int r_, auto q_tag, auto gen_tag) {
      constexpr int Q = decltype(q_tag)::value;
      if constexpr (XIN)
        step(r_ - 1, std::integral_constant<int, (Q + 2) % 3>{}, gen_tag);
      else
        step(r_, q_tag, gen_tag);
    };

    // ---- the march: 6 general steps (windows fill; S tile edge), branch-free triples, general triples to the end (rows past r_end are
    //      clamped loads and masked stores)
    int r_hi = cb + 3 < fb + 2 ? cb + 3 : fb + 2;  // last row whose step owns both its cell row and its M face ...
    if (r_hi > r_end - 2) r_hi = r_end - 2;        // ... and requests no row past the segment's last
    if (Mhi && r_hi > nM) r_hi = nM;               // ... and meets no N tile-edge formula
    if (pz && r_hi > nM - FV3_D6_PATCH + 2) r_hi = nM - FV3_D6_PATCH + 2;  // ... and consumes no face of a N corner patch (first: r - 2 = nM + 2 - PATCH)
    // (coupled: both waves of a group run the SAME sequence of group steps -- one barrier each --, the lagging wave one row behind: one more general triple at
    //  the head, so that its first branch-free row has its windows filled too, and one more row at the end)
    const int head = (pz && r0 <= FV3_D6_PATCH + 3 ? 15 : 6) + (CPL ? 3 : 0);  // (S corner patch: faces up to r - 3 = PATCH)
    int r = r0;
#pragma clang loop unroll(disable)
    for (int part = 0; part < 2; ++part) {  // (one copy of the general triple in the code: head and tail are two trips of this loop)
      const int stop = part == 0 ? r0 + head - 1 : r_end + (CPL ? 1 : 0);
#pragma clang loop unroll(disable)
      for (; r <= stop; r += 3) {
        gstep(r, std::integral_constant<int, 0>{}, std::true_type{});
        gstep(r + 1, std::integral_constant<int, 1>{}, std::true_type{});
        gstep(r + 2, std::integral_constant<int, 2>{}, std::true_type{});
      }
      if (part == 0) {
#pragma clang loop unroll(disable)
        for (; r + 2 <= r_hi; r += 3) {
          gstep(r, std::integral_constant<int, 0>{}, std::false_type{});
          gstep(r + 1, std::integral_constant<int, 1>{}, std::false_type{});
          gstep(r + 2, std::integral_constant<int, 2>{}, std::false_type{});
        }
      }
    }
  };
  const int gx = KB ? KB : nstrip, gy = KB ? nstrip * nseg : nseg, gz = KB ? g.nsub * nblk : g.nsub * nk;
  if constexpr (MODE == PX_BOTH) {
#ifndef FV3_HOST_EMU
    launch_wave_groups3<PX_WPE, 2>(c, s, gx, gy, gz, 2 * smem + sizeof(Real) * 3 * NXCH * FV3_WAVE, [=] FV3_HD(const Blk &blk, char *smem_, int wave) {
      Real *xch = (Real *)(smem_ + 2 * smem);
      if (wave == 0)
        body(std::integral_constant<int, PX_AIR>{}, std::true_type{}, blk, smem_, xch);
      else
        body(std::integral_constant<int, PX_TRC>{}, std::true_type{}, blk, smem_ + smem, xch);
    });
#endif
  } else {
    launch_waves<PX_WPE>(c, s, gx, gy, gz, smem, [=] FV3_HD(const Blk &blk, char *smem_) { body(std::integral_constant<int, MODE>{}, std::false_type{}, blk, smem_, (Real *)nullptr); });
  }
}

}  // namespace

// role 1 = delp + w, 2 = q_con + pt, 3 = both as coupled wave pairs (device only; the host emulation runs the two roles one after the other: a hand-over
// between concurrently running waves has no emulation)
void dsw_pair_march(fv3_ctx *c, fv3_stream_t s, const DswScalars &a, int role, int k_lo, int k_hi) {
  if (getenv("FV3_DEBUG_FD")) fprintf(stderr, "[d_sw] pair march, role %d, levels %d..%d\n", role, k_lo, k_hi);
  if (role == PX_AIR)
    pair_march_t<PX_AIR>(c, s, a, k_lo, k_hi);
  else if (role == PX_TRC)
    pair_march_t<PX_TRC>(c, s, a, k_lo, k_hi);
  else {
#ifdef FV3_HOST_EMU
    pair_march_t<PX_AIR>(c, s, a, k_lo, k_hi);
    pair_march_t<PX_TRC>(c, s, a, k_lo, k_hi);
#else
    pair_march_t<PX_BOTH>(c, s, a, k_lo, k_hi);
#endif
  }
}
