// fv3_tp2x.hip -- the single-tracer transport marches with their del-n chain inside, round-5 form: d_sw's vorticity transport with the wind
// update (SX_WIND) and update_dz_d's interface-height transport with the advective-form height update (SX_AREA).  Same operator, same
// expressions in the same order as tp2d_stream_t<TF_WIND | TF_FD, 6, true> / <TF_EPI | TF_AREA | TF_FD, 6, true> of fv3_tp2d.hip on the strips
// away from the W / E tile edges (bitwise equal: FV3_TP2D_MARCH=old is the A/B switch; tests/test_parity.py::
// test_single_march_is_bitwise_the_round4_march); CPU twin: oracle/fv3_oracle/fvtp2d.py, d_sw.py, nh.py.  [SURVEY A.3.7, A.4, A.8]
//
// The construction is that of fv3_tp4x.hip (see there for the reasons): no rare path in the branch-free row step -- the rows either side of a S / N tile
// edge, the rows a segment does not own and the cube-corner remaps / patch fluxes run a general form of the step; the W / E one-sided formulas are
// evaluated in the lanes of a tile-edge strip (px_al_edge), so the march serves EVERY tile --, the march unrolled by three with static rotation of the sets
// in flight, scalar-base addressing, the L sweeps' reconstruction shared between neighbouring lanes, one refined reciprocal per denominator pair.
#include <type_traits>

#include "fv3_ops.h"
#include "fv3_math.h"
#define FV3_MARCH_ST(lhs, val) FV3_ST_NT(lhs, val)
#include "fv3_ppm.h"
#include "fv3_march.h"

namespace {

#define SX_OUT 58
#define SX_ORD 6
enum { SX_WIND = 1, SX_AREA = 2 };
#ifndef SX_WPE
#define SX_WPE (sizeof(Real) == 4 ? 3 : 2)
#endif

struct SxArgs {
  const Real *q, *crx, *cry, *xfx, *yfx;
  const Real *coef;  // d2 of iteration 0 = coef[k] * q
  // SX_WIND
  Real *u, *v, *u_pre, *v_pre, *du, *dv;
  const Real *ke, *add;  // add: 2-D term added to q where the transport loads it (f0)
  // SX_WIND with the damping-heat epilogue (HEAT): the corner damping field, the new air mass, the heat of w's damping (AIR march), the accumulated
  // heat source (read + written), the per-level fraction d_con
  const Real *vdamp, *ndelp, *heat_s, *dcon;
  Real *heat_src;
  const Real *heat_zeros = nullptr;  // non-null (first sub-step of a call inside the sequencer): the accumulated heat is read as zero from this one-plane block of zeros
  // HEAT: copies (made before the march) of the u rows / v columns on the boundaries between segments / strips -- see sx_side_copy
  const Real *u_side, *v_side;
  // SX_AREA
  Real *out;
  // the staged chain's damping fluxes on the cube-corner patches (read where the chain of the march does not reach: corner-halo remaps)
  const Real *pfx, *pfy;
};

// HEAT (SX_WIND): d_sw's damping heat as the epilogue of the cell (i, r-3) -- the pre-damping winds and the damping increments of the
// cell's four faces are in the wave (this step's and the previous step's u face, the lane's and the next lane's v face), so the four fields the
// round-4 sequence stored for the damping-heat kernel and read back there (u_pre, v_pre, the two increments) never exist.  Expressions and order:
// the damping-heat kernel's (fv3_dsw.hip, the last launch of fv3_d_sw_out).
template <int KIND, bool HEAT = false>
void single_march_t(fv3_ctx *c, fv3_stream_t s, const SxArgs &a, int k_lo, int k_hi) {
  constexpr bool WIND = KIND == SX_WIND, AREA = KIND == SX_AREA;
  static_assert(!HEAT || WIND, "the damping heat is the epilogue of the wind form");
  const Geo g = c->g;
  const int nk = k_hi - k_lo + 1;
  if (nk <= 0) return;
  const int nx = g.nx, ny = g.ny, nh = g.nh, go = g.o, sj32 = g.sj32;
  // (launch geometry of tp2d_stream_t: the two kernels split the same tiles between them)
  const int nstrip = (nx + 1 + SX_OUT - 1) / SX_OUT;
  const int seg = fv3_pick_seg((long)nstrip * ((ny + 63) / 64) * g.nsub * nk, 2);
  const int nseg = (ny + seg - 1) / seg;
  static const int kb_env = getenv("FV3_Q4_KB") ? atoi(getenv("FV3_Q4_KB")) : FV3_Q4_KB_DEFAULT;
  const int KB = kb_env > 0 ? (kb_env < nk ? kb_env : nk) : 0;
  const int nblk = KB ? (nk + KB - 1) / KB : 0;
  const long st = g.st, sk = g.sk, st2 = g.st2;
  const unsigned char *gflags = c->g_dev->flags;
  const Geo *gp = c->g_dev;
  const MPtr garea = g.area, grarea = g.rarea, gd6v = g.del6_v, gd6u = g.del6_u, gdya = g.dya, gdx = g.dx, gdy = g.dy, gdxa = g.dxa, grdx = g.rdx, grdy = g.rdy, grs2 = g.rsin2, gcs = g.cosa_s;
  const SxArgs A = a;
  constexpr int NRING = 8;
  enum { RG_DU = 0, RG_DV = 1, RG_RA = 2, RG_CX = 3, RG_XV = 4, RG_AR = 5, RG_FI = 6, RG_MX = 7 };
  const size_t smem = sizeof(Real) * (size_t)NRING * 3 * FV3_WAVE;
  launch_waves<SX_WPE>(c, s, KB ? KB : nstrip, KB ? nstrip * nseg : nseg, KB ? g.nsub * nblk : g.nsub * nk, smem, [=] FV3_HD(const Blk &blk_, char *smem_) {
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
    const int i0 = 1 + bx * SX_OUT;
    const int ja = 1 + by * seg;
    const int jb = by == nseg - 1 ? ny + 1 : ja + seg - 1;
    const int cb = jb < ny ? jb : ny;
    const int ied = nx + nh, jsd = 1 - nh, jed = ny + nh, npy = ny + 1;
    const int r_end = jb + 3 < jed ? jb + 3 : jed;
    const bool S = fl & FV3_S, N = fl & FV3_N;
    // Cube corners (general steps only): the rows outside 1 .. ny of a strip with corner-halo columns see the field through the two
    // copy_corners remaps (one per sweep direction); the del-n fluxes of the faces on a corner patch come from the staged chain's arrays.
    const bool halo_cols = i0 - 3 < 1 || i0 + FV3_WAVE - 4 > nx;
    const bool c_sw_ = (fl & (FV3_W | FV3_S)) == (FV3_W | FV3_S), c_se = (fl & (FV3_E | FV3_S)) == (FV3_E | FV3_S);
    const bool c_ne = (fl & (FV3_E | FV3_N)) == (FV3_E | FV3_N), c_nw = (fl & (FV3_W | FV3_N)) == (FV3_W | FV3_N);
    const bool pz = ((c_sw_ || c_nw) && i0 - 3 <= FV3_D6_PATCH) || ((c_se || c_ne) && i0 + FV3_WAVE - 4 >= nx + 2 - FV3_D6_PATCH);
    auto on_patch = [&](int i, int j) -> bool {
      const bool wi = i >= 1 && i <= FV3_D6_PATCH, ei = i >= nx + 2 - FV3_D6_PATCH && i >= 1 && i <= nx + 1;
      const bool sj = j >= 1 && j <= FV3_D6_PATCH, nj = j >= ny + 2 - FV3_D6_PATCH && j >= 1 && j <= ny + 1;
      return (wi && sj && c_sw_) || (ei && sj && c_se) || (ei && nj && c_ne) || (wi && nj && c_nw);
    };
    // XE: this strip has W / E one-sided formulas among its x faces (the three faces either side of the tile edge).  They use the same four cells as the
    // interior edge value (+ the cell widths dxa of those cells), so a lane evaluates them beside the interior form and keeps what its face calls for.
    const bool Wst = (fl & FV3_W) && i0 <= 3, Est = (fl & FV3_E) && i0 + SX_OUT + 1 >= nx;
    const bool xe = Wst || Est;
    const long b = t * st + k * sk, m2 = t * st2;
    const Real *const dxab = (const Real *)gdxa + m2;
    const Real *const qb = A.q + b, *const crxb = A.crx + b, *const cryb = A.cry + b, *const xfxb = A.xfx + b, *const yfxb = A.yfx + b;
    const Real *const areab = (const Real *)garea + m2, *const rab = (const Real *)grarea + m2, *const d6vb = (const Real *)gd6v + m2, *const d6ub = (const Real *)gd6u + m2;
    const Real *const addb = WIND ? A.add + m2 : nullptr, *const dxb = (const Real *)gdx + m2, *const dyb = (const Real *)gdy + m2;
    const Real *const keb = WIND ? A.ke + b : nullptr;
    Real *const ub = WIND ? A.u + b : nullptr, *const vb = WIND ? A.v + b : nullptr, *const upb = WIND ? A.u_pre + b : nullptr, *const vpb = WIND ? A.v_pre + b : nullptr;
    Real *const dub = WIND ? A.du + b : nullptr, *const dvb = WIND ? A.dv + b : nullptr;
    Real *const outb = AREA ? A.out + b : nullptr;
    const Real *const vdb = HEAT ? A.vdamp + b : nullptr, *const ndpb = HEAT ? A.ndelp + b : nullptr, *const hsb = HEAT ? A.heat_s + b : nullptr;
    Real *const hob = HEAT ? A.heat_src + b : nullptr;
    const bool heat_first = HEAT && A.heat_zeros;
    const Real *const hob_ld = heat_first ? A.heat_zeros : hob;
    const Real *const rdxb = (const Real *)grdx + m2, *const rdyb = (const Real *)grdy + m2, *const rs2b = (const Real *)grs2 + m2, *const csb = (const Real *)gcs + m2;
    const Real dcon = HEAT ? A.dcon[k] : (Real)0;
    const Real *const usideb = HEAT ? A.u_side + b : nullptr, *const vsideb = HEAT ? A.v_side + b : nullptr;
    const bool side_row = HEAT && by != nseg - 1, side_col = HEAT && bx != nstrip - 1;
    const Real dcoef = A.coef[k];
    const unsigned rowB = (unsigned)sj32 * (unsigned)sizeof(Real);
    Real *const ring = (Real *)smem_;
    auto RG = [&](int var, int slot) -> Real * { return ring + (var * 3 + slot) * FV3_WAVE; };

    struct Row {
      Real qy, cx, xv, ar, cy, yv;
    };
    struct Met {
      Real du, dv, ra, qa;  // qa (WIND): the 2-D term added to q
      Real mx;              // XE strips: the cell width dxa of the lane's column
    };
    Row R[3][FV3_LPT];
    Met MN[3][FV3_LPT];
    // WIND: the winds / kinetic energy / grid spacings the epilogue of a step needs, requested one step ahead
    Real Ou[3][FV3_LPT], Odx[3][FV3_LPT], Okf[3][FV3_LPT], Ov[3][FV3_LPT], Ody[3][FV3_LPT];
    // HEAT: corner damping field / 1/dx of face row r-2; 1/dy, 1/sin^2, cos, new air mass, w's damping heat, accumulated heat of cell row r-3
    Real Hvd[3][FV3_LPT], Hrx[3][FV3_LPT], Hry[3][FV3_LPT], Hrs[3][FV3_LPT], Hcs[3][FV3_LPT], Hdp[3][FV3_LPT], Hhs[3][FV3_LPT], Hho[3][FV3_LPT];
    Real vdp[FV3_LPT], ubp[FV3_LPT], fyq[FV3_LPT];  // ... the damping field of row r-3, ub / fy of the u face r-3 (the previous step's)
    Real s_vb[FV3_LPT], s_fx[FV3_LPT];              // ... vb / fx of the lane's v face (row r-3): read by lane - 1
    unsigned pcolB[FV3_LPT];
    bool own_x[FV3_LPT], own_y[FV3_LPT];
    const Real *vsrc[FV3_LPT];  // HEAT: where the lane reads v (the side copy for the column the next strip owns)
    Real w2[FV3_LPT], w3[FV3_LPT], w4[FV3_LPT], al_q[FV3_LPT], v2[FV3_LPT], v3[FV3_LPT], v4[FV3_LPT], al_v[FV3_LPT];
    PpmCell cq[FV3_LPT], cv[FV3_LPT];
    Real p_prev[FV3_LPT], y_prev[FV3_LPT], fyin[FV3_LPT], px[FV3_LPT], fxk[FV3_LPT], fyp[FV3_LPT], sqx[FV3_LPT], sqi[FV3_LPT], sxv[FV3_LPT];
    Real sd0[FV3_LPT], sd1[FV3_LPT], sd2[FV3_LPT], gx0[FV3_LPT], gx1[FV3_LPT], gy0[FV3_LPT], gy1[FV3_LPT], dxd[FV3_LPT], dyf[FV3_LPT], zyp[FV3_LPT], zxo[FV3_LPT];
    Real wkr[FV3_LPT], xjr[FV3_LPT];
    Real s_al[2][FV3_LPT], s_bl[2][FV3_LPT], s_br[2][FV3_LPT];
    bool s_sm[2][FV3_LPT];
    int akind[FV3_LPT];                  // XE strips: which edge-value form the lane's cell takes (0 interior; 1, 2, 3: ppm_al's one-sided forms)
    Real h_mx0[FV3_LPT], h_mx3[FV3_LPT];  // XE strips: dxa of the lane's column on rows r / r-3
    Real h_era[FV3_LPT], h_q5[FV3_LPT], h_ypp[FV3_LPT], h_ox[FV3_LPT], h_oy[FV3_LPT], h_zy0[FV3_LPT];

    auto load_row = [&](int r, int l, auto gen_tag) -> Row {
      constexpr bool GEN = decltype(gen_tag)::value;
      const int rf = GEN ? (r - 2 < jsd ? jsd : r - 2) : r - 2;
      const unsigned p0 = pcolB[l] + (unsigned)r * rowB, pf = pcolB[l] + (unsigned)rf * rowB;
      Row w;
      w.qy = px_ld3(qb, p0);
      w.cx = px_ld3(crxb, p0);
      w.xv = px_ld3(xfxb, p0);
      w.ar = px_ld(areab, p0);
      w.cy = px_ld3(cryb, pf);
      w.yv = px_ld3(yfxb, pf);
      return w;
    };
    auto load_opt = [&](int q, int r, int l, auto gen_tag) {  // what step r consumes: face r-2, row r-3 (into set q)
      if constexpr (WIND) {
        constexpr bool GEN = decltype(gen_tag)::value;
        const int r3 = GEN ? (r - 3 < jsd ? jsd : r - 3) : r - 3, rf = GEN ? (r - 2 < jsd ? jsd : r - 2) : r - 2;
        const unsigned p3 = pcolB[l] + (unsigned)r3 * rowB, pf = pcolB[l] + (unsigned)rf * rowB;
        // HEAT: the cell epilogue also uses the winds of two faces the wave does not own -- the u row of the next segment's first face, the v column
        // of the next strip's first face.  Their owners update them IN PLACE at a time of their own, so these two are read from copies made
        // before the march (sx_side_copy); without the epilogue the values computed on faces a wave does not own are discarded.
        if (HEAT && GEN && side_row && rf == jb + 1)
          Ou[q][l] = px_ld3(usideb, pf);
        else
          Ou[q][l] = px_ld3(ub, pf);
        Odx[q][l] = px_ld(dxb, pf);
        Okf[q][l] = px_ld3(keb, pf);
        if constexpr (HEAT)
          Ov[q][l] = px_ld3(vsrc[l], p3);
        else
          Ov[q][l] = px_ld3(vb, p3);
        Ody[q][l] = px_ld(dyb, p3);
        if constexpr (HEAT) {
          Hvd[q][l] = px_ld3(vdb, pf);
          Hrx[q][l] = px_ld(rdxb, pf);
          Hry[q][l] = px_ld(rdyb, p3);
          Hrs[q][l] = px_ld(rs2b, p3);
          Hcs[q][l] = px_ld(csb, p3);
          Hdp[q][l] = px_ld3(ndpb, p3);
          Hhs[q][l] = px_ld3(hsb, p3);
          Hho[q][l] = px_ld3(hob_ld, p3);
        }
      }
    };
    auto load_met = [&](int r, int l) -> Met {
      const unsigned pm = pcolB[l] + (unsigned)r * rowB;
      Met m;
      m.du = px_ld(d6ub, pm);
      m.dv = px_ld(d6vb, pm);
      m.ra = px_ld(rab, pm);
      m.qa = WIND ? px_ld(addb, pm) : (Real)0;
      m.mx = (Real)1;
      if (xe) m.mx = px_ld(dxab, pm);
      return m;
    };

    const int r0 = ja - 3;
    FV3_LANES(blk_, lane, l) {
      const int i = i0 - 3 + lane, ic = i < ied ? i : ied;
      pcolB[l] = (unsigned)(go * sj32 + go + ic) * (unsigned)sizeof(Real);
      own_x[l] = i >= i0 && i < i0 + SX_OUT && i <= nx + 1;
      own_y[l] = i >= i0 && i < i0 + SX_OUT && i <= nx;
      vsrc[l] = (side_col && i == i0 + SX_OUT) ? vsideb : (const Real *)vb;
      akind[l] = 0;
      if (Wst && i >= 0 && i <= 2) akind[l] = 1 + i;
      if (Est && i >= nx && i <= nx + 2) akind[l] = 1 + (i - nx);
      h_mx0[l] = h_mx3[l] = (Real)1;
      w2[l] = w3[l] = w4[l] = al_q[l] = v2[l] = v3[l] = v4[l] = al_v[l] = (Real)0;
      cq[l] = cv[l] = PpmCell{(Real)0, (Real)0, (Real)0, false};
      p_prev[l] = y_prev[l] = fyin[l] = px[l] = fxk[l] = fyp[l] = sqx[l] = sqi[l] = sxv[l] = (Real)0;
      sd0[l] = sd1[l] = sd2[l] = gx0[l] = gx1[l] = gy0[l] = gy1[l] = dxd[l] = dyf[l] = zyp[l] = zxo[l] = wkr[l] = xjr[l] = (Real)0;
      h_era[l] = h_q5[l] = h_ypp[l] = h_ox[l] = h_oy[l] = h_zy0[l] = (Real)0;
      for (int w = 0; w < 2; ++w) {
        s_al[w][l] = s_bl[w][l] = s_br[w][l] = (Real)0;
        s_sm[w][l] = false;
      }
      for (int v = 0; v < NRING; ++v)
        for (int q = 0; q < 3; ++q) RG(v, q)[lane] = (v == RG_AR || v == RG_MX) ? (Real)1 : (Real)0;  // (warm-up steps: outputs masked, keep the divisions finite)
      for (int q = 0; q < 3; ++q) {
        Ou[q][l] = Odx[q][l] = Okf[q][l] = Ov[q][l] = Ody[q][l] = (Real)0;
        Hvd[q][l] = Hrx[q][l] = Hry[q][l] = Hrs[q][l] = Hcs[q][l] = Hdp[q][l] = Hhs[q][l] = Hho[q][l] = (Real)0;
      }
      vdp[l] = ubp[l] = fyq[l] = s_vb[l] = s_fx[l] = (Real)0;
      MN[0][l] = load_met(r0, l);
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

    auto step = [&](const int r_, auto q_tag, auto gen_tag) {
      constexpr int Q = decltype(q_tag)::value, Q1 = (Q + 1) % 3, Q2 = (Q + 2) % 3;
      constexpr bool GEN = decltype(gen_tag)::value;
      int r = r_;
      PX_OPAQUE_S(r);
      const int sy = r - 1;
      const bool y_edge = GEN && ((S && sy >= 0 && sy <= 2) || (N && sy >= npy - 1 && sy <= npy + 1));
      const int jr = r - 3, jf = r - 2;
      const bool fx_row = !GEN || (jr >= ja && jr <= cb), fy_row = !GEN || (jf >= ja && jf <= jb);
#if PX_ABL == 1
      FV3_LANES(blk_, lane, l) {
        const Met mc_ = MN[Q][l];
        const Row cu = R[Q][l];
        Real sum = ((cu.qy + cu.cx) + (cu.xv + cu.ar)) + ((cu.cy + cu.yv) + (mc_.du + mc_.dv)) + (mc_.ra + mc_.qa);
        if constexpr (WIND) sum = sum + ((Ou[Q][l] + Odx[Q][l]) + (Okf[Q][l] + Ov[Q][l]) + Ody[Q][l]);
        if constexpr (HEAT) sum = sum + ((Hvd[Q][l] + Hrx[Q][l]) + (Hry[Q][l] + Hrs[Q][l]) + (Hcs[Q][l] + Hdp[Q][l]) + (Hhs[Q][l] + Hho[Q][l]));
        {
          const int r1 = GEN ? (r + 1 < r_end ? r + 1 : r_end) : r + 1, rn = GEN ? (r + 2 < r_end ? r + 2 : r_end) : r + 2;
          load_opt(Q1, r1, l, gen_tag);
          MN[Q1][l] = load_met(r1, l);
          R[Q2][l] = load_row(rn, l, gen_tag);
        }
        if constexpr (WIND) {
          if (fx_row && own_x[l]) {
            const unsigned p = pcolB[l] + (unsigned)jr * rowB;
            if constexpr (!HEAT) {
              FV3_MARCH_ST(*fv3_at(dvb, p), sum);
              FV3_MARCH_ST(*fv3_at(vpb, p), sum);
            }
            FV3_MARCH_ST(*fv3_at(vb, p), sum);
          }
          if (fy_row && own_y[l]) {
            const unsigned p = pcolB[l] + (unsigned)jf * rowB;
            if constexpr (!HEAT) {
              FV3_MARCH_ST(*fv3_at(dub, p), sum);
              FV3_MARCH_ST(*fv3_at(upb, p), sum);
            }
            FV3_MARCH_ST(*fv3_at(ub, p), sum);
          }
          if constexpr (HEAT) {
            if (fx_row && own_y[l]) FV3_MARCH_ST(*fv3_at(hob, pcolB[l] + (unsigned)jr * rowB), sum);
          }
        } else {
          if (fx_row && own_y[l]) FV3_MARCH_ST(*fv3_at(outb, pcolB[l] + (unsigned)jr * rowB), sum);
        }
      }
      if (PX_ABL == 1) return;
#endif
      // ---- phase 1: requests of the next steps; inner y flux at face r-2, q_i at row r-3; del-n chain, own-lane part
      FV3_LANES(blk_, lane, l) {
        const Met mc_ = MN[Q][l];
        {
#if PX_ABL == 2
          PX_KEEP(Ou[Q1][l]); PX_KEEP(Odx[Q1][l]); PX_KEEP(Okf[Q1][l]); PX_KEEP(Ov[Q1][l]); PX_KEEP(Ody[Q1][l]);
          PX_KEEP(Hvd[Q1][l]); PX_KEEP(Hrx[Q1][l]); PX_KEEP(Hry[Q1][l]); PX_KEEP(Hrs[Q1][l]); PX_KEEP(Hcs[Q1][l]); PX_KEEP(Hdp[Q1][l]); PX_KEEP(Hhs[Q1][l]); PX_KEEP(Hho[Q1][l]);
          PX_KEEP(MN[Q1][l].du); PX_KEEP(MN[Q1][l].dv); PX_KEEP(MN[Q1][l].ra); PX_KEEP(MN[Q1][l].qa);
          PX_KEEP(R[Q2][l].qy); PX_KEEP(R[Q2][l].cx); PX_KEEP(R[Q2][l].xv); PX_KEEP(R[Q2][l].ar); PX_KEEP(R[Q2][l].cy); PX_KEEP(R[Q2][l].yv);
#else
          const int r1 = GEN ? (r + 1 < r_end ? r + 1 : r_end) : r + 1, rn = GEN ? (r + 2 < r_end ? r + 2 : r_end) : r + 2;
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
        const Real qraw = px_move(cu.qy);  // (the del-n chain runs on the field itself: no added term)
        Real qy = WIND ? qraw + mc_.qa : qraw, qx = qy;
        if (GEN && halo_cols && (r < 1 || r > ny)) {  // the two sweeps see the cube-corner cells through different remaps (rare, not prefetched)
          int lane_o = lane;
          FV3_LAUNDER(lane_o);
          const int i = i0 - 3 + lane_o, ic = i < ied ? i : ied;
          const int rc = r < jed ? r : jed;  // (trailing steps past the last row)
          const unsigned iy = cc_index<2>(*gp, fl, ic, rc), ix = cc_index<1>(*gp, fl, ic, rc);
          qy = qb[iy];
          qx = qb[ix];
          if constexpr (WIND) {
            qy = qy + addb[iy];
            qx = qx + addb[ix];
          }
          FV3_LANDED(qy);
          FV3_LANDED(qx);
        }
        {
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
        const Real a_ = w2[l], b_ = w3[l], c_ = w4[l], d_ = qy;  // q of rows r-3 .. r
        Real al_new;
        if (y_edge) {
          const Real *mmb = (const Real *)gdya + m2;
          auto My = [&](int s_) { return px_ld(mmb, pcolB[l] + (unsigned)s_ * rowB); };
          al_new = ppm_al_win(a_, b_, c_, d_, My, sy, S, N, npy);
          FV3_LANDED(al_new);
        } else {
          al_new = PPM_P1 * (b_ + c_) + PPM_P2 * (a_ + d_);
        }
        const PpmCell co = ppm_cell(al_q[l], al_new, b_, SX_ORD);
        al_q[l] = al_new;
        const Real fyi = ppm_face(cq[l], co, cu.cy);
        fyin[l] = fyi;
        cq[l] = co;
        const Real pn = yv * fyi;
        const Real den_y = ar3 + y_prev[l] - yv;
        const Real qi = px_quot(a_ * ar3 + p_prev[l] - pn, den_y, px_rcp(den_y));
        p_prev[l] = pn;
        h_ypp[l] = y_prev[l];
        y_prev[l] = yv;
        sqx[l] = qx;
        sqi[l] = qi;
        h_q5[l] = qy;
      }
      PX_FENCE();
      // ---- phase 2: inner x flux on row r, outer x flux on row r-3, fx of row r-3; del-n chain, x fluxes (2a / 2b / 2c: see fv3_tp4x.hip)
      FV3_LANES(blk_, lane, l) {
        s_al[0][l] = PPM_P1 * (FV3_LANE_SHR(1, sqx, l, lane) + sqx[l]) + PPM_P2 * (FV3_LANE_SHR(2, sqx, l, lane) + FV3_LANE_SHL(1, sqx, l, lane));
        s_al[1][l] = PPM_P1 * (FV3_LANE_SHR(1, sqi, l, lane) + sqi[l]) + PPM_P2 * (FV3_LANE_SHR(2, sqi, l, lane) + FV3_LANE_SHL(1, sqi, l, lane));
        if (xe) {
          s_al[0][l] = px_al_edge(s_al[0][l], akind[l], FV3_LANE_SHR(2, sqx, l, lane), FV3_LANE_SHR(1, sqx, l, lane), sqx[l], FV3_LANE_SHL(1, sqx, l, lane),
                                  FV3_LANE_SHR(2, h_mx0, l, lane), FV3_LANE_SHR(1, h_mx0, l, lane), h_mx0[l], FV3_LANE_SHL(1, h_mx0, l, lane));
          s_al[1][l] = px_al_edge(s_al[1][l], akind[l], FV3_LANE_SHR(2, sqi, l, lane), FV3_LANE_SHR(1, sqi, l, lane), sqi[l], FV3_LANE_SHL(1, sqi, l, lane),
                                  FV3_LANE_SHR(2, h_mx3, l, lane), FV3_LANE_SHR(1, h_mx3, l, lane), h_mx3[l], FV3_LANE_SHL(1, h_mx3, l, lane));
        }
      }
      FV3_LANES(blk_, lane, l) {
        const PpmCell ci = ppm_cell(s_al[0][l], FV3_LANE_SHL(1, s_al[0], l, lane), sqx[l], SX_ORD);
        const PpmCell co = ppm_cell(s_al[1][l], FV3_LANE_SHL(1, s_al[1], l, lane), sqi[l], SX_ORD);
        s_bl[0][l] = ci.bl;
        s_br[0][l] = ci.br;
        s_sm[0][l] = ci.sm;
        s_bl[1][l] = co.bl;
        s_br[1][l] = co.br;
        s_sm[1][l] = co.sm;
      }
      FV3_LANES(blk_, lane, l) {
        const Row cu = R[Q][l];
        const Real cx = cu.cx, xv = cu.xv;
        const Real cx3 = RG(RG_CX, Q)[lane], xv3 = RG(RG_XV, Q)[lane], fi3 = RG(RG_FI, Q)[lane];
        const Real dv0 = RG(RG_DV, Q)[lane], dv1 = RG(RG_DV, Q2)[lane], dv2 = RG(RG_DV, Q1)[lane];
        const Real e0 = FV3_LANE_SHR(1, sd0, l, lane), e1 = FV3_LANE_SHR(1, sd1, l, lane), e2 = FV3_LANE_SHR(1, sd2, l, lane);
        gx0[l] = dv0 * (e0 - sd0[l]);
        gx1[l] = dv1 * (sd1[l] - e1);
        Real ox = dxd[l], oy = dyf[l];  // x flux of (i, r-3), y flux of face (i, r-2)
        dxd[l] = dv2 * (sd2[l] - e2);
        if (GEN && pz) {
          int lane_o = lane;
          FV3_LAUNDER(lane_o);
          const int i = i0 - 3 + lane_o;
          if (jr >= 1 && jr <= ny && on_patch(i, jr)) ox = px_ld(A.pfx + b, pcolB[l] + (unsigned)jr * rowB);
          if (i <= nx && on_patch(i, jf)) oy = px_ld(A.pfy + b, pcolB[l] + (unsigned)jf * rowB);
          FV3_LANDED(ox);
          FV3_LANDED(oy);
        }
        h_ox[l] = ox;
        h_oy[l] = oy;
        if constexpr (AREA) {
          zxo[l] = ox;
          h_zy0[l] = zyp[l];
        }
        if constexpr (WIND && !HEAT) {  // the vorticity-damping increments of v (row r-3) / u (face r-2); the damping-heat kernel reads them again: stored
          if (fx_row && own_x[l]) FV3_MARCH_ST(*fv3_at(dvb, pcolB[l] + (unsigned)jr * rowB), ox);
          if (fy_row && own_y[l]) FV3_MARCH_ST(*fv3_at(dub, pcolB[l] + (unsigned)jf * rowB), oy);
        }
        Real ff[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const Real qv = w == 0 ? sqx[l] : sqi[l];
          const PpmCell cm{FV3_LANE_SHR(1, s_bl[w], l, lane), FV3_LANE_SHR(1, s_br[w], l, lane), w == 0 ? FV3_LANE_SHR(1, sqx, l, lane) : FV3_LANE_SHR(1, sqi, l, lane),
                           FV3_LANE_SHR1_FLAG(s_sm[w], l, lane)};
          const PpmCell c0{s_bl[w][l], s_br[w][l], qv, s_sm[w][l]};
          ff[w] = ppm_face(cm, c0, w == 0 ? cx : cx3);
        }
        const Real fxin = ff[0], fxout = ff[1];
        const Real v = (Real)0.5 * (fxout + fi3) * xv3;
        if constexpr (WIND) {
          const Real vn = Ov[Q][l] * Ody[Q][l] + wkr[l] - Okf[Q][l] - v;
          if (fx_row && own_x[l]) {
            const unsigned p = pcolB[l] + (unsigned)jr * rowB;
            if constexpr (!HEAT) FV3_MARCH_ST(*fv3_at(vpb, p), vn);
            FV3_MARCH_ST(*fv3_at(vb, p), vn - ox);
          }
          if constexpr (HEAT) {  // vb / fx of the v face (i, r-3): damping field of corner rows r-3 (previous step's) and r-2
            const Real rdy0 = Hry[Q][l];
            s_vb[l] = (vdp[l] - Hvd[Q][l] - ox) * rdy0;
            s_fx[l] = vn * rdy0;
          }
        } else {
          fxk[l] = v;
          xjr[l] = xv3;
        }
        RG(RG_FI, Q)[lane] = fxin;
        RG(RG_CX, Q)[lane] = cx;
        RG(RG_XV, Q)[lane] = xv;
        px[l] = xv * fxin;
        sxv[l] = xv;
      }
      PX_FENCE();
      // ---- phase 3: q_j on row r, outer y flux at face r-2, fy of face r-2; the epilogue
      FV3_LANES(blk_, lane, l) {
        const Row cu = R[Q][l];
        const Real p1 = FV3_LANE_SHL(1, px, l, lane), x1 = FV3_LANE_SHL(1, sxv, l, lane);
        const Real ar = cu.ar;
        const Real den_x = ar + cu.xv - x1;
        const Real qj = px_quot(h_q5[l] * ar + px[l] - p1, den_x, px_rcp(den_x));
        const Real a_ = v2[l], b_ = v3[l], c_ = v4[l], d_ = qj;
        Real al_new;
        if (y_edge) {
          const Real *mmb = (const Real *)gdya + m2;
          auto My = [&](int s_) { return px_ld(mmb, pcolB[l] + (unsigned)s_ * rowB); };
          al_new = ppm_al_win(a_, b_, c_, d_, My, sy, S, N, npy);
          FV3_LANDED(al_new);
        } else {
          al_new = PPM_P1 * (b_ + c_) + PPM_P2 * (a_ + d_);
        }
        const PpmCell co = ppm_cell(al_v[l], al_new, b_, SX_ORD);
        al_v[l] = al_new;
        const Real fyout = ppm_face(cv[l], co, cu.cy);
        cv[l] = co;
        v2[l] = b_;
        v3[l] = c_;
        v4[l] = d_;
        const Real v = (Real)0.5 * (fyout + fyin[l]) * cu.yv;
        if constexpr (WIND) {
          const Real ke_e = FV3_LANE_SHL(1, Okf[Q], l, lane);  // ke(i + 1, jf): the neighbouring lane loaded it as its ke(i, jf)
          const Real un = Ou[Q][l] * Odx[Q][l] + Okf[Q][l] - ke_e + v;
          if (fy_row && own_y[l]) {
            const unsigned p = pcolB[l] + (unsigned)jf * rowB;
            if constexpr (!HEAT) FV3_MARCH_ST(*fv3_at(upb, p), un);
            FV3_MARCH_ST(*fv3_at(ub, p), un + h_oy[l]);
          }
          wkr[l] = Okf[Q][l];  // ke(i, jr) of the next step = this step's ke(i, jf)
          if constexpr (HEAT) {
            // the cell (i, r-3): u faces r-3 (ub0 / fy0: the previous step's) and r-2, v faces i (the lane's) and i + 1 (the next lane's)
            const Real vd01 = Hvd[Q][l], vd11 = FV3_LANE_SHL(1, Hvd[Q], l, lane), rdx1 = Hrx[Q][l];
            const Real ub1 = (vd01 - vd11 + h_oy[l]) * rdx1, fy1 = un * rdx1;
            const Real ub0 = ubp[l], fy0 = fyq[l];
            const Real vb0 = s_vb[l], vb1 = FV3_LANE_SHL(1, s_vb, l, lane), fx0 = s_fx[l], fx1 = FV3_LANE_SHL(1, s_fx, l, lane);
            Real hs = Hhs[Q][l];
            if (dcon > (Real)1.0e-5) {
              const Real gy0 = fy0 * ub0, gy1 = fy1 * ub1, gx0_ = fx0 * vb0, gx1_ = fx1 * vb1;
              const Real u2 = fy0 + fy1, du2 = ub0 + ub1, v2_ = fx0 + fx1, dv2 = vb0 + vb1;
              hs = Hdp[Q][l] * (hs - (Real)0.25 * dcon * Hrs[Q][l] *
                                         ((ub0 * ub0 + ub1 * ub1 + vb0 * vb0 + vb1 * vb1) + (Real)2.0 * (gy0 + gy1 + gx0_ + gx1_) - Hcs[Q][l] * (u2 * dv2 + v2_ * du2 + du2 * dv2)));
            }
            if (fx_row && own_y[l]) FV3_MARCH_ST(*fv3_at(hob, pcolB[l] + (unsigned)jr * rowB), Hho[Q][l] + hs);
            ubp[l] = ub1;
            fyq[l] = fy1;
            vdp[l] = px_move(vd01);
          }
        } else {
          const Real fxe = FV3_LANE_SHL(1, fxk, l, lane), xje = FV3_LANE_SHL(1, xjr, l, lane), zx1 = FV3_LANE_SHL(1, zxo, l, lane);
          const Real qc = w2[l];  // q(i, r-3)
          const Real ar_ = RG(RG_AR, Q)[lane];
          const Real ra_x = ar_ + xjr[l] - xje, ra_y = ar_ + h_ypp[l] - cu.yv;
          const Real den = ra_x + ra_y - ar_;
          Real z = px_quot(qc * ar_ + fxk[l] - fxe + fyp[l] - v, den, px_rcp(den));
          z = z + (h_ox[l] - zx1 + h_zy0[l] - h_oy[l]) * h_era[l];
          if (fx_row && own_y[l]) FV3_MARCH_ST(*fv3_at(outb, pcolB[l] + (unsigned)jr * rowB), z);
          fyp[l] = v;
          zyp[l] = h_oy[l];
        }
        w2[l] = w3[l];
        w3[l] = w4[l];
        w4[l] = h_q5[l];
        RG(RG_AR, Q)[lane] = ar;
      }
      PX_FENCE();
    };

    int r_hi = cb + 3 < jb + 2 ? cb + 3 : jb + 2;
    if (r_hi > r_end - 2) r_hi = r_end - 2;
    if (N && r_hi > ny) r_hi = ny;
    if (pz && r_hi > ny - FV3_D6_PATCH + 2) r_hi = ny - FV3_D6_PATCH + 2;  // (the first step that consumes a face of a N corner patch: r - 2 = ny + 2 - PATCH)
    const int head = pz && r0 <= FV3_D6_PATCH + 3 ? 15 : 6;               // (... of a S corner patch: r - 3 <= PATCH)
    int r = r0;
#pragma clang loop unroll(disable)
    for (int part = 0; part < 2; ++part) {
      const int stop = part == 0 ? r0 + head - 1 : r_end;
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
  });
}

// The winds on the faces between segments / strips, copied before the HEAT march updates the winds in place: u on the first face row of every
// segment but the first, v on the first face column of every strip but the first (the geometry of single_march_t's launch).
// (seg_nk: the level count the march itself is launched on -- its segment length depends on it; the copies may serve fewer levels)
void sx_side_copy(fv3_ctx *c, fv3_stream_t s, const Real *u, const Real *v, Real *u_side, Real *v_side, int k_lo, int k_hi, int seg_nk) {
  const Geo g = c->g;
  const int nk = k_hi - k_lo + 1;
  if (nk <= 0) return;
  const int nstrip = (g.nx + 1 + SX_OUT - 1) / SX_OUT;
  const int seg = sx_march_seg(c, seg_nk);
  const int nseg = (g.ny + seg - 1) / seg;
  launch3(c, s, Box{1, g.nx, 1, nseg - 1, k_lo, k_hi}, [=] FV3_HD(int t, int k, int i, int js) {
    const long p = t * g.st + k * g.sk + IX(i, 1 + js * seg);
    u_side[p] = u[p];
  });
  launch3(c, s, Box{1, g.ny, 1, nstrip - 1, k_lo, k_hi}, [=] FV3_HD(int t, int k, int j, int bs) {
    const long p = t * g.st + k * g.sk + IX(1 + bs * SX_OUT, j);
    v_side[p] = v[p];
  });
}

}  // namespace

int sx_march_seg(const fv3_ctx *c, int nk) {
  const Geo &g = c->g;
  const int nstrip = (g.nx + 1 + SX_OUT - 1) / SX_OUT;
  return fv3_pick_seg((long)nstrip * ((g.ny + 63) / 64) * g.nsub * nk, 2);
}

// kind 1: the vorticity transport of d_sw with the wind update (epi: wind_u / wind_v / wind_ke / wind_du / wind_dv / wind_u_pre / wind_v_pre, fd_coef,
// fd_add); kind 2: the interface-height transport of update_dz_d (epi: out, fd_coef).  Levels k0 .. k1 all run their del-n chain inside the march.
void tp2d_single_march(fv3_ctx *c, fv3_stream_t s, int kind, const Real *q, const Real *crx, const Real *cry, const Real *xfx, const Real *yfx, int k0, int k1,
                       const TpEpi *epi, const TpHeat *heat) {
  SxArgs a;
  memset(&a, 0, sizeof(a));
  a.q = q;
  a.crx = crx;
  a.cry = cry;
  a.xfx = xfx;
  a.yfx = yfx;
  a.coef = epi->fd_coef;
  if (kind == SX_WIND) {
    a.u = epi->wind_u;
    a.v = epi->wind_v;
    a.u_pre = epi->wind_u_pre;
    a.v_pre = epi->wind_v_pre;
    a.du = const_cast<Real *>(epi->wind_du);
    a.dv = const_cast<Real *>(epi->wind_dv);
    a.ke = epi->wind_ke;
    a.add = epi->fd_add;
    a.pfx = epi->wind_dv;
    a.pfy = epi->wind_du;
    if (heat && heat->vdamp) {
      a.vdamp = heat->vdamp;
      a.ndelp = heat->ndelp;
      a.heat_s = heat->heat_s;
      a.heat_src = heat->heat_src;
      a.heat_zeros = heat->zeros;
      a.dcon = heat->dcon;
      a.u_side = epi->wind_u_pre;  // (the arrays the epilogue makes redundant serve as the side copies)
      a.v_side = epi->wind_v_pre;
      sx_side_copy(c, s, epi->wind_u, epi->wind_v, epi->wind_u_pre, epi->wind_v_pre, k0, k1 < heat->side_from - 1 ? k1 : heat->side_from - 1, k1 - k0 + 1);
      single_march_t<SX_WIND, true>(c, s, a, k0, k1);
      heat->consumed = true;
    } else {
      single_march_t<SX_WIND>(c, s, a, k0, k1);
    }
  } else {
    a.out = epi->out;
    a.pfx = epi->zfx;
    a.pfy = epi->zfy;
    single_march_t<SX_AREA>(c, s, a, k0, k1);
  }
}
