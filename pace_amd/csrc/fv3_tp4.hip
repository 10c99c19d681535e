// fv3_tp4.hip -- d_sw's four scalar transports (air mass delp, vertical velocity w, condensate q_con, potential
// temperature pt) as multi-tracer marching wave kernels.  CPU twin: oracle/fv3_oracle/d_sw.py (the four fv_tp_2d calls
// and the divisions by the new air mass of d_sw_levels).  [SURVEY A.3.2 - A.3.4, A.4]
//
// Why: as four launches of tp2d_stream_t the transports re-read the Courant numbers, the area fluxes and the cell areas
// four times, the air-mass fluxes go out to memory to come back three times as the mass fluxes of the other tracers, the
// flux-form updates go out as delp * q fields and a fifth kernel divides them by the new air mass: 141 GB + 29 GB of the
// 352 GB one d_sw call moved at C768 (PMC, round 1) for 8 fields of algorithmic traffic.  Here a wave carries several
// tracers of its strip together:
//   * crx / cry / xfx / yfx / area are loaded once per row for the tracers of the wave (and the two denominators of the
//     cross-direction updates are shared);
//   * the final air-mass flux of a face is used in the same lane, in the same step, as the mass flux of w
//     (a pointwise dependency);
//   * the old air mass is the damping weight and the epilogue multiplier of q_con / pt;
//   * the epilogue of a cell forms delp_new, (delp * q + div) / delp_new, w's del-n increment and the heat it
//     dissipates -- the post-transport kernel of round 1 is gone.
// Instantiations of the same step (ROLE):
//   QUAD  all four tracers in one wave: least traffic (28 field passes), but ~500 registers = one wave per SIMD with half
//         of the state parked in AGPRs: measured 39.4 ms at C768 (48 % of the wave cycles issuing, 46 % waiting) against
//         33.2 ms for the five round-1 launches -- kept as an experiment (FV3_DSW_SCALARS=quad);
//   AIR   delp + w (w rides on the air-mass flux of the same lane); the air-mass fluxes are also stored for TRC;
//   TRC   q_con + pt on the stored air-mass fluxes, the old air mass as damping weight, the new one as divisor.
// and (PART) of the way the cube-tile edges are served:
//   ALL       every strip, the strips that touch a W / E tile edge with the one-sided PPM formulas among their faces
//             evaluated per lane (divergent: 3.2 x the instructions of an interior strip, +130 registers);
//   INTERIOR  every strip with the interior formulas only; the faces / cells the W / E one-sided formulas reach
//             (x-faces 1..3 / nx-1..nx+1, cells and y-faces of columns 1..3 / nx-2..nx) are masked out;
//   EDGE      those columns, by a TRANSPOSED march: lanes along j, marching ~9 steps along i across the tile edge.  The
//             W / E one-sided formulas are then the march-direction edge, i.e. wave-uniform branches (what the S / N edges
//             are for the normal march), and fv_tp_2d is symmetric in x and y, so it is the same step with the roles of
//             (crx, xfx, dxa, fx, W / E, nx) and (cry, yfx, dya, fy, S / N, ny) exchanged and the two strides swapped.
// AIR + TRC, INTERIOR + EDGE (the default) = four launches: two-tracer interior waves fit 212 registers (two waves per
// SIMD, no spill; compiled with the per-lane edge path they needed 397) and run at HBM speed (9.0 ms each, 5.1 / 3.8 TB/s),
// the edge launches are ~2 % of the interior work.
// The arithmetic of each tracer is expression for expression the one of tp2d_stream_t (same operation order: bitwise
// equal results, checked by tests/test_parity.py::test_fused_scalar_march_is_bitwise_the_four_transports and the A/B
// switch FV3_DSW_SCALARS=separate).
// FD (round 3): the del-n damping chains of the wave's tracers run INSIDE the march (the reference's DelnFlux, restated as the
// marching pipeline of del6_stream in fv3_tp2d.hip: iteration s of the chain runs s rows behind the row being consumed, its
// d2 / flux values live in registers, the i-neighbours come from wavefront shuffles -- LDS lines in the transposed tile-edge
// march).  The damping fluxes of delp / w / q_con / pt then never exist as fields: four del6_stream launches (3.7 GB read,
// 4.7 GB written each at C768) and eight field reads of the two marches are gone.  What makes room for the pipeline at two
// waves per SIMD: the wave's pure delay lines (Courant number, area flux, cell area, inner flux of rows r-1 .. r-3) and the
// three metric rows of the chain are parked in a per-wave LDS ring (own-lane traffic: no ordering point).  Faces whose
// dependency cone touches a cube corner still come from the staged chain on the 8 x 8 corner patches (del6_vt_flux_patches)
// and are read from the flux fields there.  Levels whose chains are not all of order 2 (the sponge layers) take the form
// without FD.  Same expressions in the same order as del6_stream: bitwise equal (FV3_DSW_DELN=arrays is the A/B switch).
#include "fv3_ops.h"
#define FV3_MARCH_ST(lhs, val) FV3_ST_NT(lhs, val)  // (the marches' outputs are streamed: fv3_common.h)
#include "fv3_ppm.h"

#include <type_traits>
#include <utility>

#define Q4_OUT 58
#define Q4_LINE (FV3_WAVE + 6)
#define Q4_PF 2
enum { Q4_QUAD = 0, Q4_AIR = 1, Q4_TRC = 2 };
enum { Q4_ALL = 0, Q4_INTERIOR = 1, Q4_EDGE = 2 };
#define Q4_EW 3  // cells next to a W / E tile edge the EDGE launch owns
#ifndef Q4_KB
#define Q4_KB FV3_Q4_KB_DEFAULT   // > 0: that many levels of one tile are consecutive workgroups of an XCD (interior marches); 0: plane-major
#endif

// tracer identity of slot n: 0 = delp, 1 = w, 2 = q_con, 3 = pt
template <int ROLE>
FV3_HD constexpr int q4_id(int n) {
  return ROLE == Q4_TRC ? n + 2 : n;
}

// compile-time loop over the tracer slots: the slot index is a constant expression in the body (a plain unrolled loop
// leaves it to the optimizer; when a body is not unrolled the per-slot register arrays turn into indexed scratch)
template <int N0, int N1, class F>
FV3_HD inline void q4_for(F &&f) {
  if constexpr (N0 < N1) {
    f(std::integral_constant<int, N0>{});
    q4_for<N0 + 1, N1>(f);
  }
}
#define Q4_EACH(n) q4_for<0, Q4_NT>([&](auto n##_c) { constexpr int n = decltype(n##_c)::value; constexpr int id = q4_id<ROLE>(n); (void)id;
#define Q4_END });

// Naming inside the kernel: L = the lane direction (x for the normal march, y for the transposed one), M = the march
// direction.  "lc" / "r" are the Fortran-local coordinates along L / M.
// M8: the hord-8 (monotone) reconstruction for every slot (tracer_2d_1l with hord_tr = 8), a separate instantiation so
// that the hord 5 / 6 kernels of d_sw carry none of it
// HC: 0 = the PPM orders are run-time values (hord_dp / hord_vt / hord_tm of the call); 5 / 6 = all three equal that constant (the
// reference configs: 6 everywhere): the limiter test of ppm_cell folds to one comparison and the order occupies no register.
template <int ROLE, int PART, bool M8 = false, bool FD = false, int HC = 0>
static void dsw_scalars_t(fv3_ctx *c, fv3_stream_t s, const DswScalars &a_, int k_lo, int k_hi) {
  static_assert(!FD || (ROLE != Q4_QUAD && PART != Q4_ALL && !M8), "the fused del-n chains exist for the two-tracer interior / edge marches");
  constexpr int Q4_NT = ROLE == Q4_QUAD ? 4 : 2;
  constexpr bool HAS_AIR = ROLE != Q4_TRC;  // slot 0 is the air mass: its flux is the mass flux of the other slots
  constexpr bool TR = PART == Q4_EDGE;      // transposed march
#ifndef FV3_Q4_WPE
// waves per SIMD the register allocation of the two-tracer interior marches is sized for: two in fp64 (250 registers), four in the fp32 build (128)
#define FV3_Q4_WPE (sizeof(Real) == 4 ? 4 : 2)
#endif
  constexpr int WPE = (ROLE == Q4_QUAD || PART == Q4_EDGE) ? 1 : FV3_Q4_WPE;
  // interior strips: every neighbour read of the step is a wavefront shuffle (DPP), the kernel uses no LDS at all
#ifdef Q4_NO_DPP
  constexpr bool DPP = false;
#else
  constexpr bool DPP = PART == Q4_INTERIOR;
#endif
  constexpr bool PARK = FD && DPP;  // delay lines in the LDS ring (the two-waves-per-SIMD kernels)
  constexpr int NRING = FD ? (PARK ? 8 : 3) : 0;
  enum { RG_DU = 0, RG_DV = 1, RG_RA = 2, RG_CX = 3, RG_XV = 4, RG_AR = 5, RG_FI = 6 };  // ring variables (RG_FI + slot)
  constexpr int FDL = FD && !DPP ? Q4_NT * 5 + 1 : 0;  // LDS lines of the chain (transposed march): d2 of iterations 0..2, fluxes of 0..1 per slot, w's final flux
  const Geo g = c->g;
  DswScalars a = a_;
  if (TR) {  // exchange the roles of the two directions
    std::swap(a.crx, a.cry);
    std::swap(a.xfx, a.yfx);
    std::swap(a.mfx, a.mfy);
    std::swap(a.fx, a.fy);
    std::swap(a.dpx, a.dpy);
    std::swap(a.dqx, a.dqy);
    std::swap(a.dtx, a.dty);
    std::swap(a.dwx, a.dwy);
  }
  const int nk = k_hi - k_lo + 1;
  if (nk <= 0) return;
  const int nL = TR ? g.ny : g.nx, nM = TR ? g.nx : g.ny;  // cells along the lanes / along the march
  const int npL = nL + 1, npM = nM + 1;
  const int nstrip = (nL + 1 + Q4_OUT - 1) / Q4_OUT;
  const int seg = TR ? 1 : fv3_pick_seg((long)nstrip * ((g.ny + 63) / 64) * g.nsub * nk, WPE);
  const int nseg = TR ? 2 : (nM + seg - 1) / seg;  // transposed: "segment" 0 = the low (W) edge columns, 1 = the high (E) ones
  // LDS: per tracer the two L-sweep row lines (q on the new row, the M-advected q three rows behind), (area flux) * (inner
  // L flux) and the final L flux; shared: the L area flux, the old air mass of row r-3, the tile-edge metric ring
  const size_t smem_lines = DPP ? 0 : (size_t)(Q4_NT * (2 * Q4_LINE + 2 * (FV3_WAVE + 1)) + 2 * (FV3_WAVE + 1) + 32);
  const size_t smem = sizeof(Real) * (smem_lines + (size_t)FDL * (FV3_WAVE + 2) + (size_t)NRING * 4 * FV3_WAVE);
  const Geo *gp = c->g_dev;
  Real *const trash = c->trash;
  const int nh = g.nh, sj32 = g.sj32, go = g.o;
  const int LS = TR ? sj32 : 1, MS = TR ? 1 : sj32;  // element strides of one step along L / along M
  const long st = g.st, sk = g.sk, st2 = g.st2;
  const MPtr area = g.area, rarea = g.rarea;
  const MPtr metL = TR ? g.dya : g.dxa;  // cell widths of the one-sided formulas along L
  const MPtr d6L = TR ? g.del6_u : g.del6_v, d6M = TR ? g.del6_v : g.del6_u;  // del-n face coefficients of the L / M faces
  const Real *damp_w_k = g.damp_w, *ke_bg_k = g.ke_bg;
  const int bitLlo = TR ? FV3_S : FV3_W, bitLhi = TR ? FV3_N : FV3_E, bitMlo = TR ? FV3_W : FV3_S, bitMhi = TR ? FV3_E : FV3_N;
  // Launch geometry.  Plane-major (default before round 3): an XCD walks the tiles of one (sub-domain, level) plane, consecutive
  // levels land on different XCDs.  Level-major (Q4_KB levels of ONE tile are consecutive workgroups of an XCD): the tile's 2-D metric
  // rows (area, rarea, the del-n coefficients) are fetched into that XCD's L2 once per Q4_KB levels instead of once per level.
  // Measured at C768 (same box, alternating runs): the same time (d_sw 54.7 / 55.05 ms level-major (16) against 55.6 / 55.06 plane-major)
  // and 14 % fewer L2 misses of the march (FETCH_SIZE 31.7 -> 24.9 GB for delp + w, 37.1 -> 30.8 for q_con + pt: the metric rows were
  // Infinity-Cache hits, not HBM reads, which is why the time does not move).  Level-major is the default; FV3_Q4_KB=0 selects
  // plane-major (A/B).
  static const int kb_env = getenv("FV3_Q4_KB") ? atoi(getenv("FV3_Q4_KB")) : Q4_KB;
  const int KB = (PART == Q4_INTERIOR && kb_env > 0) ? (kb_env < nk ? kb_env : nk) : 0;
  const int nblk = KB ? (nk + KB - 1) / KB : 0;
#if defined(FV3_STAMPS) && !defined(FV3_HOST_EMU)
  unsigned long long *const st_buf = fv3_stamp_buf();
  constexpr unsigned long long st_kid = 1000ull + 100ull * ROLE + 10ull * PART + (FD ? 1ull : 0ull) + 10000ull * HC;
#endif
  launch_waves<WPE>(c, s, KB ? KB : nstrip, KB ? nstrip * nseg : nseg, KB ? g.nsub * nblk : g.nsub * nk, smem, [=] FV3_HD(const Blk &blk_, char *smem_) {
    Blk blk = blk_;
    int t, k;
    if (KB) {
      t = blk_.bz / nblk;
      const int kk = (blk_.bz - t * nblk) * KB + blk_.bx;
      if (kk >= nk) return;
      k = k_lo + kk;
      blk.by = blk_.by / nstrip;
      blk.bx = blk_.by - blk.by * nstrip;
    } else {
      t = blk.bz / nk;
      k = k_lo + (blk.bz - t * nk);
    }
    const int fl = gp->flags[t];
    const long b = t * st + k * sk, m2 = t * st2;
    const int l0 = 1 + blk.bx * Q4_OUT;  // first owned L face / cell of the strip
    // owned M range: cells ca..cb (and the L faces of those rows), M faces fa..fb
    int ca, cb, fa, fb;
    if (TR) {
      if (!(fl & (blk.by == 0 ? bitMlo : bitMhi))) return;  // this sub-domain does not touch that tile edge
      if (blk.by == 0) {
        ca = 1, cb = Q4_EW, fa = 1, fb = Q4_EW;
      } else {
        ca = nM - Q4_EW + 1, cb = nM, fa = nM - Q4_EW + 2, fb = nM + 1;
      }
    } else {
      ca = fa = 1 + blk.by * seg;
      fb = blk.by == nseg - 1 ? nM + 1 : fa + seg - 1;
      cb = fb < nM ? fb : nM;
    }
    const int Led = nL + nh, Msd = 1 - nh, Med = nM + nh;
    Real *lq[Q4_NT], *lqi[Q4_NT], *exp_[Q4_NT], *exf[Q4_NT];
    {
      Real *p = (Real *)smem_;
      Q4_EACH(n)
        lq[n] = p;
        lqi[n] = p + Q4_LINE;
        exp_[n] = p + 2 * Q4_LINE;
        exf[n] = p + 2 * Q4_LINE + FV3_WAVE + 1;
        p += 2 * Q4_LINE + 2 * (FV3_WAVE + 1);
      Q4_END
    }
    Real *exx = (Real *)smem_ + Q4_NT * (2 * Q4_LINE + 2 * (FV3_WAVE + 1));  // L area flux of the lane's face (read by lane - 1)
    Real *exm = exx + FV3_WAVE + 1;                                          // old delp(lc, r-3) of the lane (read by lane + 1)
    Real *emr = exm + FV3_WAVE + 1;                                          // tile-edge strips: metric ring (4 rows x 8 cells)
    // FD: the chain's lines (index lane + 1: lane - 1 .. lane + 1 readable) and the own-lane ring [variable][row & 3][lane]
    Real *fdl = (Real *)smem_ + smem_lines + 1;
    Real *ring = (Real *)smem_ + smem_lines + (size_t)FDL * (FV3_WAVE + 2);
    auto LD = [&](int n, int it) -> Real * { return fdl + (n * 5 + it) * (FV3_WAVE + 2); };       // d2 of iteration it (0..2), slot n
    auto LF = [&](int n, int it) -> Real * { return fdl + (n * 5 + 3 + it) * (FV3_WAVE + 2); };   // L flux of iteration it (0..1)
    Real *lzx = fdl + Q4_NT * 5 * (FV3_WAVE + 2);                                                 // w: final L flux of row r-3
    auto RG = [&](int var, int r) -> Real * { return ring + (var * 4 + (r & 3)) * FV3_WAVE; };
    const Real *const qall[4] = {a.delp + b, a.w + b, a.q_con + b, a.pt + b};
    const int hall[4] = {a.hord_dp, a.hord_vt, a.hord_dp, a.hord_tm};
    const Real *qin[Q4_NT];
    int hord[Q4_NT];
    Q4_EACH(n)
      qin[n] = qall[id];
      hord[n] = HC ? HC : hall[id];
    Q4_END
    const Real *crLb = a.crx + b, *crMb = a.cry + b, *afLb = a.xfx + b, *afMb = a.yfx + b;  // Courant numbers / area fluxes along L, M
    const MPtr areab = area + m2;
    const bool Llo = (fl & bitLlo) && l0 <= 3, Lhi = (fl & bitLhi) && l0 + Q4_OUT + 1 >= npL - 1;
    const bool Mlo = fl & bitMlo, Mhi = fl & bitMhi;
    const bool halo_cols = l0 - 3 < 1 || l0 + FV3_WAVE - 4 > nL;
    // (FD launches cover only levels on which all three chains are switched on -- fd_k0 in fv3_d_sw_out --: the flags are constants there)
    const bool on_vt = FD ? true : deln_on(a.dn_vt, k), on_t = FD ? true : deln_on(a.dn_t, k);
    const Real damp_vt = on_vt ? deln_damp(a.dn_vt, k) : (Real)0, damp_t = on_t ? deln_damp(a.dn_t, k) : (Real)0;
    const bool on_w = FD ? true : damp_w_k[k] > (Real)1.0e-5;
    // FD: d2 of iteration 0 = coef * q for delp / w (the damped quantity itself), q for q_con / pt (weighted by the air mass later)
    Real dcoef[Q4_NT];
    Q4_EACH(n)
      dcoef[n] = !FD ? (Real)0 : id == 0 ? deln_damp(a.dn_vt, k) : id == 1 ? deln_damp(a.dn_w, k) : (Real)1;
    Q4_END
    // cube corners of this sub-domain: the faces on their 8 x 8 patches come from the flux fields (staged chain)
    const bool c_ll = (fl & (bitLlo | bitMlo)) == (bitLlo | bitMlo), c_hl = (fl & (bitLhi | bitMlo)) == (bitLhi | bitMlo);
    const bool c_hh = (fl & (bitLhi | bitMhi)) == (bitLhi | bitMhi), c_lh = (fl & (bitLlo | bitMhi)) == (bitLlo | bitMhi);
    const bool pz_lo = FD && (c_ll || c_lh) && l0 - 3 <= FV3_D6_PATCH, pz_hi = FD && (c_hl || c_hh) && l0 + FV3_WAVE - 4 >= nL + 2 - FV3_D6_PATCH;
    // is the L face / cell column lc at M coordinate m on a corner patch?  (patch = faces 1..P resp. n+2-P..n+1 of both directions)
    auto on_patch = [&](int lc, int m) -> bool {
      const bool llo = lc >= 1 && lc <= FV3_D6_PATCH, lhi = lc >= nL + 2 - FV3_D6_PATCH && lc >= 1 && lc <= nL + 1;
      const bool mlo = m >= 1 && m <= FV3_D6_PATCH, mhi = m >= nM + 2 - FV3_D6_PATCH && m >= 1 && m <= nM + 1;
      return (llo && mlo && c_ll) || (lhi && mlo && c_hl) || (lhi && mhi && c_hh) || (llo && mhi && c_lh);
    };
    const Real *const dLall[4] = {a.dpx ? a.dpx + b : nullptr, a.dwx ? a.dwx + b : nullptr, a.dqx ? a.dqx + b : nullptr, a.dtx ? a.dtx + b : nullptr};
    const Real *const dMall[4] = {a.dpy ? a.dpy + b : nullptr, a.dwy ? a.dwy + b : nullptr, a.dqy ? a.dqy + b : nullptr, a.dty ? a.dty + b : nullptr};
    const Real dd8 = ke_bg_k[k] * fabs(a.dt);
    const int r_end = fb + 3 < Med ? fb + 3 : Med;

    struct Row {
      Real qy[Q4_NT], cx, xv, ar, cy, yv, em;
    };
    // inputs consumed at the end of a step, loaded at its top ahead of the prefetch (loads return in order: waiting for
    // them leaves the prefetched rows in flight)
    Real o_ax[FV3_LPT], o_ay[FV3_LPT];                  // accumulated mass fluxes (AIR / QUAD)
    Real o_mx[FV3_LPT], o_my[FV3_LPT], o_mc[FV3_LPT], o_dn[FV3_LPT], mbk[FV3_LPT];  // TRC: air-mass fluxes, old delp(lc, r-2) / (lc, r-3), new delp(lc, r-3)
    // ... of the NEXT step.  These inputs are consumed at the end of a step; requested at the top of the same step the wait for
    // them (one in-order memory counter, and the compiler orders the loads of a step as it likes) drained the row prefetch in
    // phase 2 of every step.  Requested one step ahead they are covered by the wait at the top of the step, like the rows.
    Real n_ax[FV3_LPT], n_ay[FV3_LPT], n_mx[FV3_LPT], n_my[FV3_LPT], n_mc[FV3_LPT], n_dn[FV3_LPT];
    // TRC, interior strips: the new air mass of a cell is RECOMPUTED from the air-mass fluxes the wave holds anyway -- old delp + (fx - fx[i+1] +
    // fy[face r-3] - fy[face r-2]) * rarea, the very expression (and operands) of the delp + w march that stored it -- instead of being read back:
    // one field stream less (2.75 GB per call at C768).  fyp_air = the air-mass M flux of the previous step's face.  FV3_NO_DN_RECOMP: the load (A/B).
#ifdef FV3_NO_DN_RECOMP
    constexpr bool DN_RECOMP = false;
#else
    constexpr bool DN_RECOMP = !HAS_AIR && DPP;
#endif
    Real fyp_air[FV3_LPT];
    Real o_dx[Q4_NT][FV3_LPT], o_dy[Q4_NT][FV3_LPT];    // damping fluxes along L / M (w's enter as an increment instead)
    Real era[FV3_LPT];                                  // rarea(lc, r-3)
    Real zx0[FV3_LPT], zx1[FV3_LPT], zy0[FV3_LPT], zy1[FV3_LPT];  // w's damping fluxes around the cell (lc, r-3): x, x + 1, y, y + 1
    Row nxt[FV3_LPT], nx2[FV3_LPT], cur[FV3_LPT];
    Real a1[FV3_LPT], a2[FV3_LPT], a3[FV3_LPT];
    Real w2[Q4_NT][FV3_LPT], w3[Q4_NT][FV3_LPT], w4[Q4_NT][FV3_LPT], w5[Q4_NT][FV3_LPT], al_q[Q4_NT][FV3_LPT];
    Real v2[Q4_NT][FV3_LPT], v3[Q4_NT][FV3_LPT], v4[Q4_NT][FV3_LPT], v5[Q4_NT][FV3_LPT], al_v[Q4_NT][FV3_LPT];
    PpmCell cq[Q4_NT][FV3_LPT], cv[Q4_NT][FV3_LPT];
    Real p_prev[Q4_NT][FV3_LPT], y_prev[FV3_LPT];
    Real fi1[Q4_NT][FV3_LPT], fi2[Q4_NT][FV3_LPT], fi3[Q4_NT][FV3_LPT];
    Real cx1[FV3_LPT], cx2[FV3_LPT], cx3[FV3_LPT], xv1[FV3_LPT], xv2[FV3_LPT], xv3[FV3_LPT];
    Real fyin[Q4_NT][FV3_LPT], px[Q4_NT][FV3_LPT];
    Real fxk[Q4_NT][FV3_LPT], fyp[Q4_NT][FV3_LPT];  // L flux of the low L face / M flux of the low M face of cell (lc, r-3)
    Real sqx[Q4_NT][FV3_LPT], sqi[Q4_NT][FV3_LPT], smb[FV3_LPT], sxv[FV3_LPT];  // DPP form: the values the neighbouring lanes read (q on the new row, the M-advected q, old delp, L area flux)
    unsigned pcol[FV3_LPT];  // in-plane offset of (lc, M coordinate 0)
    bool own_x[FV3_LPT], own_y[FV3_LPT];
    // (-DFV3_USTORE experiment builds: on the interior strips every store of a step is issued, unowned lanes / rows write to the
    //  wave's row of the sink -- fv3_store_sel in fv3_common.h has the measurement)
#ifdef FV3_USTORE
    constexpr bool UST = PART == Q4_INTERIOR;
#else
    constexpr bool UST = false;
#endif
    Real *const sink0 = trash + (size_t)(((unsigned)blk_.bx + 61u * (unsigned)blk_.by + 127u * (unsigned)blk_.bz) & (FV3_TRASH_SLOTS - 1)) * FV3_WAVE;
    // FD: the chain's state.  sd0 / sd1 / sd2: d2 of iterations 0 / 1 / 2 on rows r / r-1 / r-2 (read by lane + 1, and at the next
    // step as the row below); gx0 / gx1: L fluxes of iterations 0 / 1 on rows r / r-1 (read by lane - 1 at the next step);
    // gy0 / gy1: their M fluxes at faces r / r-1; dxd: the final L flux of row r-3; dyf: the final M flux of face r-2 (this step);
    // zyp: w's final M flux of face r-3; zxo: w's final L flux of row r-3 after the patch override (read by lane - 1)
    Real sd0[Q4_NT][FV3_LPT], sd1[Q4_NT][FV3_LPT], sd2[Q4_NT][FV3_LPT], gx0[Q4_NT][FV3_LPT], gx1[Q4_NT][FV3_LPT], gy0[Q4_NT][FV3_LPT], gy1[Q4_NT][FV3_LPT];
    Real dxd[Q4_NT][FV3_LPT], dxn[Q4_NT][FV3_LPT], dyf[Q4_NT][FV3_LPT], zyp[FV3_LPT], zxo[FV3_LPT];
    Real mdu_n[FV3_LPT], mdv_n[FV3_LPT], mra_n[FV3_LPT];  // the chain's metric terms of the next row (fetched one step ahead)
    const MPtr d6Lb = d6L + m2, d6Mb = d6M + m2, rab = rarea + m2;

    const MPtr metLb = metL + m2;
    const unsigned pbase = (unsigned)(go * sj32 + go);
    auto load_row = [&](int r, int l, int lane) -> Row {
      const int rf = r - 2 < Msd ? Msd : r - 2;
      const unsigned p0 = pcol[l] + (unsigned)(r * MS), pf = pcol[l] + (unsigned)(rf * MS);
      Row w;
      Q4_EACH(n)
        w.qy[n] = qin[n][p0];
      Q4_END
      w.cx = crLb[p0];
      w.xv = afLb[p0];
      w.ar = areab[p0];
      w.cy = crMb[pf];
      w.yv = afMb[pf];
      w.em = (Real)1;
      if (PART != Q4_INTERIOR && (Llo || Lhi) && lane < 8) {
        const bool have = lane < 4 ? Llo : Lhi;
        const int sc = lane < 4 ? lane - 1 : npL - 2 + (lane - 4);
        if (have) w.em = metLb[pbase + (unsigned)(sc * LS + r * MS)];
      }
      return w;
    };
    // the inputs step r consumes at its end (rows r-3 / faces r-2), into the n_* registers
    auto load_opt = [&](int r, int l) {
      const int r3 = r - 3 < Msd ? Msd : r - 3, rf = r - 2 < Msd ? Msd : r - 2;
      const unsigned p3 = pcol[l] + (unsigned)(r3 * MS), pf = pcol[l] + (unsigned)(rf * MS);
      if constexpr (HAS_AIR) {
        n_ax[l] = (a.acc_first ? a.zeros : a.mfx + b)[p3];
        n_ay[l] = (a.acc_first ? a.zeros : a.mfy + b)[pf];
      } else {
        n_mx[l] = (a.fx + b)[p3];
        n_my[l] = (a.fy + b)[pf];
        n_mc[l] = (a.delp + b)[pf];
        if constexpr (!DN_RECOMP) n_dn[l] = (a.o_delp + b)[p3];
      }
    };
    FV3_LANES(blk, lane, l) {
      const int lc = l0 - 3 + lane, lcc = lc < Led ? lc : Led;
      pcol[l] = pbase + (unsigned)(lcc * LS);
      own_x[l] = lc >= l0 && lc < l0 + Q4_OUT && lc <= nL + 1;
      own_y[l] = lc >= l0 && lc < l0 + Q4_OUT && lc <= nL;
      if (PART == Q4_INTERIOR) {  // the columns the W / E one-sided formulas reach belong to the EDGE launch
        if (fl & bitLlo) {
          own_x[l] = own_x[l] && lc > Q4_EW;
          own_y[l] = own_y[l] && lc > Q4_EW;
        }
        if (fl & bitLhi) {
          own_x[l] = own_x[l] && lc < nL - Q4_EW + 2;
          own_y[l] = own_y[l] && lc < nL - Q4_EW + 1;
        }
      }
      a1[l] = a2[l] = a3[l] = (Real)1;
      o_ax[l] = o_ay[l] = era[l] = zx0[l] = zx1[l] = zy0[l] = zy1[l] = y_prev[l] = (Real)0;
      o_mx[l] = o_my[l] = o_mc[l] = mbk[l] = (Real)0;
      o_dn[l] = (Real)1;
      cx1[l] = cx2[l] = cx3[l] = xv1[l] = xv2[l] = xv3[l] = (Real)0;
      Q4_EACH(n)
        w2[n][l] = w3[n][l] = w4[n][l] = w5[n][l] = al_q[n][l] = v2[n][l] = v3[n][l] = v4[n][l] = v5[n][l] = al_v[n][l] = (Real)0;
        cq[n][l] = cv[n][l] = PpmCell{(Real)0, (Real)0, (Real)0, false};
        p_prev[n][l] = fi1[n][l] = fi2[n][l] = fi3[n][l] = fyin[n][l] = px[n][l] = fxk[n][l] = fyp[n][l] = o_dx[n][l] = o_dy[n][l] = (Real)0;
        sqx[n][l] = sqi[n][l] = (Real)0;
        if constexpr (!DPP) {
          if (lane == 0) exf[n][FV3_WAVE] = exp_[n][FV3_WAVE] = (Real)0;
          if (lane < 3) lq[n][lane] = lqi[n][lane] = lq[n][FV3_WAVE + 3 + lane] = lqi[n][FV3_WAVE + 3 + lane] = (Real)0;
        }
      Q4_END
      smb[l] = sxv[l] = (Real)0;
      if constexpr (!DPP) {
        if (lane == 0) exx[FV3_WAVE] = (Real)0;
      }
      zyp[l] = zxo[l] = mdu_n[l] = mdv_n[l] = mra_n[l] = (Real)0;
      Q4_EACH(n)
        sd0[n][l] = sd1[n][l] = sd2[n][l] = gx0[n][l] = gx1[n][l] = gy0[n][l] = gy1[n][l] = dxd[n][l] = dxn[n][l] = dyf[n][l] = (Real)0;
      Q4_END
      if constexpr (FD) {
        for (int v = 0; v < NRING; ++v)
          for (int q = 0; q < 4; ++q) RG(v, q)[lane] = v == RG_AR ? (Real)1 : (Real)0;  // (warm-up steps: outputs masked, keep the divisions finite)
        if constexpr (!DPP) {
          for (int q = 0; q < FDL; ++q) {
            Real *ln = fdl + q * (FV3_WAVE + 2);
            ln[lane] = (Real)0;
            if (lane == 0) ln[-1] = ln[FV3_WAVE] = (Real)0;
          }
        }
        const unsigned pm = pcol[l] + (unsigned)((ca - 3) * MS);
        mdu_n[l] = d6Mb[pm];
        mdv_n[l] = d6Lb[pm];
        mra_n[l] = rab[pm];
      }
      n_ax[l] = n_ay[l] = n_mx[l] = n_my[l] = n_mc[l] = (Real)0;
      n_dn[l] = (Real)1;
      fyp_air[l] = (Real)0;
      load_opt(ca - 3, l);
      nxt[l] = load_row(ca - 3, l, lane);
      nx2[l] = load_row(ca - 2 < r_end ? ca - 2 : r_end, l, lane);
      if constexpr (!DPP) {
        if (lane < 32) emr[lane] = (Real)1;
        exm[lane] = (Real)0;
      }
    }

    FV3_STAMP_STATE;
    auto march = [&](auto xe_tag) {
      constexpr bool XE = decltype(xe_tag)::value;  // one-sided formulas among the L faces of this strip (evaluated per lane)
      auto step = [&](int r) {
        FV3_STAMP(0);
        const int r3 = r - 3 < Msd ? Msd : r - 3;
        const int rn = r + Q4_PF < r_end ? r + Q4_PF : r_end;
        const int sy = r - 1;  // cell whose low edge value the M windows complete at this step
        const bool m_edge = (Mlo && sy >= 0 && sy <= 2) || (Mhi && sy >= npM - 1 && sy <= npM + 1);
        const bool corner_row = halo_cols && (r < 1 || r > nM);
        // ---- phase 1: prefetch row r+2; inner M fluxes at face r-2, the M-advected q at row r-3
        FV3_LANES(blk, lane, l) {
          {
            const int rf = r - 2 < Msd ? Msd : r - 2;
            const unsigned p3 = pcol[l] + (unsigned)(r3 * MS), pf = pcol[l] + (unsigned)(rf * MS);
            if constexpr (HAS_AIR) {
              o_ax[l] = n_ax[l];
              o_ay[l] = n_ay[l];
              if constexpr (!FD) {
                if (on_vt) {
                  o_dx[0][l] = (a.dpx + b)[p3];
                  o_dy[0][l] = (a.dpy + b)[pf];
                }
                if (on_w) {
                  zx0[l] = (a.dwx + b)[p3];
                  zx1[l] = (a.dwx + b)[p3 + (unsigned)LS];
                  zy0[l] = (a.dwy + b)[p3];
                  zy1[l] = (a.dwy + b)[p3 + (unsigned)MS];
                }
              }
            } else {
              o_mx[l] = n_mx[l];
              o_my[l] = n_my[l];
              o_mc[l] = n_mc[l];
              o_dn[l] = n_dn[l];
            }
            load_opt(r + 1 < r_end ? r + 1 : r_end, l);
            if constexpr (ROLE != Q4_AIR && !FD) {
              constexpr int nq = ROLE == Q4_QUAD ? 2 : 0, np_ = nq + 1;  // slots of q_con / pt
              if (on_vt) {
                o_dx[np_][l] = (a.dtx + b)[p3];
                o_dy[np_][l] = (a.dty + b)[pf];
              }
              if (on_t) {
                o_dx[nq][l] = (a.dqx + b)[p3];
                o_dy[nq][l] = (a.dqy + b)[pf];
              }
            }
            if constexpr (FD) {
              // the chain's metric terms: row r (fetched during the previous step) goes into the ring, row r+1 is requested
              const Real du_c = mdu_n[l], dv_c = mdv_n[l], ra_c = mra_n[l];
              const int r1 = r + 1 < r_end ? r + 1 : r_end;
              const unsigned pm = pcol[l] + (unsigned)(r1 * MS);
              mdu_n[l] = d6Mb[pm];
              mdv_n[l] = d6Lb[pm];
              mra_n[l] = rab[pm];
              RG(RG_DU, r)[lane] = du_c;
              RG(RG_DV, r)[lane] = dv_c;
              RG(RG_RA, r)[lane] = ra_c;
              era[l] = RG(RG_RA, r - 3)[lane];
            } else {
              era[l] = (rarea + m2)[p3];
            }
          }
          cur[l] = nxt[l];
          nxt[l] = nx2[l];
          nx2[l] = load_row(rn, l, lane);
          FV3_STAMP(1);  // loads of the step issued (incl. the waits the register rotation forces)
          FV3_STAMP_USE(cur[l].yv);
          FV3_STAMP_USE(cur[l].cy);
          FV3_STAMP_USE(cur[l].cx);
          FV3_STAMP_USE(cur[l].xv);
          FV3_STAMP_USE(cur[l].ar);
          FV3_STAMP_USE(cur[l].qy[0]);
          FV3_STAMP_USE(cur[l].qy[Q4_NT - 1]);
          FV3_STAMP(2);  // row r in registers
          if (XE && lane < 8) emr[(r & 3) * 8 + lane] = cur[l].em;
          const Real yv = cur[l].yv;
          const Real ar3 = PARK ? RG(RG_AR, r - 3)[lane] : a3[l];
          const Real den_y = ar3 + y_prev[l] - yv;
          Real du0 = (Real)0, du1 = (Real)0, du2 = (Real)0, ra1 = (Real)0, ra2 = (Real)0;
          if constexpr (FD) {
            du0 = RG(RG_DU, r)[lane];
            du1 = RG(RG_DU, r - 1)[lane];
            du2 = RG(RG_DU, r - 2)[lane];
            ra1 = RG(RG_RA, r - 1)[lane];
            ra2 = RG(RG_RA, r - 2)[lane];
          }
          Q4_EACH(n)
            Real qy = cur[l].qy[n], qx = qy;
            if (corner_row) {  // the two sweeps see the cube-corner cells through different remaps (rare, not prefetched)
              int lane_o = lane;
              FV3_LAUNDER(lane_o);  // (the remaps' per-lane predicates are formed here, not kept as lane masks across the march)
              const int lc = l0 - 3 + lane_o, lcc = lc < Led ? lc : Led;
              const int rc = r < Med ? r : Med;
              if (TR) {
                qy = cc<1>(qin[n], *gp, fl, rc, lcc);
                qx = cc<2>(qin[n], *gp, fl, rc, lcc);
              } else {
                qy = cc<2>(qin[n], *gp, fl, lcc, rc);
                qx = cc<1>(qin[n], *gp, fl, lcc, rc);
              }
              FV3_LANDED(qy);
              FV3_LANDED(qx);
              cur[l].qy[n] = qy;
            }
            if constexpr (FD) {
              // ---- del-n chain, own-lane part (del6_stream phase A): d2 of iteration s on row r-s, its M flux at face r-s.
              //      The raw field value enters (the corner-halo remaps belong to the patches).
              const Real qraw = cur[l].qy[n];
              const Real d0c = (id == 0 || id == 1) ? dcoef[n] * qraw : qraw;
              const Real fyc0 = du0 * (sd0[n][l] - d0c);
              Real gxe0, gxe1;  // the L fluxes of the lane's high faces: the neighbouring lane's values of the previous step
              if constexpr (DPP) {
                gxe0 = FV3_LANE_SHL(1, gx0[n], l, lane);
                gxe1 = FV3_LANE_SHL(1, gx1[n], l, lane);
              } else {
                gxe0 = LF(n, 0)[lane + 1];
                gxe1 = LF(n, 1)[lane + 1];
              }
              const Real d2c1 = TR ? (gy0[n][l] - fyc0 + gx0[n][l] - gxe0) * ra1 : (gx0[n][l] - gxe0 + gy0[n][l] - fyc0) * ra1;
              const Real fyc1 = du1 * (d2c1 - sd1[n][l]);
              const Real d2c2 = TR ? (gy1[n][l] - fyc1 + gx1[n][l] - gxe1) * ra2 : (gx1[n][l] - gxe1 + gy1[n][l] - fyc1) * ra2;
              dyf[n][l] = du2 * (d2c2 - sd2[n][l]);
              gy0[n][l] = fyc0;
              gy1[n][l] = fyc1;
              sd0[n][l] = d0c;
              sd1[n][l] = d2c1;
              sd2[n][l] = d2c2;
              if constexpr (!DPP) {
                LD(n, 0)[lane] = d0c;
                LD(n, 1)[lane] = d2c1;
                LD(n, 2)[lane] = d2c2;
              }
            }
            w2[n][l] = w3[n][l];
            w3[n][l] = w4[n][l];
            w4[n][l] = w5[n][l];
            w5[n][l] = qy;
            Real al_new;
            PpmCell co;
            if constexpr (M8) {
              const MPtr mmb = (TR ? gp->dxa : gp->dya) + m2;
              auto My = [&](int s_) { return mmb[pcol[l] + (unsigned)(s_ * MS)]; };
              al_new = ppm8_al_win(w2[n][l], w3[n][l], w4[n][l], w5[n][l], My, sy, m_edge && Mlo, m_edge && Mhi, npM);
              if (m_edge) FV3_LANDED(al_new);
              co = ppm8_cell(al_q[n][l], al_new, w3[n][l], ppm8_dm(w2[n][l], w3[n][l], w4[n][l]), ppm8_edge_cell(sy - 1, Mlo, Mhi, npM));
            } else {
            if (m_edge) {
              const MPtr mmb = (TR ? gp->dxa : gp->dya) + m2;
              auto My = [&](int s_) { return mmb[pcol[l] + (unsigned)(s_ * MS)]; };
              al_new = ppm_al_win(w2[n][l], w3[n][l], w4[n][l], w5[n][l], My, sy, Mlo, Mhi, npM);
              FV3_LANDED(al_new);
            } else {
              al_new = PPM_P1 * (w3[n][l] + w4[n][l]) + PPM_P2 * (w2[n][l] + w5[n][l]);
            }
            co = ppm_cell(al_q[n][l], al_new, w3[n][l], hord[n]);
            }
            al_q[n][l] = al_new;
            fyin[n][l] = ppm_face(cq[n][l], co, cur[l].cy);
            cq[n][l] = co;
            const Real pn = yv * fyin[n][l];
            const Real qi = (w2[n][l] * ar3 + p_prev[n][l] - pn) / den_y;
            p_prev[n][l] = pn;
            if constexpr (DPP) {
              sqx[n][l] = qx;
              sqi[n][l] = qi;
            } else {
              lq[n][3 + lane] = qx;
              lqi[n][3 + lane] = qi;
            }
            if constexpr (HC != 0) FV3_SCHED_FENCE();
          Q4_END
          y_prev[l] = yv;
          if constexpr (DPP)
            smb[l] = HAS_AIR ? w2[0][l] : mbk[l];
          else
            exm[lane] = HAS_AIR ? w2[0][l] : mbk[l];  // old air mass of the cell (lc, r-3)
        }
        if constexpr (!DPP) blk.wave_sync();
        if constexpr (HC != 0) FV3_SCHED_FENCE();
        FV3_STAMP(3);
        // ---- phase 2: inner L fluxes on row r, outer L fluxes on row r-3, final L fluxes of row r-3
        const int jr = r - 3;
        const bool fx_row = jr >= ca && jr <= cb;
        FV3_LANES(blk, lane, l) {
          const Real cx = cur[l].cx, xv = cur[l].xv;
          Real fxin[Q4_NT], fxout[Q4_NT];
          if constexpr (PARK) {  // rows r-3 of the parked delay lines
            cx3[l] = RG(RG_CX, r - 3)[lane];
            xv3[l] = RG(RG_XV, r - 3)[lane];
            Q4_EACH(n)
              fi3[n][l] = RG(RG_FI + n, r - 3)[lane];
            Q4_END
          }
          if constexpr (FD) {
            // ---- del-n chain, L fluxes (del6_stream phase B): the low neighbour's d2 of this step
            const Real dv0 = RG(RG_DV, r)[lane], dv1 = RG(RG_DV, r - 1)[lane], dv2 = RG(RG_DV, r - 2)[lane];
            Q4_EACH(n)
              Real w0, w1, w2_;
              if constexpr (DPP) {
                w0 = FV3_LANE_SHR(1, sd0[n], l, lane);
                w1 = FV3_LANE_SHR(1, sd1[n], l, lane);
                w2_ = FV3_LANE_SHR(1, sd2[n], l, lane);
              } else {
                w0 = LD(n, 0)[lane - 1];
                w1 = LD(n, 1)[lane - 1];
                w2_ = LD(n, 2)[lane - 1];
              }
              gx0[n][l] = dv0 * (w0 - sd0[n][l]);
              gx1[n][l] = dv1 * (sd1[n][l] - w1);
              dxn[n][l] = dv2 * (sd2[n][l] - w2_);  // final L flux of row r-2: the one of row r-3 (dxd) is consumed below
              // the damping fluxes this step consumes: L flux of (lc, r-3), M flux of face (lc, r-2); on a cube-corner patch from the fields
              o_dx[n][l] = dxd[n][l];
              o_dy[n][l] = dyf[n][l];
            Q4_END
            if (pz_lo || pz_hi) {
              int lane_o = lane;
              FV3_LAUNDER(lane_o);
              const int lc = l0 - 3 + lane_o;
              const int jr_ = r - 3, jf_ = r - 2;
              const bool px_ = jr_ >= 1 && jr_ <= nM && on_patch(lc, jr_), py_ = lc <= nL && on_patch(lc, jf_);
              Q4_EACH(n)
                if (px_) o_dx[n][l] = dLall[id][pcol[l] + (unsigned)(jr_ * MS)];
                if (py_) o_dy[n][l] = dMall[id][pcol[l] + (unsigned)(jf_ * MS)];
              Q4_END
              // (FD marches carry two tracers; an asm operand cannot name a capture inside the generic lambda of Q4_EACH)
              FV3_LANDED(o_dx[0][l]);
              FV3_LANDED(o_dy[0][l]);
              FV3_LANDED(o_dx[Q4_NT - 1][l]);
              FV3_LANDED(o_dy[Q4_NT - 1][l]);
            }
            if constexpr (HAS_AIR) {
              zx0[l] = o_dx[1][l];
              zy0[l] = zyp[l];
              zy1[l] = o_dy[1][l];
              zxo[l] = o_dx[1][l];
              if constexpr (!DPP) lzx[lane] = o_dx[1][l];
            }
          }
          Q4_EACH(n)
            if (XE) {
              const int lc = l0 - 3 + lane;
              auto EI = [&](int s_) { return s_ <= 2 ? s_ + 1 : s_ - (npL - 2) + 4; };  // ring column of an edge cell
              auto Qx = [&](int s_) { return lq[n][s_ - l0 + 6]; };
              auto Mx = [&](int s_) { return emr[(r & 3) * 8 + EI(s_)]; };
              fxin[n] = M8 ? ppm8_flux(Qx, Mx, cx, lc, Llo, Lhi, npL) : ppm_flux(Qx, Mx, cx, lc, Llo, Lhi, npL, hord[n]);
              auto Qi = [&](int s_) { return lqi[n][s_ - l0 + 6]; };
              auto Mx3 = [&](int s_) { return emr[((r - 3) & 3) * 8 + EI(s_)]; };
              fxout[n] = M8 ? ppm8_flux(Qi, Mx3, cx3[l], lc, Llo, Lhi, npL) : ppm_flux(Qi, Mx3, cx3[l], lc, Llo, Lhi, npL, hord[n]);
            } else if constexpr (DPP) {
              // the six cells around the face from the neighbouring lanes' registers (flux-limiter neighbour reads)
              const Real a0 = FV3_LANE_SHR(3, sqx[n], l, lane), a1_ = FV3_LANE_SHR(2, sqx[n], l, lane), a2_ = FV3_LANE_SHR(1, sqx[n], l, lane), a3_ = sqx[n][l],
                         a4 = FV3_LANE_SHL(1, sqx[n], l, lane), a5 = FV3_LANE_SHL(2, sqx[n], l, lane);
              const Real b0 = FV3_LANE_SHR(3, sqi[n], l, lane), b1 = FV3_LANE_SHR(2, sqi[n], l, lane), b2 = FV3_LANE_SHR(1, sqi[n], l, lane), b3 = sqi[n][l],
                         b4 = FV3_LANE_SHL(1, sqi[n], l, lane), b5 = FV3_LANE_SHL(2, sqi[n], l, lane);
              fxin[n] = M8 ? ppm8_flux_int(a0, a1_, a2_, a3_, a4, a5, cx) : ppm_flux_int(a0, a1_, a2_, a3_, a4, a5, cx, hord[n]);
              fxout[n] = M8 ? ppm8_flux_int(b0, b1, b2, b3, b4, b5, cx3[l]) : ppm_flux_int(b0, b1, b2, b3, b4, b5, cx3[l], hord[n]);
            } else {
              const Real *aq = lq[n] + lane, *bq = lqi[n] + lane;
              fxin[n] = M8 ? ppm8_flux_int(aq[0], aq[1], aq[2], aq[3], aq[4], aq[5], cx) : ppm_flux_int(aq[0], aq[1], aq[2], aq[3], aq[4], aq[5], cx, hord[n]);
              fxout[n] = M8 ? ppm8_flux_int(bq[0], bq[1], bq[2], bq[3], bq[4], bq[5], cx3[l]) : ppm_flux_int(bq[0], bq[1], bq[2], bq[3], bq[4], bq[5], cx3[l], hord[n]);
            }
            if constexpr (HC != 0) FV3_SCHED_FENCE();
          Q4_END
          const Real mb = HAS_AIR ? w2[0][l] : mbk[l];                    // old delp(lc, r-3)
          Real mwest;
          if constexpr (DPP)
            mwest = FV3_LANE_SHR(1, smb, l, lane);
          else
            mwest = lane > 0 ? exm[lane - 1] : (Real)0;
          const Real mw = mwest + mb;                                     // + old delp(lc-1, r-3)
          Real vx[Q4_NT];
          Real vm = o_mx[l];  // TRC: the stored air-mass flux of the face
          Q4_EACH(n)
            if constexpr (id == 0) {  // air mass: area-flux weighted, plain damping flux
              Real v = (Real)0.5 * (fxout[n] + fi3[n][l]) * xv3[l];
              if (on_vt) v = v + o_dx[n][l];
              if constexpr (UST) {
                const unsigned p = pcol[l] + (unsigned)(jr * MS);
                const bool ok = fx_row && own_x[l];
                fv3_store_sel((a.mfx + b) + p, sink0 + lane, ok, o_ax[l] + v);
                if constexpr (ROLE == Q4_AIR) fv3_store_sel((a.fx + b) + p, sink0 + lane, ok, v);
              } else if (fx_row && own_x[l]) {
                const unsigned p = pcol[l] + (unsigned)(jr * MS);
                FV3_MARCH_ST((a.mfx + b)[p], o_ax[l] + v);
                if constexpr (ROLE == Q4_AIR) FV3_MARCH_ST((a.fx + b)[p], v);
              }
              vm = v;
              vx[n] = v;
            } else {  // riding on the air-mass flux; q_con / pt with the mass-weighted damping flux
              Real v = (Real)0.5 * (fxout[n] + fi3[n][l]) * vm;
              if constexpr (id == 2) {
                if (on_t) v = v + (Real)0.5 * damp_t * mw * o_dx[n][l];
              } else if constexpr (id == 3) {
                if (on_vt) v = v + (Real)0.5 * damp_vt * mw * o_dx[n][l];
              }
              vx[n] = v;
            }
          Q4_END
          Q4_EACH(n)
            fxk[n][l] = vx[n];
            if constexpr (!DPP) exf[n][lane] = vx[n];
            if constexpr (PARK) {
              RG(RG_FI + n, r)[lane] = fxin[n];
            } else {
              fi3[n][l] = fi2[n][l];
              fi2[n][l] = fi1[n][l];
              fi1[n][l] = fxin[n];
            }
            px[n][l] = xv * fxin[n];
            if constexpr (!DPP) exp_[n][lane] = px[n][l];
            if constexpr (FD) {
              dxd[n][l] = dxn[n][l];
              if constexpr (!DPP) {
                LF(n, 0)[lane] = gx0[n][l];
                LF(n, 1)[lane] = gx1[n][l];
              }
            }
          Q4_END
          if constexpr (PARK) {
            RG(RG_CX, r)[lane] = cx;
            RG(RG_XV, r)[lane] = xv;
          } else {
            cx3[l] = cx2[l];
            cx2[l] = cx1[l];
            cx1[l] = cx;
            xv3[l] = xv2[l];
            xv2[l] = xv1[l];
            xv1[l] = xv;
          }
          if constexpr (DPP)
            sxv[l] = xv;
          else
            exx[lane] = xv;
        }
        if constexpr (!DPP) blk.wave_sync();
        if constexpr (HC != 0) FV3_SCHED_FENCE();
        FV3_STAMP(4);
        // ---- phase 3: the L-advected q on row r, outer M fluxes at face r-2, final M fluxes, the cell update of (lc, r-3)
        const int jf = r - 2;
        const bool fy_row = jf >= fa && jf <= fb;
        FV3_LANES(blk, lane, l) {
          Real x1;
          if constexpr (DPP)
            x1 = FV3_LANE_SHL(1, sxv, l, lane);
          else
            x1 = exx[lane + 1];
          const Real ar = cur[l].ar;
          const Real den_x = ar + cur[l].xv - x1;
          Real fyout[Q4_NT];
          Q4_EACH(n)
            Real p1;
            if constexpr (DPP)
              p1 = FV3_LANE_SHL(1, px[n], l, lane);
            else
              p1 = exp_[n][lane + 1];
            const Real qj = (cur[l].qy[n] * ar + px[n][l] - p1) / den_x;
            v2[n][l] = v3[n][l];
            v3[n][l] = v4[n][l];
            v4[n][l] = v5[n][l];
            v5[n][l] = qj;
            Real al_new;
            PpmCell co;
            if constexpr (M8) {
              const MPtr mmb = (TR ? gp->dxa : gp->dya) + m2;
              auto My = [&](int s_) { return mmb[pcol[l] + (unsigned)(s_ * MS)]; };
              al_new = ppm8_al_win(v2[n][l], v3[n][l], v4[n][l], v5[n][l], My, sy, m_edge && Mlo, m_edge && Mhi, npM);
              if (m_edge) FV3_LANDED(al_new);
              co = ppm8_cell(al_v[n][l], al_new, v3[n][l], ppm8_dm(v2[n][l], v3[n][l], v4[n][l]), ppm8_edge_cell(sy - 1, Mlo, Mhi, npM));
            } else {
            if (m_edge) {
              const MPtr mmb = (TR ? gp->dxa : gp->dya) + m2;
              auto My = [&](int s_) { return mmb[pcol[l] + (unsigned)(s_ * MS)]; };
              al_new = ppm_al_win(v2[n][l], v3[n][l], v4[n][l], v5[n][l], My, sy, Mlo, Mhi, npM);
              FV3_LANDED(al_new);
            } else {
              al_new = PPM_P1 * (v3[n][l] + v4[n][l]) + PPM_P2 * (v2[n][l] + v5[n][l]);
            }
            co = ppm_cell(al_v[n][l], al_new, v3[n][l], hord[n]);
            }
            al_v[n][l] = al_new;
            fyout[n] = ppm_face(cv[n][l], co, cur[l].cy);
            cv[n][l] = co;
            if constexpr (HC != 0) FV3_SCHED_FENCE();
          Q4_END
          const Real mb = HAS_AIR ? w2[0][l] : mbk[l], mc = HAS_AIR ? w3[0][l] : o_mc[l];  // old delp(lc, r-3), old delp(lc, r-2)
          Real vy[Q4_NT];
          Real vm = o_my[l];
          Q4_EACH(n)
            if constexpr (id == 0) {
              Real v = (Real)0.5 * (fyout[n] + fyin[n][l]) * cur[l].yv;
              if (on_vt) v = v + o_dy[n][l];
              if constexpr (UST) {
                const unsigned p = pcol[l] + (unsigned)(jf * MS);
                const bool ok = fy_row && own_y[l];
                fv3_store_sel((a.mfy + b) + p, sink0 + lane, ok, o_ay[l] + v);
                if constexpr (ROLE == Q4_AIR) fv3_store_sel((a.fy + b) + p, sink0 + lane, ok, v);
              } else if (fy_row && own_y[l]) {
                const unsigned p = pcol[l] + (unsigned)(jf * MS);
                FV3_MARCH_ST((a.mfy + b)[p], o_ay[l] + v);
                if constexpr (ROLE == Q4_AIR) FV3_MARCH_ST((a.fy + b)[p], v);
              }
              vm = v;
              vy[n] = v;
            } else {
              Real v = (Real)0.5 * (fyout[n] + fyin[n][l]) * vm;
              if constexpr (id == 2) {
                if (on_t) v = v + (Real)0.5 * damp_t * (mb + mc) * o_dy[n][l];
              } else if constexpr (id == 3) {
                if (on_vt) v = v + (Real)0.5 * damp_vt * (mb + mc) * o_dy[n][l];
              }
              vy[n] = v;
            }
          Q4_END
          Real fxe[Q4_NT];  // (read outside the branch below: a shuffle needs the source lane active)
          Q4_EACH(n)
            if constexpr (DPP)
              fxe[n] = FV3_LANE_SHL(1, fxk[n], l, lane);
            else
              fxe[n] = exf[n][lane + 1];
          Q4_END
          Real fe_air = (Real)0;
          if constexpr (DN_RECOMP) fe_air = FV3_LANE_SHL(1, o_mx, l, lane);  // air-mass flux through the cell's high L face
          if constexpr (FD && HAS_AIR) {  // w's damping flux through the high L face of the cell: the neighbouring lane's
            if constexpr (DPP)
              zx1[l] = FV3_LANE_SHL(1, zxo, l, lane);
            else
              zx1[l] = lzx[lane + 1];
          }
          const bool cell_ok = fx_row && own_y[l];
          if (UST || cell_ok) {
            // flux-form updates of the cell (lc, r-3): low L / M fluxes fxk / fyp, high L flux from lane + 1, high M flux = vy
            // (x terms first, as the reference writes the divergence)
            const unsigned p = pcol[l] + (unsigned)(jr * MS);
            Real *const sk = sink0 + lane;
            Real up[Q4_NT];
            Q4_EACH(n)
              const Real fe = fxe[n];  // L flux of the high L face: the neighbouring lane's
              const Real dv_ = TR ? (fyp[n][l] - vy[n] + fxk[n][l] - fe) * era[l] : (fxk[n][l] - fe + fyp[n][l] - vy[n]) * era[l];
              up[n] = id == 0 ? w2[n][l] + dv_ : mb * w2[n][l] + dv_;
            Q4_END
            Real dpn;  // new air mass of the cell
            if constexpr (HAS_AIR)
              dpn = up[0];
            else if constexpr (DN_RECOMP)
              dpn = mb + (o_mx[l] - fe_air + fyp_air[l] - o_my[l]) * era[l];
            else
              dpn = o_dn[l];
            Q4_EACH(n)
              if constexpr (id == 0) {
                fv3_store_sel((a.o_delp + b) + p, sk, cell_ok, dpn);
              } else if constexpr (id == 1) {
                Real wn = up[n] / dpn, hs = (Real)0;
                if (on_w) {
                  const Real dwv = TR ? (zy0[l] - zy1[l] + zx0[l] - zx1[l]) * era[l] : (zx0[l] - zx1[l] + zy0[l] - zy1[l]) * era[l];  // (x terms first)
                  hs = dd8 - dwv * (w2[n][l] + (Real)0.5 * dwv);
                  wn = wn + dwv;
                }
                fv3_store_sel((a.o_w + b) + p, sk, cell_ok, wn);
                fv3_store_sel((a.heat + b) + p, sk, cell_ok, hs);
              } else if constexpr (id == 2) {
                fv3_store_sel((a.o_q_con + b) + p, sk, cell_ok, up[n] / dpn);
              } else {
                fv3_store_sel((a.o_pt + b) + p, sk, cell_ok, up[n] / dpn);
              }
            Q4_END
          }
          if constexpr (!HAS_AIR) mbk[l] = o_mc[l];
          if constexpr (DN_RECOMP) fyp_air[l] = o_my[l];
          Q4_EACH(n)
            fyp[n][l] = vy[n];
          Q4_END
          if constexpr (FD && HAS_AIR) zyp[l] = zy1[l];
          if constexpr (PARK) {
            RG(RG_AR, r)[lane] = cur[l].ar;
          } else {
            a3[l] = a2[l];
            a2[l] = a1[l];
            a1[l] = cur[l].ar;
          }
        }
        if constexpr (!DPP) blk.wave_sync();
        FV3_STAMP(5);
      };
      // The prefetched rows rotate through three register sets (cur <- nxt <- nx2).  Unrolled by the rotation period the
      // copies are renames (the four-tracer wave, which has the registers); rolled they wait for the row fetched one
      // step earlier (the two-tracer waves: -50 registers, what lets them fit two waves per SIMD).
      if constexpr (XE || PART != Q4_ALL || ROLE != Q4_QUAD) {
        for (int r = ca - 3; r <= r_end; ++r) step(r);
      } else {
        for (int r = ca - 3; r <= r_end; r += Q4_PF + 1) {
          step(r);
          step(r + 1);
          step(r + 2);
        }
      }
    };
    if constexpr (PART == Q4_INTERIOR) {
      march(std::false_type{});
    } else {
      if (Llo || Lhi)
        march(std::true_type{});
      else
        march(std::false_type{});
    }
#if defined(FV3_STAMPS) && !defined(FV3_HOST_EMU)
    FV3_STAMP_FLUSH(st_buf, st_kid, blk.tid);
#endif
  });
}

void dsw_scalars_stream(fv3_ctx *c, fv3_stream_t s, const DswScalars &a, int mode) {
  int any = 0;
  for (int t = 0; t < c->g.nsub; ++t) any |= c->g.flags[t];
  const bool edges = any & (FV3_W | FV3_E);
  const int nz1 = c->g.nz - 1;
  if (mode == 0) {
    dsw_scalars_t<Q4_QUAD, Q4_ALL>(c, s, a, 0, nz1);
  } else if (mode == 2) {  // two-tracer waves, all strips in one launch (per-lane tile-edge formulas)
    dsw_scalars_t<Q4_AIR, Q4_ALL>(c, s, a, 0, nz1);
    dsw_scalars_t<Q4_TRC, Q4_ALL>(c, s, a, 0, nz1);
  } else {
    // levels below fd_k0 (the sponge layers: chains of lower order) read the damping fluxes del6_stream wrote; from fd_k0 on the
    // marches run the chains themselves
    const int kf = a.fd_k0 < 0 ? 0 : a.fd_k0 > nz1 + 1 ? nz1 + 1 : a.fd_k0;
    // The few levels below fd_k0 are launches of a handful of waves per CU that last as long as one wave's march: they go to the
    // auxiliary stream and run BESIDE the marches of the other levels (disjoint levels of the same arrays; the q_con + pt march of
    // a level follows the delp + w march of that level on either stream).  Events: 2 = fork, 3 = join.
    // (FV3_DSW_SPONGE_SERIAL=1: the sponge-level launches in program order on the caller's stream -- experiment R5-17)
    static const bool sponge_serial = getenv("FV3_DSW_SPONGE_SERIAL") && getenv("FV3_DSW_SPONGE_SERIAL")[0] == '1';
    fv3_stream_t s2 = kf > 0 && kf <= nz1 && !sponge_serial ? fv3_aux(c, s) : s;
    if (s2 != s) {
      fv3_signal(c, s, 2);
      fv3_wait(c, s2, 2);
    }
    dsw_scalars_t<Q4_AIR, Q4_INTERIOR>(c, s2, a, 0, kf - 1);
    if (edges) dsw_scalars_t<Q4_AIR, Q4_EDGE>(c, s2, a, 0, kf - 1);
    dsw_scalars_t<Q4_TRC, Q4_INTERIOR>(c, s2, a, 0, kf - 1);
    if (edges) dsw_scalars_t<Q4_TRC, Q4_EDGE>(c, s2, a, 0, kf - 1);
    if (s2 != s) fv3_signal(c, s2, 3);
    // (the big launches: the instantiation with the PPM order as a constant when the configuration has the reference's 6 everywhere;
    //  FV3_HORD_CONST=0 keeps the run-time form -- A/B, same values)
    //  Measured: -0.15 ms of d_sw for 12 spilled VGPRs in the two-tracer marches (the branch-free limiter needs ~24 more registers):
    //  off by default here (FV3_HORD_CONST=1 selects it), on in the single-tracer marches of fv3_tp2d.hip.
    static const bool hc_on = getenv("FV3_HORD_CONST") && getenv("FV3_HORD_CONST")[0] == '1';
    if (hc_on && a.hord_dp == 6 && a.hord_vt == 6 && a.hord_tm == 6) {
      dsw_scalars_t<Q4_AIR, Q4_INTERIOR, false, true, 6>(c, s, a, kf, nz1);
      if (edges) dsw_scalars_t<Q4_AIR, Q4_EDGE, false, true>(c, s, a, kf, nz1);
      dsw_scalars_t<Q4_TRC, Q4_INTERIOR, false, true, 6>(c, s, a, kf, nz1);
      if (edges) dsw_scalars_t<Q4_TRC, Q4_EDGE, false, true>(c, s, a, kf, nz1);
    } else {
      // Round 5: EVERY tile runs the march of fv3_tp4x.hip (PPM order 6 only; W / E one-sided formulas in the lanes, cube-corner remaps and patch
      // fluxes in its general steps).  FV3_DSW_MARCH=old: the round-4 kernels on every tile (A/B; read per call: the parity test flips it).
      const char *me = getenv("FV3_DSW_MARCH");
      const bool px_on = !(me && !strcmp(me, "old")) && a.hord_dp == 6 && a.hord_vt == 6 && a.hord_tm == 6;
      // (measured and dropped: the transposed tile-edge marches on the auxiliary stream beside the interior ones -- d_sw 51.5 ms either way)
      if (px_on) {  // (every tile: the W / E one-sided formulas in the lanes, the cube-corner remaps / patch fluxes in the general steps)
        // Round 6, FV3_DSW_MARCH=coupled (measured, NOT the default): the two roles as coupled wave pairs -- one workgroup of two waves per tile, the air-mass
        // fluxes / old air mass / shared rows handed over through LDS (fv3_tp4x.hip, PX_BOTH).  Bitwise equal and 27 GB lighter, but 24.7 ms against 14.8 for the
        // two launches at C768: a role's waves in flight are halved and every row step costs the slower role's time (EXPERIMENTS R6-5).
        if (me && !strcmp(me, "coupled")) {
          dsw_pair_march(c, s, a, 3, kf, nz1);
        } else {
          dsw_pair_march(c, s, a, 1, kf, nz1);
          dsw_pair_march(c, s, a, 2, kf, nz1);
        }
      } else {
        dsw_scalars_t<Q4_AIR, Q4_INTERIOR, false, true>(c, s, a, kf, nz1);
        if (edges) dsw_scalars_t<Q4_AIR, Q4_EDGE, false, true>(c, s, a, kf, nz1);
        dsw_scalars_t<Q4_TRC, Q4_INTERIOR, false, true>(c, s, a, kf, nz1);
        if (edges) dsw_scalars_t<Q4_TRC, Q4_EDGE, false, true>(c, s, a, kf, nz1);
      }
    }
    if (s2 != s) fv3_wait(c, s, 3);
  }
}

void tracer_pair_stream(fv3_ctx *c, fv3_stream_t s, const DswScalars &a) {
  int any = 0;
  for (int t = 0; t < c->g.nsub; ++t) any |= c->g.flags[t];
  const bool edges = any & (FV3_W | FV3_E);
  const int nz1 = c->g.nz - 1;
  if (a.hord_dp == 8) {
    dsw_scalars_t<Q4_TRC, Q4_INTERIOR, true>(c, s, a, 0, nz1);
    if (edges) dsw_scalars_t<Q4_TRC, Q4_EDGE, true>(c, s, a, 0, nz1);
  } else {
    dsw_scalars_t<Q4_TRC, Q4_INTERIOR>(c, s, a, 0, nz1);
    if (edges) dsw_scalars_t<Q4_TRC, Q4_EDGE>(c, s, a, 0, nz1);
  }
}
