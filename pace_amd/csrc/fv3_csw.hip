// fv3_csw.hip -- C-grid half step c_sw (d2a2c_vect, divergence_corner, upwind transport of
// delp / pt / w, C-grid kinetic energy + absolute vorticity, time-centred uc / vc).
// CPU twin: oracle/fv3_oracle/c_sw.py.  [SURVEY A.2; reference operator CGridShallowWaterDynamics,
// checkpoint variables REF tests/savepoint/thresholds/fv_dynamics.yaml:2-75]
//
// Five stages: (A) ua/va  (B) uc,ut / vc,vt with the dt2*dy*sin geometry factor folded in  (C) corner divergence
// (D) delpc, ptc, wc, ke + absolute vorticity  (E) uc/vc update.  utmp / vtmp of d2a2c_vect are pure functions of (u, v)
// and are evaluated in registers.
// Default form: ONE marching wave kernel does all five stages on the interior rectangle of every sub-domain (csw_fused_stream);
// the generic per-point stage kernels at the end of this file finish the windows along the sub-domain boundary, where the
// tile-edge formulas live.  FV3_CSW_MARCH=abc / =0 and FV3_CSW_B_GENERIC select the earlier forms (bitwise equal, A/B tests).
#include "fv3_ops.h"

#define CSW_A1 ((Real)0.5625)
#define CSW_A2 ((Real)-0.0625)
#define CSW_C1 ((Real)(-2.0 / 14.0))
#define CSW_C2 ((Real)(11.0 / 14.0))
#define CSW_C3 ((Real)(5.0 / 14.0))

namespace {

struct D2A {
  Geo g;
  const Real *u, *v;  // level bases
  bool W, E, S, N;

  FV3_HD bool two_pt(int i, int j) const {
    return (S && j <= 3) || (N && j >= g.npy - 3) || (W && i <= 3) || (E && i >= g.npx - 3);
  }
  // raw (no corner fix) utmp / vtmp
  FV3_HD Real utmp0(int i, int j) const {
    if (two_pt(i, j)) return (Real)0.5 * (u[IX(i, j)] + u[IX(i, j + 1)]);
    return CSW_A2 * (u[IX(i, j - 1)] + u[IX(i, j + 2)]) + CSW_A1 * (u[IX(i, j)] + u[IX(i, j + 1)]);
  }
  FV3_HD Real vtmp0(int i, int j) const {
    if (two_pt(i, j)) return (Real)0.5 * (v[IX(i, j)] + v[IX(i + 1, j)]);
    return CSW_A2 * (v[IX(i - 1, j)] + v[IX(i + 2, j)]) + CSW_A1 * (v[IX(i, j)] + v[IX(i + 1, j)]);
  }
  // utmp as seen by the x-direction A->C interpolation (corner halo rows fixed from vtmp)
  FV3_HD Real utmp_x(int i, int j) const {
    const int npx = g.npx, npy = g.npy, je = g.ny;
    if (S && j == 0) {
      if (W && i >= -2 && i <= 0) return -vtmp0(0, 1 - i);
      if (E && i >= npx && i <= npx + 2) return vtmp0(npx, i - npx + 1);
    }
    if (N && j == npy) {
      if (E && i >= npx && i <= npx + 2) return -vtmp0(npx, je - (i - npx));
      if (W && i >= -2 && i <= 0) return vtmp0(0, je + i);
    }
    return utmp0(i, j);
  }
  FV3_HD Real vtmp_y(int i, int j) const {
    const int npx = g.npx, npy = g.npy, ie = g.nx;
    if (W && i == 0) {
      if (S && j >= -2 && j <= 0) return -utmp0(1 - j, 0);
      if (N && j >= npy && j <= npy + 2) return utmp0(j - npy + 1, npy);
    }
    if (E && i == npx) {
      if (S && j >= -2 && j <= 0) return utmp0(ie + j, 0);
      if (N && j >= npy && j <= npy + 2) return -utmp0(ie - (j - npy), npy);
    }
    return vtmp0(i, j);
  }
};

FV3_HD inline Real edge_interp4(Real u1, Real u2, Real u3, Real u4, Real d1, Real d2, Real d3, Real d4) {
  return (Real)0.5 * ((((Real)2 * d2 + d1) * u2 - d2 * u1) / (d1 + d2) + (((Real)2 * d3 + d4) * u3 - d3 * u4) / (d3 + d4));
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Stages A + B + C on the interior rectangle of every sub-domain as ONE marching wave kernel.
// The three stages read nothing but (u, v): utmp (4-point in j) lives in a 4-row register window of u, vtmp
// (4-point in i) comes from wavefront shuffles of the v row, uc is a shuffle of utmp, vc a window of vtmp, and the corner
// divergence needs the 2 x 2 block of ua / va around the corner -- the row before from registers, the column before by
// shuffle.  A lane therefore loads 2 values per point where the two-row stage kernel loaded 20 and the divergence kernel
// re-read u, v, ua, va; ua / va / uc / vc / ut / vt / divgd are each written once.
// Geometry: a lane carries CSW_CPL = 2 adjacent columns (one 16-byte access per field row: FV3_VLANES in fv3_common.h), a
// strip owns CSW_OUT = 124 columns (virtual lanes 2..125; 0, 1, 126, 127 are the reach of the shuffles), a segment `seg`
// rows; step R loads row R and finishes ua / va / uc / ut and the corner divergence of row R-2, vc / vt of row R-1.
// Only points whose whole stencil uses the interior formulas are produced here (the same rectangle as before:
// 6 cells away from a cube-tile edge); the generic stage kernels keep the four windows along the sub-domain boundary.
// ---------------------------------------------------------------------------------------------
#ifndef CSW_CPL
#define CSW_CPL 2
#endif
#define CSW_NV (FV3_WAVE * CSW_CPL)
#define CSW_OUT (CSW_NV - 4)
#ifndef CSW_WPE
#define CSW_WPE 2
#endif
#define CSW_LPT FV3_VLPT(CSW_CPL)
#define CSW_LANES(vl, l) FV3_VLANES(CSW_CPL, blk, vl, l)
#define CSW_SHR(K, arr) FV3_VSHR(CSW_CPL, K, arr, l, vl)
#define CSW_SHL(K, arr) FV3_VSHL(CSW_CPL, K, arr, l, vl)
struct CswRect {
  int i_lo, i_hi, j_lo, j_hi;
};
FV3_HD inline CswRect csw_rect(int fl, int nx, int ny, int npx, int npy) {
  CswRect r;
  r.i_lo = (fl & FV3_W) ? 6 : 0;
  r.i_hi = (fl & FV3_E) ? npx - 5 : nx + 1;
  r.j_lo = (fl & FV3_S) ? 6 : 0;
  r.j_hi = (fl & FV3_N) ? npy - 5 : ny + 1;
  return r;
}

static void csw_abc_stream(fv3_ctx *c, fv3_stream_t s, const Real *u, const Real *v, Real *ua, Real *va, Real *uc, Real *vc, Real *ut, Real *vt, Real *divgd,
                           Real dt2, bool do_div) {
  const Geo g = c->g;
  const Geo *gp = c->g_dev;
  const int nk = g.nz;
  const int nstrip = (g.nx + 2 + CSW_OUT - 1) / CSW_OUT;
  const int seg = fv3_pick_seg((long)nstrip * CSW_CPL * ((g.ny + 63) / 64) * g.nsub * nk, CSW_WPE);
  const int nseg = (g.ny + 2 + seg - 1) / seg;
  const int sj32 = g.sj32, go = g.o, nx = g.nx, ny = g.ny, npx = g.npx, npy = g.npy, nh = g.nh;
  const long st = g.st, sk = g.sk, st2 = g.st2;
  const MPtr m_cosa_s = g.cosa_s, m_rsin2 = g.rsin2, m_cosa_u = g.cosa_u, m_rsin_u = g.rsin_u, m_dy = g.dy, m_sin3 = g.sin_sg3, m_sin1 = g.sin_sg1;
  const MPtr m_cosa_v = g.cosa_v, m_rsin_v = g.rsin_v, m_dx = g.dx, m_sin4 = g.sin_sg4, m_sin2 = g.sin_sg2;
  const MPtr m_cos4 = g.cos_sg4, m_cos2 = g.cos_sg2, m_dyc = g.dyc, m_cos3 = g.cos_sg3, m_cos1 = g.cos_sg1, m_dxc = g.dxc, m_rac = g.rarea_c;
  // Launch geometry: the "plane" an XCD walks (fv3_grid) is one (sub-domain, strip, segment) tile and the tiles of the plane are
  // its levels, so the waves resident on an XCD are all the levels of a few tiles and share the metric rows in that XCD's L2.
  const int ntile = nstrip * nseg;
  launch_waves<CSW_WPE>(c, s, nk, 1, ntile * g.nsub, 0, [=] FV3_HD(const Blk &blk, char *) {
    const int k = blk.bx, t = blk.bz / ntile, tile = blk.bz - t * ntile;
    const int tby = tile / nstrip, tbx = tile - tby * nstrip;
    const CswRect rc = csw_rect(gp->flags[t], nx, ny, npx, npy);
    const int c0 = rc.i_lo + tbx * CSW_OUT;
    const int ja = rc.j_lo + tby * seg;
    if (c0 > rc.i_hi || ja > rc.j_hi) return;
    const int jb = ja + seg - 1 < rc.j_hi ? ja + seg - 1 : rc.j_hi;
    const long b = t * st + k * sk, m2 = t * st2;
    const Real *ub = u + b, *vb = v + b;
    const int Led = nx + nh, Msd = 1 - nh;
    const unsigned pbase = (unsigned)(go * sj32 + go);
    const int r_end = jb + 2;

    // per (virtual) lane: windows of u / v (rows R-3 .. R) and vtmp, rows R+1 / R+2 in flight, the metric terms of the
    // step (rows R-2: cs .. c4, R-1: cv .. s2) and of the next one, values carried from the step before
    Real u0[CSW_LPT], u1[CSW_LPT], u2[CSW_LPT], u3[CSW_LPT], v0[CSW_LPT], v1[CSW_LPT], v2[CSW_LPT], v3[CSW_LPT];
    Real t0[CSW_LPT], t1[CSW_LPT], t2[CSW_LPT], t3[CSW_LPT];
    Real nu1[CSW_LPT], nv1[CSW_LPT], nu2[CSW_LPT], nv2[CSW_LPT];
    Real s_ut[CSW_LPT], s_v[CSW_LPT], s_ua[CSW_LPT], s_uf[CSW_LPT];  // values the neighbouring lanes read
    Real uav[CSW_LPT], ufv[CSW_LPT], va_prev[CSW_LPT], vf_prev[CSW_LPT], s4_prev[CSW_LPT], c4_prev[CSW_LPT], s2_prev[CSW_LPT];
#define CSW_MET(X) X(cs) X(r2) X(cu) X(ru) X(dy) X(s3) X(s1) X(c2) X(dyc) X(c3) X(c1) X(dxc) X(rac) X(s4) X(c4) X(cv) X(rv) X(dx) X(s2)
#define CSW_DECL(n) Real mc_##n[CSW_LPT], mn_##n[CSW_LPT];
    CSW_MET(CSW_DECL)
#undef CSW_DECL
    Real o_a[CSW_LPT], o_b[CSW_LPT], o_c[CSW_LPT], o_d[CSW_LPT], o_e[CSW_LPT], o_f[CSW_LPT];  // results of a step on their way out
    bool own[CSW_LPT], own_d[CSW_LPT];
    // Row accesses: the CPL columns of a lane are one 16-byte access (device); slot l sits at pcol0 + l.  The strips' last
    // lanes are clamped as a pair (their columns are not owned).
#ifdef FV3_HOST_EMU
    unsigned pcolv[CSW_LPT];
    for (int q_ = 0; q_ < CSW_NV; ++q_) {
      const int lc0 = c0 - 2 + (q_ / CSW_CPL) * CSW_CPL;
      pcolv[q_] = pbase + (unsigned)(lc0 + CSW_CPL - 1 <= Led ? lc0 : Led - (CSW_CPL - 1)) + (unsigned)(q_ % CSW_CPL);
    }
#define CSW_ROW(dst, ptr, rowoff) \
  for (int q_ = 0; q_ < CSW_NV; ++q_) dst[q_] = (ptr)[pcolv[q_] + (unsigned)(rowoff)];
#define CSW_STORE(ptr, rowoff, src, ownarr) \
  for (int q_ = 0; q_ < CSW_NV; ++q_)       \
    if (ownarr[q_]) (ptr)[pcolv[q_] + (unsigned)(rowoff)] = src[q_];
#else
    unsigned pcol0;
    {
      const int lc0 = c0 - 2 + blk.tid * CSW_CPL;
      pcol0 = pbase + (unsigned)(lc0 + CSW_CPL - 1 <= Led ? lc0 : Led - (CSW_CPL - 1));
    }
#define CSW_ROW(dst, ptr, rowoff) fv3_ld_cpl<CSW_CPL>(dst, (ptr) + (pcol0 + (unsigned)(rowoff)));
#define CSW_STORE(ptr, rowoff, src, ownarr) fv3_st_cpl<CSW_CPL>((ptr) + (pcol0 + (unsigned)(rowoff)), src, ownarr);
#endif
    // the metric terms of step R_: rows R_-2 (cs .. c4) and R_-1 (cv .. s2) of the lane's own columns (sin_sg3 / cos_sg3 of the
    // column to the west come from the neighbouring slot)
#define CSW_LOAD_MET(R_)                                                                      \
  {                                                                                           \
    const int ra_ = ((R_) - 2 < Msd ? Msd : (R_) - 2) * sj32, rb_ = ((R_) - 1 < Msd ? Msd : (R_) - 1) * sj32; \
    CSW_ROW(mn_cs, m_cosa_s + m2, ra_)                                                        \
    CSW_ROW(mn_r2, m_rsin2 + m2, ra_)                                                         \
    CSW_ROW(mn_cu, m_cosa_u + m2, ra_)                                                        \
    CSW_ROW(mn_ru, m_rsin_u + m2, ra_)                                                        \
    CSW_ROW(mn_dy, m_dy + m2, ra_)                                                            \
    CSW_ROW(mn_s3, m_sin3 + m2, ra_)                                                          \
    CSW_ROW(mn_s1, m_sin1 + m2, ra_)                                                          \
    CSW_ROW(mn_c2, m_cos2 + m2, ra_)                                                          \
    CSW_ROW(mn_dyc, m_dyc + m2, ra_)                                                          \
    CSW_ROW(mn_c3, m_cos3 + m2, ra_)                                                          \
    CSW_ROW(mn_c1, m_cos1 + m2, ra_)                                                          \
    CSW_ROW(mn_dxc, m_dxc + m2, ra_)                                                          \
    CSW_ROW(mn_rac, m_rac + m2, ra_)                                                          \
    CSW_ROW(mn_s4, m_sin4 + m2, ra_)                                                          \
    CSW_ROW(mn_c4, m_cos4 + m2, ra_)                                                          \
    CSW_ROW(mn_cv, m_cosa_v + m2, rb_)                                                        \
    CSW_ROW(mn_rv, m_rsin_v + m2, rb_)                                                        \
    CSW_ROW(mn_dx, m_dx + m2, rb_)                                                            \
    CSW_ROW(mn_s2, m_sin2 + m2, rb_)                                                          \
  }
    CSW_LANES(vl, l)
      const int lc = c0 - 2 + vl;
      own[l] = vl >= 2 && vl <= CSW_OUT + 1 && lc <= rc.i_hi;
      own_d[l] = own[l] && lc >= 1;
      u0[l] = u1[l] = u2[l] = u3[l] = v0[l] = v1[l] = v2[l] = v3[l] = t0[l] = t1[l] = t2[l] = t3[l] = (Real)0;
      s_ut[l] = s_v[l] = s_ua[l] = s_uf[l] = uav[l] = ufv[l] = va_prev[l] = vf_prev[l] = s4_prev[l] = c4_prev[l] = s2_prev[l] = (Real)0;
      o_a[l] = o_b[l] = o_c[l] = o_d[l] = o_e[l] = o_f[l] = (Real)0;
#define CSW_ZERO(n) mc_##n[l] = (Real)0;
      CSW_MET(CSW_ZERO)
#undef CSW_ZERO
    FV3_VLANES_END
    {
      const int ra = (ja - 2) * sj32, rb = (ja - 1 < r_end ? ja - 1 : r_end) * sj32;
      CSW_ROW(nu1, ub, ra)
      CSW_ROW(nv1, vb, ra)
      CSW_ROW(nu2, ub, rb)
      CSW_ROW(nv2, vb, rb)
      CSW_LOAD_MET(ja - 2)
    }
    for (int R = ja - 2; R <= r_end; ++R) {
      const int rn = (R + 2 < r_end ? R + 2 : r_end) * sj32;
      const int jr = R - 2, jv = R - 1;
      const bool row_r = jr >= ja && jr <= jb, row_v = jv >= ja && jv <= jb;
      // ---- phase 1: rotate the rows in flight, the u / v windows, the metric terms; fetch ahead; utmp of row R-2
      CSW_LANES(vl, l)
        u0[l] = u1[l];
        u1[l] = u2[l];
        u2[l] = u3[l];
        u3[l] = nu1[l];
        v0[l] = v1[l];
        v1[l] = v2[l];
        v2[l] = v3[l];
        v3[l] = nv1[l];
        nu1[l] = nu2[l];
        nv1[l] = nv2[l];
#define CSW_ROT(n) mc_##n[l] = mn_##n[l];
        CSW_MET(CSW_ROT)
#undef CSW_ROT
        s_ut[l] = CSW_A2 * (u0[l] + u3[l]) + CSW_A1 * (u1[l] + u2[l]);
        s_v[l] = v3[l];
      FV3_VLANES_END
      CSW_ROW(nu2, ub, rn)
      CSW_ROW(nv2, vb, rn)
      CSW_LOAD_MET(R + 1)
      // ---- phase 2: vtmp of row R; ua / va / uc / ut of row R-2, vc / vt of row R-1; the u-face term of the divergence
      CSW_LANES(vl, l)
        const Real vt_new = CSW_A2 * (CSW_SHR(1, s_v) + CSW_SHL(2, s_v)) + CSW_A1 * (s_v[l] + CSW_SHL(1, s_v));
        t0[l] = t1[l];
        t1[l] = t2[l];
        t2[l] = t3[l];
        t3[l] = vt_new;
        const Real ut0 = s_ut[l], vt0 = t1[l];
        const Real ua_ = (ut0 - vt0 * mc_cs[l]) * mc_r2[l], va_ = (vt0 - ut0 * mc_cs[l]) * mc_r2[l];
        const Real ucv = CSW_A2 * (CSW_SHR(2, s_ut) + CSW_SHL(1, s_ut)) + CSW_A1 * (CSW_SHR(1, s_ut) + ut0);
        const Real utv = (ucv - v1[l] * mc_cu[l]) * mc_ru[l];
        const Real vcv = CSW_A2 * (t0[l] + t3[l]) + CSW_A1 * (t1[l] + t2[l]);
        const Real vtv = (vcv - u2[l] * mc_cv[l]) * mc_rv[l];
        const Real s3w = CSW_SHR(1, mc_s3);  // sin_sg3 of the column to the west
        o_a[l] = ua_;
        o_b[l] = va_;
        o_c[l] = ucv;
        o_d[l] = utv > (Real)0 ? dt2 * utv * mc_dy[l] * s3w : dt2 * utv * mc_dy[l] * mc_s1[l];
        o_e[l] = vcv;
        o_f[l] = vtv > (Real)0 ? dt2 * vtv * mc_dx[l] * mc_s4[l] : dt2 * vtv * mc_dx[l] * mc_s2[l];
        // u-face term of the corner divergence at (lc, R-2): cos / sin sums over the rows R-3 and R-2
        ufv[l] = (u1[l] - (Real)0.25 * (va_prev[l] + va_) * (c4_prev[l] + mc_c2[l])) * mc_dyc[l] * (Real)0.5 * (s4_prev[l] + s2_prev[l]);
        s_uf[l] = ufv[l];
        uav[l] = ua_;
        s_ua[l] = ua_;
        va_prev[l] = va_;
        c4_prev[l] = mc_c4[l];
        s4_prev[l] = mc_s4[l];
        s2_prev[l] = mc_s2[l];
      FV3_VLANES_END
      if (row_r) {
        CSW_STORE(ua + b, jr * sj32, o_a, own)
        CSW_STORE(va + b, jr * sj32, o_b, own)
        CSW_STORE(uc + b, jr * sj32, o_c, own)
        CSW_STORE(ut + b, jr * sj32, o_d, own)
      }
      if (row_v) {
        CSW_STORE(vc + b, jv * sj32, o_e, own)
        CSW_STORE(vt + b, jv * sj32, o_f, own)
      }
      // ---- phase 3: the v-face term and the divergence of the corner (lc, R-2)
      if (do_div) {
        CSW_LANES(vl, l)
          const Real s3w = CSW_SHR(1, mc_s3), c3w = CSW_SHR(1, mc_c3);
          const Real vf = (v1[l] - (Real)0.25 * (CSW_SHR(1, s_ua) + uav[l]) * (c3w + mc_c1[l])) * mc_dxc[l] * (Real)0.5 * (s3w + mc_s1[l]);
          const Real dv = vf_prev[l] - vf + CSW_SHR(1, s_uf) - ufv[l];
          o_a[l] = mc_rac[l] * dv;
          vf_prev[l] = vf;
        FV3_VLANES_END
        if (row_r && jr >= 1) CSW_STORE(divgd + b, jr * sj32, o_a, own_d)
      }
    }
#undef CSW_ROW
#undef CSW_STORE
#undef CSW_LOAD_MET
#undef CSW_MET
  });
}

// ---------------------------------------------------------------------------------------------
// The whole of c_sw (stages A - E) on the interior rectangle as ONE marching wave kernel: besides (u, v) a lane reads
// delp / pt / w of its column once and writes the ten outputs once (35 GB at C768 where the stage kernels moved 97).
// The intermediate C-grid winds uc0 / vc0, the kinetic energy and the corner vorticity live in the march: step R loads row R
// and finishes, in this order, ua / va / uc0 / ut / the divergence / the upwind transport / ke / the vorticity of row R-2
// and vc0 / vt of row R-1 (as csw_abc_stream), the time-centred vc of row R-2 and the time-centred uc of row R-3 (which
// needs the vorticity of the row above).  Regions, with R0 = [i_lo, i_hi] x [j_lo, j_hi] the interior rectangle:
//   ua va ut vt           R0                              divgd   corners of R0 from (1, 1)
//   delpc ptc omga, ke    RD = R0 minus its last column / row (the east / north face values come from the neighbour)
//   vorticity             VR = corners of R0 from (i_lo + 1, j_lo + 1)
//   final uc / vc         RE = R0 minus a one-cell rim (ke of the cell to the west / south, vorticity of the corner above)
// On the rim R0 \ RE the march stores the intermediate uc0 / vc0 instead, and ke / vorticity on the two cells next to
// the rectangle's boundary: exactly what the generic stage D / E kernels -- which now only run on the windows along the
// sub-domain boundary -- read from the interior.  One lane per column here (CSW_CPL = 1): 58 owned columns per strip.
// ---------------------------------------------------------------------------------------------
#define CSWF_OUT (FV3_WAVE - 6)
#ifndef CSWF_WPE
#define CSWF_WPE 2
#endif
#ifndef CSWF_NW
#define CSWF_NW 4  // waves (= levels) per workgroup
#endif
// (the outputs are written once and not read again by this kernel: non-temporal stores, 6 % faster at C768)
#if !defined(CSWF_NO_NT) && !defined(FV3_HOST_EMU)
#define CSWF_ST(ptr, val) __builtin_nontemporal_store((Real)(val), (ptr))
#define CSWF_STE(base, idx, val) __builtin_nontemporal_store((Real)(val), &FV3_EL(base, idx))
#else
#define CSWF_ST(ptr, val) (*(ptr) = (val))
#define CSWF_STE(base, idx, val) (FV3_EL(base, idx) = (val))
#endif
static void csw_fused_stream(fv3_ctx *c, fv3_stream_t s, const Real *u, const Real *v, const Real *delp, const Real *pt, const Real *w, Real *ua, Real *va, Real *uc,
                             Real *vc, Real *ut, Real *vt, Real *divgd, Real *delpc, Real *ptc, Real *omga, Real *ke, Real *vort, Real dt2, bool do_div, bool uava_thin) {
  const Geo g = c->g;
  const Geo *gp = c->g_dev;
  const int nk = g.nz;
  const int *nord_tab = g.nord;  // (per level: 0 = d_sw forms the divergence from ua / va there)
  const int nstrip = (g.nx + 2 + CSWF_OUT - 1) / CSWF_OUT;
  const int seg = fv3_pick_seg((long)nstrip * ((g.ny + 63) / 64) * g.nsub * nk, CSWF_WPE);
  const int nseg = (g.ny + 2 + seg - 1) / seg;
  const int sj32 = g.sj32, go = g.o, nx = g.nx, ny = g.ny, npx = g.npx, npy = g.npy, nh = g.nh;
  const long st = g.st, sk = g.sk, st2 = g.st2;
  const MPtr m_cosa_s = g.cosa_s, m_rsin2 = g.rsin2, m_cosa_u = g.cosa_u, m_rsin_u = g.rsin_u, m_dy = g.dy, m_sin3 = g.sin_sg3, m_sin1 = g.sin_sg1;
  const MPtr m_cosa_v = g.cosa_v, m_rsin_v = g.rsin_v, m_dx = g.dx, m_sin4 = g.sin_sg4, m_sin2 = g.sin_sg2;
  const MPtr m_cos4 = g.cos_sg4, m_cos2 = g.cos_sg2, m_dyc = g.dyc, m_cos3 = g.cos_sg3, m_cos1 = g.cos_sg1, m_dxc = g.dxc, m_rac = g.rarea_c;
  const MPtr m_rarea = g.rarea, m_fC = g.fC, m_sina_u = g.sina_u, m_rdxc = g.rdxc, m_sina_v = g.sina_v, m_rdyc = g.rdyc;
  const int ntile = nstrip * nseg;
  // (level-major launch geometry: see csw_abc_stream; CSWF_NW consecutive levels of a tile form one workgroup = run on one CU)
  launch_wave_groups<CSWF_WPE, CSWF_NW>(c, s, (nk + CSWF_NW - 1) / CSWF_NW, ntile * g.nsub, sizeof(Real) * 2 * 25 * FV3_WAVE, [=] FV3_HD(const Blk &blk, char *smem_) {
    // (a wave beyond the last level stays: the others need its share of the metric terms; it repeats the last level and stores nothing)
    const int k_ = blk.bx * CSWF_NW + blk.by, t = blk.bz / ntile, tile = blk.bz - t * ntile;
    const bool live = k_ < nk;
    const int k = live ? k_ : nk - 1;
    // Round 6 (uava_thin: inside the sequencer, not the last sub-step of a call): ua / va of this level are stored in full only where d_sw reads them (no damping
    // chain on the level); elsewhere only the two cells next to the rectangle's boundary, which the boundary-window kernels read -- two field writes less.
    const bool sua = !uava_thin || nord_tab[k] == 0;
    const int tby = tile / nstrip, tbx = tile - tby * nstrip;
    const CswRect rc = csw_rect(gp->flags[t], nx, ny, npx, npy);
    const int c0 = rc.i_lo + tbx * CSWF_OUT;
    const int ja = rc.j_lo + tby * seg;
    if (c0 > rc.i_hi || ja > rc.j_hi) return;
    const int jb = ja + seg - 1 < rc.j_hi ? ja + seg - 1 : rc.j_hi;
    const long b = t * st + k * sk, m2 = t * st2;
    const Real *ub = u + b, *vb = v + b, *db = delp + b, *pb_ = pt + b, *wb = w + b;
    const int Led = nx + nh, Msd = 1 - nh;
    const unsigned pbase = (unsigned)(go * sj32 + go);
    const int r_end = jb + 3 < rc.j_hi + 2 ? jb + 3 : rc.j_hi + 2;
    const int re_i0 = rc.i_lo + 1 > 1 ? rc.i_lo + 1 : 1, re_j0 = rc.j_lo + 1 > 1 ? rc.j_lo + 1 : 1;  // first column / row of RE (and of VR)

    Real u0[FV3_LPT], u1[FV3_LPT], u2[FV3_LPT], u3[FV3_LPT], v0[FV3_LPT], v1[FV3_LPT], v2[FV3_LPT], v3[FV3_LPT];
    Real t0[FV3_LPT], t1[FV3_LPT], t2[FV3_LPT], t3[FV3_LPT];
    Real d0[FV3_LPT], d1[FV3_LPT], d2[FV3_LPT], p0[FV3_LPT], p1[FV3_LPT], p2[FV3_LPT], w0[FV3_LPT], w1[FV3_LPT], w2[FV3_LPT];  // delp / pt / w rows R-3 .. R-1
    // Rows in flight: u / v of rows R+1, R+2 and delp / pt / w of rows R, R+1, in TWO register sets that alternate by the step's STATIC parity
    // (the march is unrolled by two).  A rolled rotation (next = next2 at the top of every step) is a register copy of a load still in flight:
    // it waits for the row requested ONE step ago, i.e. the prefetch distance of two rows was really one (-DCSWF_ROLLED: that form, A/B).
    Real nfu[2][FV3_LPT], nfv[2][FV3_LPT], nfd[2][FV3_LPT], nfp[2][FV3_LPT], nfw[2][FV3_LPT];
    // values the neighbouring lanes read
    Real s_ut[FV3_LPT], s_v[FV3_LPT], s_ua[FV3_LPT], s_uf[FV3_LPT], s_uc[FV3_LPT], s_d[FV3_LPT], s_p[FV3_LPT], s_w[FV3_LPT];
    Real s_f1[FV3_LPT], s_f[FV3_LPT], s_f2[FV3_LPT], s_pvd[FV3_LPT], s_ke[FV3_LPT], s_vo[FV3_LPT], s_s3[FV3_LPT], s_c3[FV3_LPT];
    // carried from the step before
    Real va_prev[FV3_LPT], vf_prev[FV3_LPT], s4_prev[FV3_LPT], c4_prev[FV3_LPT], s2_prev[FV3_LPT];
    Real vt_prev[FV3_LPT], vc_prev[FV3_LPT], cv_prev[FV3_LPT], cu_prev[FV3_LPT], uc_prev[FV3_LPT], dxc_prev[FV3_LPT];
    Real g1_lo[FV3_LPT], g_lo[FV3_LPT], g2_lo[FV3_LPT], ke_prev[FV3_LPT], vo_prev[FV3_LPT];
    // within a step
    Real uav[FV3_LPT], vav[FV3_LPT], ucv_[FV3_LPT], vcv_[FV3_LPT], ufv[FV3_LPT], utq[FV3_LPT], vtq[FV3_LPT], ke_w[FV3_LPT], kev_[FV3_LPT], vov_[FV3_LPT];
#define CSW_METT(X)                                                                                                                          \
  X(0, cs, m_cosa_s, pa) X(1, r2, m_rsin2, pa) X(2, cu, m_cosa_u, pa) X(3, ru, m_rsin_u, pa) X(4, dy, m_dy, pa) X(5, s3, m_sin3, pa)              \
  X(6, s1, m_sin1, pa) X(7, c2, m_cos2, pa) X(8, dyc, m_dyc, pa) X(9, c3, m_cos3, pa) X(10, c1, m_cos1, pa) X(11, dxc, m_dxc, pa)                 \
  X(12, rac, m_rac, pa) X(13, s4, m_sin4, pa) X(14, c4, m_cos4, pa) X(15, ra, m_rarea, pa) X(16, fc, m_fC, pa) X(17, sv, m_sina_v, pa)            \
  X(18, rdy, m_rdyc, pa) X(19, cv, m_cosa_v, pb) X(20, rv, m_rsin_v, pb) X(21, dx, m_dx, pb) X(22, s2, m_sin2, pb) X(23, su, m_sina_u, pc)        \
  X(24, rdx, m_rdxc, pc)
#define CSWF_NMET 25
#define CSW_DECL(i, n, ptr, pp) Real mc_##n[FV3_LPT], mn_##n[FV3_LPT];
    CSW_METT(CSW_DECL)
#undef CSW_DECL
    unsigned pcol[FV3_LPT];
    bool c_r0[FV3_LPT], c_rd[FV3_LPT], c_vr[FV3_LPT], c_div[FV3_LPT], c_re[FV3_LPT], c_kb[FV3_LPT], c_vb[FV3_LPT];
    // metric terms of step R_: rows R_-2 (cs .. rdy), R_-1 (cv .. s2), R_-3 (su, rdx) of the lane's column
#define CSW_ROWS(R_)                                                                                                                          \
  const unsigned pa = pcol[l] + (unsigned)(((R_) - 2 < Msd ? Msd : (R_) - 2) * sj32), pb = pcol[l] + (unsigned)(((R_) - 1 < Msd ? Msd : (R_) - 1) * sj32), \
                 pc = pcol[l] + (unsigned)(((R_) - 3 < Msd ? Msd : (R_) - 3) * sj32);
#define CSW_LD1(i, n, ptr, pp) mn_##n[l] = FV3_EL(ptr + m2, pp);
#define CSW_LOAD_MET(R_) \
  {                      \
    CSW_ROWS(R_)         \
    CSW_METT(CSW_LD1)    \
  }
    // Device: the CSWF_NW waves of the workgroup walk the same rows of CSWF_NW levels.  Wave wv fetches the terms i = wv (mod
    // CSWF_NW) of step R+2 during step R, hands them over through LDS during step R+1 (one barrier per step, two slots), and
    // every wave reads the 25 terms of its next step from LDS: a quarter of the metric bytes cross the L2 -> CU fabric, which is
    // what bounds this kernel (25 metric terms against 5 field values per point).
#if !defined(FV3_HOST_EMU)
    constexpr bool SHARE = CSWF_NW > 1;
#else
    constexpr bool SHARE = false;
#endif
    const int wv = blk.by;
    (void)wv;
    Real *lds = (Real *)smem_;
    Real sh[(CSWF_NMET + CSWF_NW - 1) / CSWF_NW][FV3_LPT];
#define CSW_LDSH(i, n, ptr, pp) \
  if ((i % CSWF_NW) == wv) sh[i / CSWF_NW][l] = FV3_EL(ptr + m2, pp);
#define CSW_WRSH(i, n, ptr, pp) \
  if ((i % CSWF_NW) == wv) lds[(slot_ * CSWF_NMET + i) * FV3_WAVE + lane] = sh[i / CSWF_NW][l];
#define CSW_RDSH(i, n, ptr, pp) mn_##n[l] = lds[(slot_ * CSWF_NMET + i) * FV3_WAVE + lane];
    FV3_LANES(blk, lane, l) {
      const int lc = c0 - 3 + lane, lcc = lc < Msd ? Msd : lc < Led ? lc : Led;  // (lane 0 of a first strip: no such column, and nothing owned reads it)
      pcol[l] = pbase + (unsigned)lcc;
      const bool own = live && lane >= 3 && lane <= CSWF_OUT + 2;
      c_r0[l] = own && lc <= rc.i_hi;
      c_rd[l] = own && lc <= rc.i_hi - 1;
      c_vr[l] = own && lc >= re_i0 && lc <= rc.i_hi;
      c_div[l] = own && lc >= 1 && lc <= rc.i_hi;
      c_re[l] = own && lc >= re_i0 && lc <= rc.i_hi - 1;
      c_kb[l] = lc <= rc.i_lo + 1 || lc >= rc.i_hi - 2;
      c_vb[l] = lc <= rc.i_lo + 2 || lc >= rc.i_hi - 1;
      u0[l] = u1[l] = u2[l] = u3[l] = v0[l] = v1[l] = v2[l] = v3[l] = t0[l] = t1[l] = t2[l] = t3[l] = (Real)0;
      d0[l] = d1[l] = d2[l] = p0[l] = p1[l] = p2[l] = w0[l] = w1[l] = w2[l] = (Real)0;
      s_ut[l] = s_v[l] = s_ua[l] = s_uf[l] = s_uc[l] = s_d[l] = s_p[l] = s_w[l] = s_f1[l] = s_f[l] = s_f2[l] = s_pvd[l] = s_ke[l] = s_vo[l] = s_s3[l] = s_c3[l] = (Real)0;
      va_prev[l] = vf_prev[l] = s4_prev[l] = c4_prev[l] = s2_prev[l] = vt_prev[l] = vc_prev[l] = cv_prev[l] = cu_prev[l] = uc_prev[l] = dxc_prev[l] = (Real)0;
      g1_lo[l] = g_lo[l] = g2_lo[l] = ke_prev[l] = vo_prev[l] = (Real)0;
      uav[l] = vav[l] = ucv_[l] = vcv_[l] = ufv[l] = utq[l] = vtq[l] = ke_w[l] = kev_[l] = vov_[l] = (Real)0;
#define CSW_ZERO(i, n, ptr, pp) mc_##n[l] = (Real)0;
      CSW_METT(CSW_ZERO)
#undef CSW_ZERO
      // (delp = 1 in the rows not yet loaded: the warm-up steps divide by the transported air mass)
      d0[l] = d1[l] = d2[l] = (Real)1;
      const int R0_ = ja - 3;
      auto row = [&](int r) { return (unsigned)((r < Msd ? Msd : r > r_end ? r_end : r) * sj32); };
      nfu[0][l] = FV3_EL(ub, pcol[l] + row(R0_));
      nfv[0][l] = FV3_EL(vb, pcol[l] + row(R0_));
      nfu[1][l] = FV3_EL(ub, pcol[l] + row(R0_ + 1));
      nfv[1][l] = FV3_EL(vb, pcol[l] + row(R0_ + 1));
      nfd[0][l] = FV3_EL(db, pcol[l] + row(R0_ - 1));
      nfp[0][l] = FV3_EL(pb_, pcol[l] + row(R0_ - 1));
      nfw[0][l] = FV3_EL(wb, pcol[l] + row(R0_ - 1));
      nfd[1][l] = FV3_EL(db, pcol[l] + row(R0_));
      nfp[1][l] = FV3_EL(pb_, pcol[l] + row(R0_));
      nfw[1][l] = FV3_EL(wb, pcol[l] + row(R0_));
      CSW_LOAD_MET(R0_)
      if constexpr (SHARE) {
        CSW_ROWS(R0_ + 1)
        CSW_METT(CSW_LDSH)
      }
    }
#ifdef CSWF_ROLLED
    for (int R = ja - 3; R <= r_end; ++R) {
      const int h_ = 0;
#else
    for (int R2_ = ja - 3; R2_ <= r_end; R2_ += 2) {
#pragma unroll
    for (int h_ = 0; h_ < 2; ++h_) {
      const int R = R2_ + h_;
      if (R > r_end) break;
#endif
      const unsigned rn2 = (unsigned)((R + 2 < r_end ? R + 2 : r_end) * sj32), rn1 = (unsigned)((R + 1 < r_end ? R + 1 : r_end) * sj32);
      const int jd = R - 2, jv = R - 1, je = R - 3;
      const bool seg_d = jd >= ja && jd <= jb, seg_v = jv >= ja && jv <= jb, seg_e = je >= ja && je <= jb;
      const bool r_rd = seg_d && jd <= rc.j_hi - 1, r_vr = seg_d && jd >= re_j0, r_div = seg_d && jd >= 1;
      const bool r_re_d = seg_d && jd >= re_j0 && jd <= rc.j_hi - 1, r_re_v = seg_v && jv >= re_j0 && jv <= rc.j_hi - 1, r_re_e = seg_e && je >= re_j0 && je <= rc.j_hi - 1;
      const bool kb_row = jd <= rc.j_lo + 1 || jd >= rc.j_hi - 2, vb_row = jd <= rc.j_lo + 2 || jd >= rc.j_hi - 1;
      // ---- phase 1: rotate the windows, the rows in flight, the metric terms; fetch ahead; utmp of row R-2
      FV3_LANES(blk, lane, l) {
        u0[l] = u1[l];
        u1[l] = u2[l];
        u2[l] = u3[l];
        u3[l] = nfu[h_][l];
        v0[l] = v1[l];
        v1[l] = v2[l];
        v2[l] = v3[l];
        v3[l] = nfv[h_][l];
        d0[l] = d1[l];
        d1[l] = d2[l];
        d2[l] = nfd[h_][l];
        p0[l] = p1[l];
        p1[l] = p2[l];
        p2[l] = nfp[h_][l];
        w0[l] = w1[l];
        w1[l] = w2[l];
        w2[l] = nfw[h_][l];
#ifdef CSWF_ROLLED
        nfu[0][l] = nfu[1][l];
        nfv[0][l] = nfv[1][l];
        nfd[0][l] = nfd[1][l];
        nfp[0][l] = nfp[1][l];
        nfw[0][l] = nfw[1][l];
        nfu[1][l] = FV3_EL(ub, pcol[l] + rn2);
        nfv[1][l] = FV3_EL(vb, pcol[l] + rn2);
        nfd[1][l] = FV3_EL(db, pcol[l] + rn1);
        nfp[1][l] = FV3_EL(pb_, pcol[l] + rn1);
        nfw[1][l] = FV3_EL(wb, pcol[l] + rn1);
#else
        nfu[h_][l] = FV3_EL(ub, pcol[l] + rn2);
        nfv[h_][l] = FV3_EL(vb, pcol[l] + rn2);
        nfd[h_][l] = FV3_EL(db, pcol[l] + rn1);
        nfp[h_][l] = FV3_EL(pb_, pcol[l] + rn1);
        nfw[h_][l] = FV3_EL(wb, pcol[l] + rn1);
#endif
#define CSW_ROT(i, n, ptr, pp) mc_##n[l] = mn_##n[l];
        CSW_METT(CSW_ROT)
#undef CSW_ROT
        if constexpr (SHARE) {
          const int slot_ = (R + 1) & 1;
          CSW_METT(CSW_WRSH)  // this wave's share of step R+1 (fetched during step R-1)
          {
            CSW_ROWS(R + 2)
            CSW_METT(CSW_LDSH)
          }
          blk.group_sync();
          CSW_METT(CSW_RDSH)
        } else {
          CSW_LOAD_MET(R + 1)
        }
        s_ut[l] = CSW_A2 * (u0[l] + u3[l]) + CSW_A1 * (u1[l] + u2[l]);
        s_v[l] = v3[l];
        s_d[l] = d1[l];
        s_p[l] = p1[l];
        s_w[l] = w1[l];
        s_s3[l] = mc_s3[l];
        s_c3[l] = mc_c3[l];
      }
      // ---- phase 2: vtmp of row R; ua / va / uc0 / ut of row R-2, vc0 / vt of row R-1; the west-face fluxes of the transport
      FV3_LANES(blk, lane, l) {
        const Real vt_new = CSW_A2 * (FV3_LANE_SHR(1, s_v, l, lane) + FV3_LANE_SHL(2, s_v, l, lane)) + CSW_A1 * (s_v[l] + FV3_LANE_SHL(1, s_v, l, lane));
        t0[l] = t1[l];
        t1[l] = t2[l];
        t2[l] = t3[l];
        t3[l] = vt_new;
        const Real ut0 = s_ut[l], vt0 = t1[l];
        const Real ua_ = (ut0 - vt0 * mc_cs[l]) * mc_r2[l], va_ = (vt0 - ut0 * mc_cs[l]) * mc_r2[l];
        const Real ucv = CSW_A2 * (FV3_LANE_SHR(2, s_ut, l, lane) + FV3_LANE_SHL(1, s_ut, l, lane)) + CSW_A1 * (FV3_LANE_SHR(1, s_ut, l, lane) + ut0);
        const Real utv = (ucv - v1[l] * mc_cu[l]) * mc_ru[l];
        const Real vcv = CSW_A2 * (t0[l] + t3[l]) + CSW_A1 * (t1[l] + t2[l]);
        const Real vtv = (vcv - u2[l] * mc_cv[l]) * mc_rv[l];
        const Real s3w = FV3_LANE_SHR(1, s_s3, l, lane);
        const Real ut_o = utv > (Real)0 ? dt2 * utv * mc_dy[l] * s3w : dt2 * utv * mc_dy[l] * mc_s1[l];
        const Real vt_o = vtv > (Real)0 ? dt2 * vtv * mc_dx[l] * mc_s4[l] : dt2 * vtv * mc_dx[l] * mc_s2[l];
        const unsigned pd = pcol[l] + (unsigned)(jd * sj32), pv = pcol[l] + (unsigned)(jv * sj32);
        if (c_r0[l] && seg_d) {
          if (sua || c_kb[l] || kb_row) {
            CSWF_STE(ua + b, pd, ua_);
            CSWF_STE(va + b, pd, va_);
          }
          CSWF_STE(ut + b, pd, ut_o);
          if (!(c_re[l] && r_re_d)) CSWF_STE(uc + b, pd, ucv);  // the rim of the rectangle: the stage E kernel finishes it
        }
        if (c_r0[l] && seg_v) {
          CSWF_STE(vt + b, pv, vt_o);
          if (!(c_re[l] && r_re_v)) CSWF_STE(vc + b, pv, vcv);
        }
        // west-face fluxes of the cell (lc, R-2): upwind cell lc-1 or lc
        {
          const Real dw = FV3_LANE_SHR(1, s_d, l, lane), pw = FV3_LANE_SHR(1, s_p, l, lane), ww = FV3_LANE_SHR(1, s_w, l, lane);
          const bool up = ut_o > (Real)0;
          const Real f1 = ut_o * (up ? dw : d1[l]);
          s_f1[l] = f1;
          s_f[l] = f1 * (up ? pw : p1[l]);
          s_f2[l] = f1 * (up ? ww : w1[l]);
        }
        // u-face term of the corner divergence at (lc, R-2): cos / sin sums over the rows R-3 and R-2
        ufv[l] = (u1[l] - (Real)0.25 * (va_prev[l] + va_) * (c4_prev[l] + mc_c2[l])) * mc_dyc[l] * (Real)0.5 * (s4_prev[l] + s2_prev[l]);
        s_uf[l] = ufv[l];
        uav[l] = ua_;
        vav[l] = va_;
        s_ua[l] = ua_;
        ucv_[l] = ucv;
        s_uc[l] = ucv;
        vcv_[l] = vcv;
        utq[l] = ut_o;
        vtq[l] = vt_o;
        s_pvd[l] = vc_prev[l] * mc_dyc[l];
        ke_w[l] = FV3_LANE_SHR(1, s_ke, l, lane);  // ke(lc-1, R-3) (before phase 3 overwrites the row)
        va_prev[l] = va_;
        c4_prev[l] = mc_c4[l];
        s4_prev[l] = mc_s4[l];
        s2_prev[l] = mc_s2[l];
      }
      // ---- phase 3: the divergence, the transport, the kinetic energy and the vorticity of row R-2
      FV3_LANES(blk, lane, l) {
        const unsigned pd = pcol[l] + (unsigned)(jd * sj32);
        if (do_div) {
          const Real s3w = FV3_LANE_SHR(1, s_s3, l, lane), c3w = FV3_LANE_SHR(1, s_c3, l, lane);
          const Real vf = (v1[l] - (Real)0.25 * (FV3_LANE_SHR(1, s_ua, l, lane) + uav[l]) * (c3w + mc_c1[l])) * mc_dxc[l] * (Real)0.5 * (s3w + mc_s1[l]);
          const Real dv = vf_prev[l] - vf + FV3_LANE_SHR(1, s_uf, l, lane) - ufv[l];
          if (c_div[l] && r_div) CSWF_STE(divgd + b, pd, mc_rac[l] * dv);
          vf_prev[l] = vf;
        }
        // transport: x fluxes at the faces lc (own) and lc + 1 (the neighbouring lane's), y fluxes at the faces R-2 (carried) and R-1
        const Real fx1e = FV3_LANE_SHL(1, s_f1, l, lane), fxe = FV3_LANE_SHL(1, s_f, l, lane), fx2e = FV3_LANE_SHL(1, s_f2, l, lane);
        const Real vth = vtq[l];
        const bool upn = vth > (Real)0;
        const Real g1_hi = vth * (upn ? d1[l] : d2[l]);
        const Real g_hi = g1_hi * (upn ? p1[l] : p2[l]);
        const Real g2_hi = g1_hi * (upn ? w1[l] : w2[l]);
        const Real dpc = d1[l] + (s_f1[l] - fx1e + g1_lo[l] - g1_hi) * mc_ra[l];
        if (c_rd[l] && r_rd) {
          CSWF_STE(delpc + b, pd, dpc);
          CSWF_STE(ptc + b, pd, (p1[l] * d1[l] + (s_f[l] - fxe + g_lo[l] - g_hi) * mc_ra[l]) / dpc);
          CSWF_STE(omga + b, pd, (w1[l] * d1[l] + (s_f2[l] - fx2e + g2_lo[l] - g2_hi) * mc_ra[l]) / dpc);
        }
        g1_lo[l] = g1_hi;
        g_lo[l] = g_hi;
        g2_lo[l] = g2_hi;
        // kinetic energy of the cell, absolute vorticity of the corner
        const Real uce = FV3_LANE_SHL(1, s_uc, l, lane);  // (read outside the selection: a shuffle needs every source lane active)
        const Real kev = uav[l] > (Real)0 ? ucv_[l] : uce;
        const Real vov = vav[l] > (Real)0 ? vc_prev[l] : vcv_[l];
        const Real kv = (Real)0.5 * dt2 * (uav[l] * kev + vav[l] * vov);
        const Real vo = uc_prev[l] * dxc_prev[l] - ucv_[l] * mc_dxc[l] - FV3_LANE_SHR(1, s_pvd, l, lane) + vc_prev[l] * mc_dyc[l];
        const Real vr = mc_fc[l] + mc_rac[l] * vo;
        if (c_rd[l] && r_rd && (c_kb[l] || kb_row)) CSWF_STE(ke + b, pd, kv);
        if (c_vr[l] && r_vr && (c_vb[l] || vb_row)) CSWF_STE(vort + b, pd, vr);
        kev_[l] = kv;
        vov_[l] = vr;
        s_vo[l] = vr;
      }
      // ---- phase 4: the time-centred vc of row R-2 and uc of row R-3
      FV3_LANES(blk, lane, l) {
        const Real kv = kev_[l], vr = vov_[l];
        {
          const Real ucp = uc_prev[l];
          const Real fy1 = dt2 * (v0[l] - ucp * cu_prev[l]) / mc_su[l];
          const Real fyv = fy1 > (Real)0 ? vo_prev[l] : vr;
          const Real un = ucp + fy1 * fyv + mc_rdx[l] * (ke_w[l] - ke_prev[l]);
          if (c_re[l] && r_re_e) CSWF_STE(uc + b, pcol[l] + (unsigned)(je * sj32), un);
        }
        {
          const Real vcp = vc_prev[l];
          const Real fx1 = dt2 * (u1[l] - vcp * cv_prev[l]) / mc_sv[l];
          const Real vre = FV3_LANE_SHL(1, s_vo, l, lane);
          const Real fxv = fx1 > (Real)0 ? vr : vre;
          const Real vn = vcp - fx1 * fxv + mc_rdy[l] * (ke_prev[l] - kv);
          if (c_re[l] && r_re_d) CSWF_STE(vc + b, pcol[l] + (unsigned)(jd * sj32), vn);
        }
        s_ke[l] = kv;
        ke_prev[l] = kv;
        vo_prev[l] = vr;
        uc_prev[l] = ucv_[l];
        dxc_prev[l] = mc_dxc[l];
        cu_prev[l] = mc_cu[l];
        vc_prev[l] = vcv_[l];
        cv_prev[l] = mc_cv[l];
      }
    }
#ifndef CSWF_ROLLED
    }
#endif
#undef CSW_LOAD_MET
#undef CSW_ROWS
#undef CSW_LD1
#undef CSW_LDSH
#undef CSW_WRSH
#undef CSW_RDSH
#undef CSW_METT
  });
}

extern "C" int fv3_c_sw(fv3_ctx *c, const fv3_field *delp_, const fv3_field *pt_, const fv3_field *u_, const fv3_field *v_, const fv3_field *w_,
                        const fv3_field *uc_, const fv3_field *vc_, const fv3_field *ua_, const fv3_field *va_, const fv3_field *ut_, const fv3_field *vt_,
                        const fv3_field *divgd_, const fv3_field *omga_, const fv3_field *delpc_, const fv3_field *ptc_, double dt2d, void *stream) {
  if (!c) return FV3_ERR_ARG;
  FV3_FIELD(delp, delp_) FV3_FIELD(pt, pt_) FV3_FIELD(u, u_) FV3_FIELD(v, v_) FV3_FIELD(w, w_) FV3_FIELD(uc, uc_) FV3_FIELD(vc, vc_)
  FV3_FIELD(ua, ua_) FV3_FIELD(va, va_) FV3_FIELD(ut, ut_) FV3_FIELD(vt, vt_) FV3_FIELD(divgd, divgd_) FV3_FIELD(omga, omga_)
  FV3_FIELD(delpc, delpc_) FV3_FIELD(ptc, ptc_)
  const Geo g = c->g;
  fv3_stream_t s = (fv3_stream_t)stream;
  const Real dt2 = (Real)dt2d;
  const int nz1 = g.nz - 1;
  const int nord = c->cfg.nord;
  Real *ke = c->scratch[SC_A], *vort = c->scratch[SC_B];

  // (A) contravariant A-grid winds on is-2..ie+2
  // Where stage B's two-row interior kernel runs (below) it also produces ua / va -- it holds utmp(i, j) and vtmp(i, j)
  // of its points anyway -- so this per-point form then only covers the boundary windows and the outer ring.
  // A/B switches, same results: FV3_CSW_B_GENERIC: every point by the generic per-point stage kernels; FV3_CSW_MARCH=0: the round-1
  // form (two-row stage-B kernel on the interior rectangle, the other stages as full stage kernels); FV3_CSW_MARCH=abc: stages
  // A - C of the interior as a marching kernel, D and E as stage kernels; default: the whole interior as one marching kernel
  const bool b_split = g.nx >= 16 && g.ny >= 16 && !getenv("FV3_CSW_B_GENERIC");
  const char *march_env = getenv("FV3_CSW_MARCH");
  const bool march = b_split && !(march_env && !strcmp(march_env, "0"));
  const bool fused = march && !(march_env && !strcmp(march_env, "abc"));
  // levels one thread of the stage kernels walks (the metric terms of its point are read once for them): FV3_KC for launches
  // over whole sub-domains; 2 when they only cover the boundary windows, which are too few points to fill the chip otherwise
  static const int win_kc = getenv("FV3_CSW_WIN_KC") ? atoi(getenv("FV3_CSW_WIN_KC")) : 2;
  const int KC = fused ? win_kc : FV3_KC;
  const int nkc = (nz1 + KC) / KC;
  // interior rectangle of sub-domain t: columns [i_lo, i_hi], rows [j_lo, j_hi] (the two-row kernel needs an even row count)
  auto b_rect = [=] FV3_HD(int fl) {
    CswRect r = csw_rect(fl, g.nx, g.ny, g.npx, g.npy);
    if (!march) r.j_hi = r.j_lo + 2 * ((r.j_hi - r.j_lo + 1) / 2) - 1;
    return r;
  };
  auto stage_a = [=] FV3_HD(int t, int kp, int i, int j) {
    const int fl = g.flags[t];
    if (b_split) {
      const CswRect rc = b_rect(fl);
      if (i >= rc.i_lo && i <= rc.i_hi && j >= rc.j_lo && j <= rc.j_hi) return;  // done by the interior kernel
    }
    const long m2 = t * g.st2;
    const unsigned p = IX(i, j);
    const Real cs = (g.cosa_s + m2)[p], rs2 = (g.rsin2 + m2)[p];
#pragma unroll 1
    for (int kk = 0; kk < KC; ++kk) {
      const int k = KC * kp + kk;
      if (k > nz1) break;
      const long b = t * g.st + k * g.sk;
      D2A d{g, u + b, v + b, (fl & FV3_W) != 0, (fl & FV3_E) != 0, (fl & FV3_S) != 0, (fl & FV3_N) != 0};
      const Real ut_ = d.utmp0(i, j), vt_ = d.vtmp0(i, j);
      (ua + b)[p] = (ut_ - vt_ * cs) * rs2;
      (va + b)[p] = (vt_ - ut_ * cs) * rs2;
    }
  };
  // Round 5: the boundary windows of stages A and B read u / v (and, within two cells of a tile edge, the window's own ua / va) and write
  // window cells only; the interior march reads the five inputs and writes interior cells only.  Neither waits for the other, so the eight
  // small window launches (1.0 ms of a few dozen waves each) go to the auxiliary stream and run BESIDE the march; stage C -- whose windows
  // read the march's ut / vt across the rim -- joins them.  FV3_CSW_WIN_OVERLAP=0: in program order (A/B; same values).  Events 0 = fork, 1 = join.
  static const bool win_overlap = !(getenv("FV3_CSW_WIN_OVERLAP") && getenv("FV3_CSW_WIN_OVERLAP")[0] == '0');
  fv3_stream_t sw_ = (fused && b_split && win_overlap) ? fv3_aux(c, s) : s;
  if (sw_ != s) {
    fv3_signal(c, s, 0);
    fv3_wait(c, sw_, 0);
  }
  if (b_split) {
    // the four windows along the sub-domain boundary (launch_frame: W / E as narrow 8 x 32 workgroups, S / N as they lie)
    const int e0 = g.nx - 3;
    launch_frame(c, sw_, Frame{{Box{-1, 5, -1, g.ny + 2, 0, nkc - 1}, Box{e0, e0 + 5, -1, g.ny + 2, 0, 0}, Box{6, g.nx - 4, -1, 5, 0, 0}, Box{6, g.nx - 4, g.ny - 4, g.ny + 2, 0, 0}}}, stage_a);
  } else {
    launch3(c, s, Box{-1, g.nx + 2, -1, g.ny + 2, 0, nkc - 1}, stage_a);
  }

  // (B) C-grid winds + contravariant ut, vt (already scaled: dt2 * ut * dy * sin_sg)
  // The generic per-point form does 32 u / v loads per point and is bound by them (halving them in a timing
  // experiment took 2.5 ms off c_sw).  Points whose whole stencil uses the interior formulas are therefore done by
  // a lean kernel in which a thread owns TWO rows and shares the utmp / vtmp rows between them (20 loads per point);
  // the generic form then only runs on four windows along the sub-domain boundary (and skips the interior).
  if (fused) {
    // (FV3_SEQ_UAVA=every: ua / va stored in full by every sub-step, A/B; read per call)
    const char *uae = getenv("FV3_SEQ_UAVA");
    csw_fused_stream(c, s, u, v, delp, pt, w, ua, va, uc, vc, ut, vt, divgd, delpc, ptc, omga, ke, vort, dt2, nord > 0, c->seq_uava_thin && !(uae && !strcmp(uae, "every")));
  } else if (march) {
    csw_abc_stream(c, s, u, v, ua, va, uc, vc, ut, vt, divgd, dt2, nord > 0);
  } else if (b_split) {
    launch3(c, s, Box{0, g.nx + 1, 0, (g.ny + 2) / 2, 0, nkc - 1}, [=] FV3_HD(int t, int kp, int i_, int jp) {
      const CswRect rc = b_rect(g.flags[t]);
      if (i_ < rc.i_lo || i_ > rc.i_hi || jp >= (rc.j_hi - rc.j_lo + 1) / 2) return;
      const long m2 = t * g.st2;
      int i = i_, j = rc.j_lo + 2 * jp;
      const int j_base = j;
      // metric terms of the two points, shared by the levels of the chunk
      Real mcu[2], mru[2], mdy[2], ms3[2], ms1[2], mcv[2], mrv[2], mdx[2], ms4[2], ms2[2], mcs[2], mr2[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const unsigned q = IX(i, j + r);
        mcu[r] = (g.cosa_u + m2)[q];
        mru[r] = (g.rsin_u + m2)[q];
        mdy[r] = (g.dy + m2)[q];
        ms3[r] = (g.sin_sg3 + m2)[IX(i - 1, j + r)];
        ms1[r] = (g.sin_sg1 + m2)[q];
        mcv[r] = (g.cosa_v + m2)[q];
        mrv[r] = (g.rsin_v + m2)[q];
        mdx[r] = (g.dx + m2)[q];
        ms4[r] = (g.sin_sg4 + m2)[IX(i, j + r - 1)];
        ms2[r] = (g.sin_sg2 + m2)[q];
        mcs[r] = (g.cosa_s + m2)[q];
        mr2[r] = (g.rsin2 + m2)[q];
      }
#pragma unroll 1
      for (int kk = 0; kk < KC; ++kk) {
        const int k = KC * kp + kk;
        if (k > nz1) break;
        i = i_;
        j = j_base;
        FV3_LAUNDER(i);
        FV3_LAUNDER(j);
        const long b = t * g.st + k * g.sk;
        const Real *ul = u + b, *vl = v + b;
        // utmp at columns i-2 .. i+1 for the two rows: five u rows j-1 .. j+3 per column
        Real ut_[4][2], uc0[2] = {(Real)0, (Real)0};  // uc0: u(i, j + r) for the v part
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int ii = i - 2 + a;
          const Real um = ul[IX(ii, j - 1)], u0 = ul[IX(ii, j)], u1 = ul[IX(ii, j + 1)], u2 = ul[IX(ii, j + 2)], u3 = ul[IX(ii, j + 3)];
          ut_[a][0] = CSW_A2 * (um + u2) + CSW_A1 * (u0 + u1);
          ut_[a][1] = CSW_A2 * (u0 + u3) + CSW_A1 * (u1 + u2);
          if (a == 2) {
            uc0[0] = u0;
            uc0[1] = u1;
          }
        }
        // vtmp at rows j-2 .. j+2: four v columns i-1 .. i+2 per row
        Real vt_[5], vc0[2] = {(Real)0, (Real)0};  // vc0: v(i, j + r) for the u part
#pragma unroll
        for (int a = 0; a < 5; ++a) {
          const int jj = j - 2 + a;
          const Real vm = vl[IX(i - 1, jj)], v0 = vl[IX(i, jj)], v1 = vl[IX(i + 1, jj)], v2 = vl[IX(i + 2, jj)];
          vt_[a] = CSW_A2 * (vm + v2) + CSW_A1 * (v0 + v1);
          if (a == 2) vc0[0] = v0;
          if (a == 3) vc0[1] = v0;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const unsigned p = IX(i, j + r);
          {  // stage A's ua / va of the point: utmp(i, j) and vtmp(i, j) are at hand
            const Real ut0 = ut_[2][r], vt0 = vt_[2 + r];
            (ua + b)[p] = (ut0 - vt0 * mcs[r]) * mr2[r];
            (va + b)[p] = (vt0 - ut0 * mcs[r]) * mr2[r];
          }
          const Real ucv = CSW_A2 * (ut_[0][r] + ut_[3][r]) + CSW_A1 * (ut_[1][r] + ut_[2][r]);
          const Real utv = (ucv - vc0[r] * mcu[r]) * mru[r];
          (uc + b)[p] = ucv;
          (ut + b)[p] = utv > (Real)0 ? dt2 * utv * mdy[r] * ms3[r] : dt2 * utv * mdy[r] * ms1[r];
          const Real vcv = CSW_A2 * (vt_[r] + vt_[r + 3]) + CSW_A1 * (vt_[r + 1] + vt_[r + 2]);
          const Real vtv = (vcv - uc0[r] * mcv[r]) * mrv[r];
          (vc + b)[p] = vcv;
          (vt + b)[p] = vtv > (Real)0 ? dt2 * vtv * mdx[r] * ms4[r] : dt2 * vtv * mdx[r] * ms2[r];
        }
      }
    });
  }
  auto stage_b = [=] FV3_HD(int t, int kp, int i_, int j_) {
    const int fl = g.flags[t];
    if (b_split) {
      const CswRect rc = b_rect(fl);
      if (i_ >= rc.i_lo && i_ <= rc.i_hi && j_ >= rc.j_lo && j_ <= rc.j_hi) return;  // done by the interior kernel
    }
    const long m2 = t * g.st2;
    const bool W = fl & FV3_W, E = fl & FV3_E, S = fl & FV3_S, N = fl & FV3_N;
    const int npx = g.npx, npy = g.npy;
    int i = i_, j = j_;
    unsigned p = IX(i, j);
    // the metric terms of the point, shared by the levels of the chunk (tile-edge clauses read their own)
    const Real m_cosa_u = (g.cosa_u + m2)[p], m_rsin_u = (g.rsin_u + m2)[p], m_dy = (g.dy + m2)[p], m_s3w = (g.sin_sg3 + m2)[IX(i - 1, j)], m_s1 = (g.sin_sg1 + m2)[p];
    const Real m_cosa_v = (g.cosa_v + m2)[p], m_rsin_v = (g.rsin_v + m2)[p], m_dx = (g.dx + m2)[p], m_s4s = (g.sin_sg4 + m2)[IX(i, j - 1)], m_s2 = (g.sin_sg2 + m2)[p];
#pragma unroll 1
    for (int kk = 0; kk < KC; ++kk) {
    const int k = KC * kp + kk;
    if (k > nz1) break;
    i = i_;
    j = j_;
    FV3_LAUNDER(i);
    FV3_LAUNDER(j);
    p = IX(i, j);
    const long b = t * g.st + k * g.sk;
    D2A d{g, u + b, v + b, W, E, S, N};
    const Real *uaa = ua + b, *vaa = va + b;
    // ua / va with the corner fixes the edge interpolation reads
    auto UA = [&](int ii, int jj) -> Real {
      if (S && jj == 0) {
        if (W && ii == -1) return -vaa[IX(0, 2)];
        if (W && ii == 0) return -vaa[IX(0, 1)];
        if (E && ii == npx) return vaa[IX(npx, 1)];
        if (E && ii == npx + 1) return vaa[IX(npx, 2)];
      }
      if (N && jj == npy) {
        if (E && ii == npx) return -vaa[IX(npx, npy - 1)];
        if (E && ii == npx + 1) return -vaa[IX(npx, npy - 2)];
        if (W && ii == -1) return vaa[IX(0, npy - 2)];
        if (W && ii == 0) return vaa[IX(0, npy - 1)];
      }
      return uaa[IX(ii, jj)];
    };
    auto VA = [&](int ii, int jj) -> Real {
      if (W && ii == 0) {
        if (S && jj == -1) return -uaa[IX(2, 0)];
        if (S && jj == 0) return -uaa[IX(1, 0)];
        if (N && jj == npy) return uaa[IX(1, npy)];
        if (N && jj == npy + 1) return uaa[IX(2, npy)];
      }
      if (E && ii == npx) {
        if (S && jj == 0) return uaa[IX(npx - 1, 0)];
        if (S && jj == -1) return uaa[IX(npx - 2, 0)];
        if (N && jj == npy) return -uaa[IX(npx - 1, npy)];
        if (N && jj == npy + 1) return -uaa[IX(npx - 2, npy)];
      }
      return vaa[IX(ii, jj)];
    };
    if (j <= g.ny + 1) {  // uc, ut on i = is-1..ie+2, j = js-1..je+1
      Real ucv = (Real)0, utv;
      bool edge = false;
      if (W && i <= 2) {
        edge = true;
        if (i == 0)
          ucv = CSW_C1 * d.utmp_x(-2, j) + CSW_C2 * d.utmp_x(-1, j) + CSW_C3 * d.utmp_x(0, j);
        else if (i == 2)
          ucv = CSW_C1 * d.utmp_x(3, j) + CSW_C2 * d.utmp_x(2, j) + CSW_C3 * d.utmp_x(1, j);
      } else if (E && i >= npx - 1) {
        edge = true;
        if (i == npx - 1)
          ucv = CSW_C1 * d.utmp_x(npx - 3, j) + CSW_C2 * d.utmp_x(npx - 2, j) + CSW_C3 * d.utmp_x(npx - 1, j);
        else if (i == npx + 1)
          ucv = CSW_C3 * d.utmp_x(npx, j) + CSW_C2 * d.utmp_x(npx + 1, j) + CSW_C1 * d.utmp_x(npx + 2, j);
      }
      if (edge && (i == 1 || i == npx)) {
        utv = edge_interp4(UA(i - 2, j), UA(i - 1, j), UA(i, j), UA(i + 1, j), (g.dxa + m2)[IX(i - 2, j)], (g.dxa + m2)[IX(i - 1, j)], (g.dxa + m2)[IX(i, j)],
                           (g.dxa + m2)[IX(i + 1, j)]);
        ucv = utv > (Real)0 ? utv * m_s3w : utv * m_s1;
      } else {
        if (!edge) ucv = CSW_A2 * (d.utmp_x(i - 2, j) + d.utmp_x(i + 1, j)) + CSW_A1 * (d.utmp_x(i - 1, j) + d.utmp_x(i, j));
        utv = (ucv - (v + b)[p] * m_cosa_u) * m_rsin_u;
      }
      (uc + b)[p] = ucv;
      (ut + b)[p] = utv > (Real)0 ? dt2 * utv * m_dy * m_s3w : dt2 * utv * m_dy * m_s1;
    }
    if (i <= g.nx + 1) {  // vc, vt on i = is-1..ie+1, j = js-1..je+2
      Real vcv = (Real)0, vtv;
      bool edge = false;
      if (S && j <= 2) {
        edge = true;
        if (j == 0)
          vcv = CSW_C1 * d.vtmp_y(i, -2) + CSW_C2 * d.vtmp_y(i, -1) + CSW_C3 * d.vtmp_y(i, 0);
        else if (j == 2)
          vcv = CSW_C1 * d.vtmp_y(i, 3) + CSW_C2 * d.vtmp_y(i, 2) + CSW_C3 * d.vtmp_y(i, 1);
      } else if (N && j >= npy - 1) {
        edge = true;
        if (j == npy - 1)
          vcv = CSW_C1 * d.vtmp_y(i, npy - 3) + CSW_C2 * d.vtmp_y(i, npy - 2) + CSW_C3 * d.vtmp_y(i, npy - 1);
        else if (j == npy + 1)
          vcv = CSW_C1 * d.vtmp_y(i, npy + 2) + CSW_C2 * d.vtmp_y(i, npy + 1) + CSW_C3 * d.vtmp_y(i, npy);
      }
      if (edge && (j == 1 || j == npy)) {
        vtv = edge_interp4(VA(i, j - 2), VA(i, j - 1), VA(i, j), VA(i, j + 1), (g.dya + m2)[IX(i, j - 2)], (g.dya + m2)[IX(i, j - 1)], (g.dya + m2)[IX(i, j)],
                           (g.dya + m2)[IX(i, j + 1)]);
        vcv = vtv > (Real)0 ? vtv * m_s4s : vtv * m_s2;
      } else {
        if (!edge) vcv = CSW_A2 * (d.vtmp_y(i, j - 2) + d.vtmp_y(i, j + 1)) + CSW_A1 * (d.vtmp_y(i, j - 1) + d.vtmp_y(i, j));
        vtv = (vcv - (u + b)[p] * m_cosa_v) * m_rsin_v;
      }
      (vc + b)[p] = vcv;
      (vt + b)[p] = vtv > (Real)0 ? dt2 * vtv * m_dx * m_s4s : dt2 * vtv * m_dx * m_s2;
    }
    }
  };
  {
    const Box nat{0, g.nx + 2, 0, g.ny + 2, 0, nkc - 1};
    if (b_split) {
      // (one right-sized launch per window: launch3w sizes every window's grid for the largest one)
      // (the 6-column W / E windows run as 8-column x 32-row workgroups: launch_frame)
      const int e0 = g.nx - 3;
      launch_frame(c, sw_, Frame{{Box{0, 5, 0, g.ny + 2, 0, nkc - 1}, Box{e0, e0 + 5, 0, g.ny + 2, 0, 0}, Box{6, g.nx - 4, 0, 5, 0, 0}, Box{6, g.nx - 4, g.ny - 4, g.ny + 2, 0, 0}}}, stage_b);
    } else {
      launch3(c, s, nat, stage_b);
    }
  }
  if (sw_ != s) {
    fv3_signal(c, sw_, 1);
    fv3_wait(c, s, 1);
  }

  // the in-place corner fixes of ua / va the reference leaves behind (checkpointed as uad / vad)
  launch3(c, s, Box{1, 1, 1, 1, 0, nz1}, [=] FV3_HD(int t, int k, int, int) {
    const int fl = g.flags[t];
    const long b = t * g.st + k * g.sk;
    const bool W = fl & FV3_W, E = fl & FV3_E, S = fl & FV3_S, N = fl & FV3_N;
    const int npx = g.npx, npy = g.npy;
    Real *uaa = ua + b, *vaa = va + b;
    Real n_ua[8], n_va[8];
    // read everything first (sources are never targets of the other set's sources)
    if (W && S) { n_ua[0] = -vaa[IX(0, 2)]; n_ua[1] = -vaa[IX(0, 1)]; }
    if (E && S) { n_ua[2] = vaa[IX(npx, 1)]; n_ua[3] = vaa[IX(npx, 2)]; }
    if (E && N) { n_ua[4] = -vaa[IX(npx, npy - 1)]; n_ua[5] = -vaa[IX(npx, npy - 2)]; }
    if (W && N) { n_ua[6] = vaa[IX(0, npy - 2)]; n_ua[7] = vaa[IX(0, npy - 1)]; }
    if (W && S) { uaa[IX(-1, 0)] = n_ua[0]; uaa[IX(0, 0)] = n_ua[1]; }
    if (E && S) { uaa[IX(npx, 0)] = n_ua[2]; uaa[IX(npx + 1, 0)] = n_ua[3]; }
    if (E && N) { uaa[IX(npx, npy)] = n_ua[4]; uaa[IX(npx + 1, npy)] = n_ua[5]; }
    if (W && N) { uaa[IX(-1, npy)] = n_ua[6]; uaa[IX(0, npy)] = n_ua[7]; }
    if (W && S) { n_va[0] = -uaa[IX(2, 0)]; n_va[1] = -uaa[IX(1, 0)]; }
    if (E && S) { n_va[2] = uaa[IX(npx - 1, 0)]; n_va[3] = uaa[IX(npx - 2, 0)]; }
    if (E && N) { n_va[4] = -uaa[IX(npx - 1, npy)]; n_va[5] = -uaa[IX(npx - 2, npy)]; }
    if (W && N) { n_va[6] = uaa[IX(1, npy)]; n_va[7] = uaa[IX(2, npy)]; }
    if (W && S) { vaa[IX(0, -1)] = n_va[0]; vaa[IX(0, 0)] = n_va[1]; }
    if (E && S) { vaa[IX(npx, 0)] = n_va[2]; vaa[IX(npx, -1)] = n_va[3]; }
    if (E && N) { vaa[IX(npx, npy)] = n_va[4]; vaa[IX(npx, npy + 1)] = n_va[5]; }
    if (W && N) { vaa[IX(0, npy)] = n_va[6]; vaa[IX(0, npy + 1)] = n_va[7]; }
  });

  // (C) divergence on corners.  Two levels per thread: the ten metric terms of a corner are read once and
  // used for both levels (they are 2/3 of this kernel's bytes).
  if (nord > 0) {
    const int npair = (nz1 + KC) / KC;
    auto stage_c = [=] FV3_HD(int t, int kp, int i, int j) {
      const int fl = g.flags[t];
      if (march) {  // the corners whose 2 x 2 block of ua / va the marching kernel holds are done there
        const CswRect rc = b_rect(fl);
        if (i >= rc.i_lo && i >= 1 && i <= rc.i_hi && j >= rc.j_lo && j >= 1 && j <= rc.j_hi) return;
      }
      const long m2 = t * g.st2;
      const bool W = fl & FV3_W, E = fl & FV3_E, S = fl & FV3_S, N = fl & FV3_N;
      const int npx = g.npx, npy = g.npy;
      // metric factors of the four faces around the corner (exactly the sub-expressions of the one-level form)
      struct Face {
        Real cs, f;  // cos sum, dyc * 0.5 * sin sum  -- kept as separate factors: (.. * dyc) * 0.5 * (sin + sin)
        Real d, ss;
        bool edge;
      };
      auto UFm = [&](int ii, int jj) {
        const unsigned q = IX(ii, jj), qm = IX(ii, jj - 1);
        Face m;
        m.edge = (S && jj == 1) || (N && jj == npy);
        m.cs = (g.cos_sg4 + m2)[qm] + (g.cos_sg2 + m2)[q];
        m.d = (g.dyc + m2)[q];
        m.ss = (g.sin_sg4 + m2)[qm] + (g.sin_sg2 + m2)[q];
        m.f = (Real)0;
        return m;
      };
      auto VFm = [&](int ii, int jj) {
        const unsigned q = IX(ii, jj), qm = IX(ii - 1, jj);
        Face m;
        m.edge = (W && ii == 1) || (E && ii == npx);
        m.cs = (g.cos_sg3 + m2)[qm] + (g.cos_sg1 + m2)[q];
        m.d = (g.dxc + m2)[q];
        m.ss = (g.sin_sg3 + m2)[qm] + (g.sin_sg1 + m2)[q];
        m.f = (Real)0;
        return m;
      };
      const Face mu0 = UFm(i - 1, j), mu1 = UFm(i, j), mv0 = VFm(i, j - 1), mv1 = VFm(i, j);
      const bool cSW = W && S && i == 1 && j == 1, cSE = E && S && i == npx && j == 1, cNE = E && N && i == npx && j == npy, cNW = W && N && i == 1 && j == npy;
      Face mvc = mv0;
      if (cSW) mvc = VFm(1, 0);
      if (cSE) mvc = VFm(npx, 0);
      if (cNE) mvc = VFm(npx, npy);
      if (cNW) mvc = VFm(1, npy);
      const Real rac = (g.rarea_c + m2)[IX(i, j)];
#pragma unroll 1
      for (int kk = 0; kk < KC; ++kk) {
        const int k = KC * kp + kk;
        if (k > nz1) break;
        const long b = t * g.st + k * g.sk;
        auto UF = [&](int ii, int jj, const Face &m) -> Real {
          const unsigned q = IX(ii, jj), qm = IX(ii, jj - 1);
          if (m.edge) return (u + b)[q] * m.d * (Real)0.5 * m.ss;
          return ((u + b)[q] - (Real)0.25 * ((va + b)[qm] + (va + b)[q]) * m.cs) * m.d * (Real)0.5 * m.ss;
        };
        auto VF = [&](int ii, int jj, const Face &m) -> Real {
          const unsigned q = IX(ii, jj), qm = IX(ii - 1, jj);
          if (m.edge) return (v + b)[q] * m.d * (Real)0.5 * m.ss;
          return ((v + b)[q] - (Real)0.25 * ((ua + b)[qm] + (ua + b)[q]) * m.cs) * m.d * (Real)0.5 * m.ss;
        };
        Real dv = VF(i, j - 1, mv0) - VF(i, j, mv1) + UF(i - 1, j, mu0) - UF(i, j, mu1);
        if (cSW) dv -= VF(1, 0, mvc);
        if (cSE) dv -= VF(npx, 0, mvc);
        if (cNE) dv += VF(npx, npy, mvc);
        if (cNW) dv += VF(1, npy, mvc);
        (divgd + b)[IX(i, j)] = rac * dv;
      }
    };
    if (march) {  // the four windows along the sub-domain boundary
      const int e0 = g.nx - 3;
      launch_frame(c, s, Frame{{Box{1, 5, 1, g.ny + 1, 0, npair - 1}, Box{e0, e0 + 4, 1, g.ny + 1, 0, 0}, Box{6, g.nx - 4, 1, 5, 0, 0}, Box{6, g.nx - 4, g.ny - 3, g.ny + 1, 0, 0}}}, stage_c);
    } else {
      launch3(c, s, Box{1, g.nx + 1, 1, g.ny + 1, 0, npair - 1}, stage_c);
    }
  }

  // (D) upwind transport (delpc, ptc, wc), kinetic energy, absolute vorticity
  auto stage_d = [=] FV3_HD(int t, int kp, int i_, int j_) {
    const int fl = g.flags[t];
    // (fused form: the cells of RD and the corners of VR are done by the marching kernel -- see csw_fused_stream)
    bool do_cell = true, do_vort = true;
    if (fused) {
      const CswRect rc = b_rect(fl);
      do_cell = !(i_ >= rc.i_lo && i_ <= rc.i_hi - 1 && j_ >= rc.j_lo && j_ <= rc.j_hi - 1);
      do_vort = !(i_ >= rc.i_lo + 1 && i_ <= rc.i_hi && j_ >= rc.j_lo + 1 && j_ <= rc.j_hi);
      if (!do_cell && !do_vort) return;
    }
    const long m2 = t * g.st2;
    const bool W = fl & FV3_W, E = fl & FV3_E, S = fl & FV3_S, N = fl & FV3_N;
    const int npx = g.npx, npy = g.npy;
    int i = i_, j = j_;
    unsigned p = IX(i, j);
    // metric terms of the point, shared by the levels of the chunk (tile-edge clauses read their own)
    const bool corner_pt = i >= 1 && j >= 1 && do_vort;
    const Real ra = (g.rarea + m2)[p];
    const Real m_dxc_s = corner_pt ? (g.dxc + m2)[IX(i, j - 1)] : (Real)0, m_dxc = (g.dxc + m2)[p];
    const Real m_dyc_w = corner_pt ? (g.dyc + m2)[IX(i - 1, j)] : (Real)0, m_dyc = (g.dyc + m2)[p];
    const Real m_fc = (g.fC + m2)[p], m_rac = (g.rarea_c + m2)[p];
#pragma unroll 1
    for (int kk = 0; kk < KC; ++kk) {
    const int k = KC * kp + kk;
    if (k > nz1) break;
    i = i_;
    j = j_;
    FV3_LAUNDER(i);
    FV3_LAUNDER(j);
    p = IX(i, j);
    const long b = t * g.st + k * g.sk;
    if (do_cell) {
      // x fluxes at faces i and i+1 (fill_4corners x), y fluxes at j and j+1 (fill_4corners y)
      Real fx1[2], fx[2], fx2[2], fy1[2], fy[2], fy2[2];
      for (int a = 0; a < 2; ++a) {
        const int ii = i + a;
        const Real utv = (ut + b)[IX(ii, j)];
        const long src = utv > (Real)0 ? f4_index<1>(g, fl, ii - 1, j) : f4_index<1>(g, fl, ii, j);
        fx1[a] = utv * (delp + b)[src];
        fx[a] = fx1[a] * (pt + b)[src];
        fx2[a] = fx1[a] * (w + b)[src];
        const int jj = j + a;
        const Real vtv = (vt + b)[IX(i, jj)];
        const long srcy = vtv > (Real)0 ? f4_index<2>(g, fl, i, jj - 1) : f4_index<2>(g, fl, i, jj);
        fy1[a] = vtv * (delp + b)[srcy];
        fy[a] = fy1[a] * (pt + b)[srcy];
        fy2[a] = fy1[a] * (w + b)[srcy];
      }
      const Real dpc = (delp + b)[p] + (fx1[0] - fx1[1] + fy1[0] - fy1[1]) * ra;
      (delpc + b)[p] = dpc;
      (ptc + b)[p] = ((pt + b)[p] * (delp + b)[p] + (fx[0] - fx[1] + fy[0] - fy[1]) * ra) / dpc;
      (omga + b)[p] = ((w + b)[p] * (delp + b)[p] + (fx2[0] - fx2[1] + fy2[0] - fy2[1]) * ra) / dpc;
    }
    if (do_cell) {
      const Real uav = (ua + b)[p], vav = (va + b)[p];
      Real kev = uav > (Real)0 ? (uc + b)[p] : (uc + b)[IX(i + 1, j)];
      Real vov = vav > (Real)0 ? (vc + b)[p] : (vc + b)[IX(i, j + 1)];
      if ((W && i == 1) || (E && i == npx)) {
        if (uav > (Real)0) kev = (uc + b)[p] * (g.sin_sg1 + m2)[p] + (v + b)[p] * (g.cos_sg1 + m2)[p];
      }
      if ((W && i == 0) || (E && i == npx - 1)) {
        if (!(uav > (Real)0)) kev = (uc + b)[IX(i + 1, j)] * (g.sin_sg3 + m2)[p] + (v + b)[IX(i + 1, j)] * (g.cos_sg3 + m2)[p];
      }
      if ((S && j == 1) || (N && j == npy)) {
        if (vav > (Real)0) vov = (vc + b)[p] * (g.sin_sg2 + m2)[p] + (u + b)[p] * (g.cos_sg2 + m2)[p];
      }
      if ((S && j == 0) || (N && j == npy - 1)) {
        if (!(vav > (Real)0)) vov = (vc + b)[IX(i, j + 1)] * (g.sin_sg4 + m2)[p] + (u + b)[IX(i, j + 1)] * (g.cos_sg4 + m2)[p];
      }
      (ke + b)[p] = (Real)0.5 * dt2 * (uav * kev + vav * vov);
    }
    if (corner_pt) {  // absolute vorticity on corners is..ie+1, js..je+1
      auto FY = [&](int ii, int jj) { return (vc + b)[IX(ii, jj)] * (g.dyc + m2)[IX(ii, jj)]; };
      Real vo = (uc + b)[IX(i, j - 1)] * m_dxc_s - (uc + b)[p] * m_dxc - (vc + b)[IX(i - 1, j)] * m_dyc_w + (vc + b)[p] * m_dyc;
      if (W && S && i == 1 && j == 1) vo += FY(0, 1);
      if (E && S && i == npx && j == 1) vo -= FY(npx, 1);
      if (E && N && i == npx && j == npy) vo -= FY(npx, npy);
      if (W && N && i == 1 && j == npy) vo += FY(0, npy);
      (vort + b)[p] = m_fc + m_rac * vo;
    }
    }
  };
  // Round 6, FV3_CSW_DEFER=1 (measured, NOT the default: R6-11): inside the sequencer the eight window launches of stages D and E (1.15 ms of small launches behind
  // the march) go to the auxiliary stream and the operator returns without waiting for them: update_dz_c, which follows, reads ut / vt / zh only, and
  // riem_solver_c -- the first reader of the windows' delpc / ptc / wc -- is where the sequencer joins (fv3_csw_join).  Same values; c_sw shrinks by 1.1 ms and
  // update_dz_c, bandwidth-bound beside the windows, grows by 0.85: four same-box pairs gave -0.7 / -0.0 / +0.6 / +0.1 ms per sub-step.  Events 6 = fork, 7 = join.
  static const bool defer_on = getenv("FV3_CSW_DEFER") && getenv("FV3_CSW_DEFER")[0] == '1';
  fv3_stream_t sd_ = (fused && defer_on && c->seq_csw_defer) ? fv3_aux(c, s) : s;
  if (sd_ != s) {
    fv3_signal(c, s, 6);
    fv3_wait(c, sd_, 6);
  }
  if (fused) {  // the four windows along the sub-domain boundary
    const int e0 = g.nx - 4;
    launch_frame(c, sd_, Frame{{Box{0, 6, 0, g.ny + 1, 0, nkc - 1}, Box{e0, e0 + 5, 0, g.ny + 1, 0, 0}, Box{7, g.nx - 5, 0, 6, 0, 0}, Box{7, g.nx - 5, g.ny - 4, g.ny + 1, 0, 0}}}, stage_d);
  } else {
    launch3(c, s, Box{0, g.nx + 1, 0, g.ny + 1, 0, nkc - 1}, stage_d);
  }

  // (E) time-centred C-grid winds (two levels per thread: the six metric terms are read once)
  auto stage_e = [=] FV3_HD(int t, int kp, int i, int j) {
    const int fl = g.flags[t];
    if (fused) {  // the cells of RE are done by the marching kernel
      const CswRect rc = b_rect(fl);
      if (i >= rc.i_lo + 1 && i <= rc.i_hi - 1 && j >= rc.j_lo + 1 && j <= rc.j_hi - 1) return;
    }
    const long m2 = t * g.st2;
    const bool W = fl & FV3_W, E = fl & FV3_E, S = fl & FV3_S, N = fl & FV3_N;
    const int npx = g.npx, npy = g.npy;
    const unsigned p = IX(i, j);
    const bool edge_u = (W && i == 1) || (E && i == npx), edge_v = (S && j == 1) || (N && j == npy);
    const Real cau = (g.cosa_u + m2)[p], sau = (g.sina_u + m2)[p], rdxc = (g.rdxc + m2)[p];
    const Real cav = (g.cosa_v + m2)[p], sav = (g.sina_v + m2)[p], rdyc = (g.rdyc + m2)[p];
#pragma unroll 1
    for (int kk = 0; kk < KC; ++kk) {
      const int k = KC * kp + kk;
      if (k > nz1) break;
      const long b = t * g.st + k * g.sk;
      if (j <= g.ny) {
        const Real ucv = (uc + b)[p];
        const Real fy1 = edge_u ? dt2 * (v + b)[p] : dt2 * ((v + b)[p] - ucv * cau) / sau;
        const Real fyv = fy1 > (Real)0 ? (vort + b)[p] : (vort + b)[IX(i, j + 1)];
        (uc + b)[p] = ucv + fy1 * fyv + rdxc * ((ke + b)[IX(i - 1, j)] - (ke + b)[p]);
      }
      if (i <= g.nx) {
        const Real vcv = (vc + b)[p];
        const Real fx1 = edge_v ? dt2 * (u + b)[p] : dt2 * ((u + b)[p] - vcv * cav) / sav;
        const Real fxv = fx1 > (Real)0 ? (vort + b)[p] : (vort + b)[IX(i + 1, j)];
        (vc + b)[p] = vcv - fx1 * fxv + rdyc * ((ke + b)[IX(i, j - 1)] - (ke + b)[p]);
      }
    }
  };
  {
    const int nke = (nz1 + KC) / KC - 1;
    if (fused) {
      const int e0 = g.nx - 4;
      launch_frame(c, sd_, Frame{{Box{1, 6, 1, g.ny + 1, 0, nke}, Box{e0, e0 + 5, 1, g.ny + 1, 0, 0}, Box{7, g.nx - 5, 1, 6, 0, 0}, Box{7, g.nx - 5, g.ny - 4, g.ny + 1, 0, 0}}}, stage_e);
    } else {
      launch3(c, s, Box{1, g.nx + 1, 1, g.ny + 1, 0, nke}, stage_e);
    }
  }
  if (sd_ != s) {
    fv3_signal(c, sd_, 7);
    c->csw_pending = true;  // (the sequencer joins: fv3_csw_join)
  }
  return fv3_post(c, s, "c_sw");
}

// the join c_sw left to the sequencer (seq_csw_defer): the caller's stream waits for the stage D / E windows on the auxiliary stream
int fv3_csw_join(fv3_ctx *c, void *stream) {
  if (c && c->csw_pending) {
    fv3_wait(c, (fv3_stream_t)stream, 7);
    c->csw_pending = false;
  }
  return FV3_OK;
}
