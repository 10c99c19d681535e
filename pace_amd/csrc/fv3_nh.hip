// fv3_nh.hip -- non-hydrostatic column / pressure-gradient operators: update_dz_c, update_dz_d,
// Riem_Solver_c, Riem_Solver3 (SIM1), p_grad_c, nh_p_grad, pk3_halo, edge_pe, Ray_fast,
// del2_cubed, apply_diffusive_heating.  CPU twin: oracle/fv3_oracle/nh.py.  [SURVEY A.5-A.12]
//
// Column kernels: one thread per (i, j) column walks k; because i is the fastest index every
// per-level access of a wavefront is one coalesced row.  Per-level temporaries that do not fit
// registers (tridiagonal gam / pp / w) live in context scratch fields with the same layout.
#include "fv3_ops.h"
#include "fv3_math.h"
#include "fv3_agpr.h"

// column access: uniform (sub-domain, level) base + 32-bit in-plane offset
#define K_(arr, k) ((arr) + tb + (long)(k)*g.sk)[pix]

namespace {

// SIM1_solver (MOIST_CAPPA form).  On entry: DZ = layer thickness (negative), W2 = w, PM = layer
// mean pressure; delp / cappa / pt are the column inputs.  PP (nz+1), GAM, W2 are scratch.
// On exit: W2 = new w, DZ = new dz, PE (nz+1) = non-hydrostatic pressure perturbation.
struct Sim1 {
  Geo g;
  Real rgas, rgrav, p_fac, ptop;
  // setup(k, pm, dz): produces (and stores) the layer-mean pressure and thickness of level k; called once per
  //   level, in increasing k, from inside the first sweep (the caller's own prefix sums live in the functor);
  // finish(k, dz): consumes the new thickness of level k; called once per level, in decreasing k, from inside
  //   the last sweep;
  // wout (optional): receives the new w (written in the pressure-perturbation sweep, after the old w is read).
  // The sweeps are fused where their directions agree: 6 column walks instead of 8, no read-back of PM / DZ
  // in the first one and of DZ / W2 in the last one.
  template <class Setup, class Finish>
  FV3_HD void run(long tb, unsigned pix, Real dt, const Real *delp, const Real *cappa, const Real *pt, const Real *w1, Real ws, Real *PM, Real *DZ, Real *W2, Real *PP,
                  Real *GAM, Real *PE, Real *wout, Setup setup, Finish finish) const {
    const int nz = g.nz;
    const Real t1g = (Real)2.0 * dt * dt, rdt = (Real)1.0 / dt, r3 = (Real)(1.0 / 3.0);
    auto DM = [&](int k) { return K_(delp, k) * rgrav; };
    auto GM = [&](int k) { return (Real)1.0 / ((Real)1.0 - K_(cappa, k)); };
    // ---- sweep 1 (up): caller's setup + forward elimination for pp
    {
      Real pm_k, dz_k;
      setup(0, pm_k, dz_k);
      Real dm_m = (Real)0, dm_k = DM(0);
      Real pe_k = fv3_exp(GM(0) * fv3_log(-dm_k / dz_k * rgas * K_(pt, 0))) - pm_k;
      Real bet = (Real)0;
      K_(PP, 0) = (Real)0;
      for (int k = 0; k < nz; ++k) {
        Real bb, dd;
        Real pe_n = (Real)0, dm_n = (Real)0;
        if (k < nz - 1) {
          Real pm_n, dz_n;
          setup(k + 1, pm_n, dz_n);
          dm_n = DM(k + 1);
          const Real g_rat = dm_k / dm_n;
          pe_n = fv3_exp(GM(k + 1) * fv3_log(-dm_n / dz_n * rgas * K_(pt, k + 1))) - pm_n;
          bb = (Real)2.0 * ((Real)1.0 + g_rat);
          dd = (Real)3.0 * (pe_k + g_rat * pe_n);
        } else {
          bb = (Real)2.0;
          dd = (Real)3.0 * pe_k;
        }
        if (k == 0) {
          bet = bb;
          K_(PP, 1) = dd / bet;
        } else {
          const Real g_prev = dm_m / dm_k;
          const Real gam = g_prev / bet;
          K_(GAM, k) = gam;
          bet = bb - gam;
          K_(PP, k + 1) = (dd - K_(PP, k)) / bet;
        }
        pe_k = pe_n;
        dm_m = dm_k;
        dm_k = dm_n;
      }
      // ---- sweep 2 (down): back substitution
      for (int k = nz - 1; k >= 1; --k) K_(PP, k) = K_(PP, k) - K_(GAM, k) * K_(PP, k + 1);
    }
    // ---- sweep 3 (up): forward elimination for w
    {
      // pem[k] rolling prefix sum (same operation order as the setup pass)
      Real pem_k = ptop;  // pem[0]
      auto AA = [&](int k, Real pem_at_k) {
        return t1g * (Real)0.5 * (GM(k - 1) + GM(k)) / (K_(DZ, k - 1) + K_(DZ, k)) * (pem_at_k + K_(PP, k));
      };
      Real pem1 = pem_k + K_(delp, 0);  // pem[1]
      Real aa_k1 = AA(1, pem1);         // aa[1]
      Real bet = DM(0) - aa_k1;
      K_(W2, 0) = (DM(0) * K_(w1, 0) + dt * K_(PP, 1)) / bet;
      Real pem_cur = pem1;  // pem[k] for k = 1
      Real aa_k = aa_k1;
      for (int k = 1; k < nz - 1; ++k) {
        const Real pem_next = pem_cur + K_(delp, k);  // pem[k+1]
        const Real aa_n = AA(k + 1, pem_next);
        const Real gam = aa_k / bet;
        K_(GAM, k) = gam;
        bet = DM(k) - (aa_k + aa_n + aa_k * gam);
        K_(W2, k) = (DM(k) * K_(w1, k) + dt * (K_(PP, k + 1) - K_(PP, k)) - aa_k * K_(W2, k - 1)) / bet;
        aa_k = aa_n;
        pem_cur = pem_next;
      }
      // pem_cur = pem[nz-1]; bottom
      const Real pem_nz = pem_cur + K_(delp, nz - 1);
      const Real p1 = t1g * GM(nz - 1) / K_(DZ, nz - 1) * (pem_nz + K_(PP, nz));
      const Real gam = aa_k / bet;
      K_(GAM, nz - 1) = gam;
      bet = DM(nz - 1) - (aa_k + p1 + aa_k * gam);
      K_(W2, nz - 1) = (DM(nz - 1) * K_(w1, nz - 1) + dt * (K_(PP, nz) - K_(PP, nz - 1)) - p1 * ws - aa_k * K_(W2, nz - 2)) / bet;
      // ---- sweep 4 (down): back substitution
      for (int k = nz - 2; k >= 0; --k) K_(W2, k) = K_(W2, k) - K_(GAM, k + 1) * K_(W2, k + 1);
    }
    // ---- sweep 5 (up): new pressure perturbation (+ the new w leaves through wout)
    K_(PE, 0) = (Real)0;
    for (int k = 0; k < nz; ++k) {
      const Real w2k = K_(W2, k);
      K_(PE, k + 1) = K_(PE, k) + DM(k) * (w2k - K_(w1, k)) * rdt;
      if (wout) K_(wout, k) = w2k;
    }
    // ---- sweep 6 (down): new layer thickness, handed to the caller's finish
    Real p1 = (K_(PE, nz - 1) + (Real)2.0 * K_(PE, nz)) * r3;
    {
      const Real dzn = -DM(nz - 1) * rgas * K_(pt, nz - 1) * fv3_exp((K_(cappa, nz - 1) - (Real)1.0) * fv3_log(fv3_max(p_fac * K_(PM, nz - 1), p1 + K_(PM, nz - 1))));
      K_(DZ, nz - 1) = dzn;
      finish(nz - 1, dzn);
    }
    for (int k = nz - 2; k >= 0; --k) {
      const Real g_rat = DM(k) / DM(k + 1);
      const Real bb = (Real)2.0 * ((Real)1.0 + g_rat);
      p1 = (K_(PE, k) + bb * K_(PE, k + 1) + g_rat * K_(PE, k + 2)) * r3 - g_rat * p1;
      const Real dzn = -DM(k) * rgas * K_(pt, k) * fv3_exp((K_(cappa, k) - (Real)1.0) * fv3_log(fv3_max(p_fac * K_(PM, k), p1 + K_(PM, k))));
      K_(DZ, k) = dzn;
      finish(k, dzn);
    }
  }
};

// ---------------------------------------------------------------------------------------------
// SIM1 as a wave kernel.  The column solver above moves ~45 full-field passes through HBM per call
// (every tridiagonal temporary is a context scratch field); here a lane owns one column, the
// PP -> W2 -> PE chain of temporaries lives in ONE LDS line per lane (slot k of a column's line
// holds PP(k+1), then W2(k), then PE(k+1): each value dies exactly where its successor is born),
// the two gam arrays live in the lane's ACCUMULATION registers (RA form, fv3_agpr.h: a wave at one wave per SIMD owns 512
// registers per lane and the solver uses a third of them), only PM still goes through memory, and the old thickness is
// recomputed from the interface heights instead of being stored.  nz * 512 B of LDS per wave = 4 waves / CU at L79, so
// memory-level parallelism comes from explicit register prefetch: KWALK keeps U levels of every input in flight while the U
// levels loaded before are being computed.  At one wave per SIMD nothing hides a wave's own VALU time: fields are addressed as
// scalar base + 32-bit byte offset (KW_), divisions use the unscaled sequence (fv3_div), log / exp are range-specific (fv3_math.h).
// Operation order per value is the one of Sim1::run (bitwise the same results).
// ---------------------------------------------------------------------------------------------
template <int N>
struct KRec {
  Real v[N];
};

template <int N, int U, bool UP, class Load, class Body>
FV3_HD inline void k_walk(int nz, Load load, Body body) {
  KRec<N> buf[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int kk = u < nz ? u : nz - 1;
    buf[u] = load(UP ? kk : nz - 1 - kk);
  }
  for (int c = 0; c < nz; c += U) {
    KRec<N> nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int kk = c + U + u;
      kk = kk < nz ? kk : nz - 1;
      nxt[u] = load(UP ? kk : nz - 1 - kk);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (c + u < nz) body(UP ? c + u : nz - 1 - (c + u), buf[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) buf[u] = nxt[u];
  }
}

// k_walk as a statement macro: the body (the trailing block, which sees the level K and the record r) is expanded INSIDE the function
// that owns the register arrays -- a lambda that captures them by reference sends them to scratch memory (the element store turns into a
// scalar store through a pointer before the closure is dissolved, and nothing promotes the array afterwards).  `load` may be a lambda:
// it touches no register array.
#define KWALK(N, U, UP, load, ...) KWALK_E(N, U, UP, load, , __VA_ARGS__)
// ... with a statement EPI that runs after the U bodies of a loop iteration (sees c_: the first level index of the group)
#define KWALK_E(N, U, UP, load, EPI, ...)                                    \
  {                                                                          \
    KRec<N> buf_[U];                                                         \
    _Pragma("unroll") for (int u_ = 0; u_ < U; ++u_) {                       \
      const int kk_ = u_ < nz ? u_ : nz - 1;                                 \
      buf_[u_] = load(UP ? kk_ : nz - 1 - kk_);                              \
    }                                                                        \
    for (int c_ = 0; c_ < nz; c_ += U) {                                     \
      KRec<N> nxt_[U];                                                       \
      _Pragma("unroll") for (int u_ = 0; u_ < U; ++u_) {                     \
        int kk_ = c_ + U + u_;                                               \
        kk_ = kk_ < nz ? kk_ : nz - 1;                                       \
        nxt_[u_] = load(UP ? kk_ : nz - 1 - kk_);                            \
      }                                                                      \
      _Pragma("unroll") for (int u_ = 0; u_ < U; ++u_) if (c_ + u_ < nz) {   \
        const int K = UP ? c_ + u_ : nz - 1 - (c_ + u_);                     \
        const KRec<N> &r = buf_[u_];                                         \
        __VA_ARGS__                                                          \
      }                                                                      \
      EPI;                                                                   \
      _Pragma("unroll") for (int u_ = 0; u_ < U; ++u_) buf_[u_] = nxt_[u_];  \
    }                                                                        \
  }

// Scalar-base addressing: a wave-uniform 64-bit base (the field of the sub-domain) + a 32-bit BYTE offset per lane that carries the level too.
// With the plain `(arr + tb + k * sk)[pix]` the compiler forms a 64-bit address per access with a vector instruction (it cannot prove that
// pix * 8 fits 32 bits, and `global_load v, v_off, s[base]` is only selected when the zero-extension of the offset is in the same basic
// block as the access: a loop-invariant offset never is); here the offset changes with the level, so one 32-bit add per level serves every
// field of that level.  riem_wave_ok() checks that a sub-domain's field stays below 4 GB.  -DFV3_RIEM_ADDR64: the plain form (A/B).
#ifdef FV3_RIEM_ADDR64
#define KW_(arr, k) ((arr) + tb + (long)(k)*sk)[pix]
#else
#define KW_(arr, k) (*fv3_at((arr) + tb, pix * (unsigned)sizeof(Real) + (unsigned)(k) * (unsigned)(sk * (long)sizeof(Real))))
#endif
#ifndef FV3_RIEM_U
#define FV3_RIEM_U 4   // levels in flight in the sweeps with 4-5 inputs
#endif
#ifndef FV3_RIEM_U1
#define FV3_RIEM_U1 8  // ... in the sweeps with 1-2 inputs
#endif

// RA form of Sim1W::run: the tridiagonal `gam` of a column lives in the lane's accumulation registers (fv3_agpr.h) instead of going
// through the scratch field GAM -- four field passes less per call.  80 levels at most, fp64 build; deeper columns keep the scratch field.
#define FV3_KREG_LEVELS (sizeof(Real) == 8 ? FV3_AGPR_LEVELS : FV3_AGPR_LEVELS_F32)
#if defined(__HIP_DEVICE_COMPILE__)
// (level k of the column sits in slot k + 1: the forward sweeps produce gam(m - 1) while they walk level m, so the four values of a
//  U = 4 loop iteration are the slots c_ .. c_ + 3 = ONE aligned group, stored by one access site per sweep: KREG_SET4)
#define KREG_DECL(n)
#define KREG_SET(n, k, v) fv3_agpr_set((k) + 1, (Real)(v))
#define KREG_GET(n, k) fv3_kreg_get((Real)0, (k) + 1)
#define KREG_SET4(n, c, q) fv3_agpr_set4((c) >> 2, (q)[0], (q)[1], (q)[2], (q)[3])
__device__ __attribute__((always_inline)) inline double fv3_kreg_get(double, int k) { return fv3_agpr_get(k); }
__device__ __attribute__((always_inline)) inline float fv3_kreg_get(float, int k) { return fv3_agpr_get_f32(k); }
#else
#define KREG_DECL(n) Real n[FV3_AGPR_LEVELS_F32 + 4]
#define KREG_SET(n, k, v) n[(k) + 1] = (v)
#define KREG_GET(n, k) n[(k) + 1]
#define KREG_SET4(n, c, q) n[c] = (q)[0], n[(c) + 1] = (q)[1], n[(c) + 2] = (q)[2], n[(c) + 3] = (q)[3]
#endif

struct Sim1W {
  long sk;
  int nz;
  Real rgas, rgrav, p_fac, ptop;
  // cl.pm(k, delp_k, qcon_k): layer-mean pressure of level k (called once per level, increasing k);
  // cl.out_pe(k1, pe, delp_k): new pressure perturbation at interface k1 = k + 1 (increasing k);
  // cl.finish(k, dz): new thickness of level k (decreasing k).
  // A = this lane's LDS line (stride FV3_WAVE); zint = interface heights (nz + 1 levels) the old
  // thickness comes from; wout (optional, may alias w1) receives the new w.
  // GL: the gam arrays live in a second LDS line B (2 waves / CU at L79) instead of the scratch field GAM.
  // RA: gam lives in the lane's accumulation registers (KREG_*; nz <= FV3_KREG_LEVELS): four field passes less per call.
  // ZL: sweep 1 takes the interface heights from the LDS line A instead of zint -- z(k + 1) in slot k, z(0) = z0 (riem_solver3's pre-sweep put them
  //     there; slot m is read at step m and overwritten with PP at step m + 1); sweep 3 reads zint as always.
  template <bool GL, bool RA, bool ZL = false, class C>
  FV3_HD void run(Real *A, Real *B, long tb, unsigned pix, Real dt, const Real *delp, const Real *cappa, const Real *pt, const Real *qcon, const Real *zint, const Real *w1,
                  Real ws, Real *PM, Real *GAM, Real *wout, C &cl, Real z0 = (Real)0) const {
    const Real t1g = (Real)2.0 * dt * dt, rdt = (Real)1.0 / dt, r3 = (Real)(1.0 / 3.0);
    constexpr int U = FV3_RIEM_U, U1 = FV3_RIEM_U1;
    constexpr int UB = RA ? 1 : U1;  // the back substitutions: nothing to prefetch when gam is in registers (one access site)
    Real pp_nz;
    KREG_DECL(rg);  // gam
#ifdef FV3_RIEM_NO_G4  // (A/B: one access site per level of the unrolled body)
    constexpr bool G4 = false;
#else
    constexpr bool G4 = RA && U == 4;  // the forward sweeps store the four gam of a loop iteration through one access site
#endif
    Real gq[4] = {(Real)0, (Real)0, (Real)0, (Real)0};
    (void)gq;
#define GAM_PUT(k, v)                       \
  do {                                      \
    if constexpr (RA) KREG_SET(rg, k, v);   \
    else if (GL) B[(k)*FV3_WAVE] = (v);     \
    else KW_(GAM, k) = (v);                 \
  } while (0)
// (inside a KWALK body: u_ is the body's static position in its group)
#define GAM_PUT_U(k, v)                     \
  do {                                      \
    if constexpr (G4) gq[u_ & 3] = (v);     \
    else GAM_PUT(k, v);                     \
  } while (0)
#define GAM_FLUSH4()                        \
  do {                                      \
    if constexpr (G4) {                     \
      KREG_SET4(rg, c_, gq);                \
      gq[0] = gq[1] = gq[2] = gq[3] = (Real)0; \
    }                                       \
  } while (0)
    // ---- sweep 1 (up): layer pressures, forward elimination for pp.  PP(k+1) -> slot k
    {
      Real g_prev = (Real)0, dm_k = (Real)0, pe_k = (Real)0, bet = (Real)0, pp_k = (Real)0;  // g_prev = dm(k-1) / dm(k) = the previous step's g_rat
      Real z_top = ZL ? z0 : KW_(zint, 0);
      auto ld = [&](int k) {
        KRec<5> q;
        q.v[0] = KW_(delp, k);
        q.v[1] = KW_(cappa, k);
        q.v[2] = KW_(pt, k);
        q.v[3] = KW_(qcon, k);
        q.v[4] = ZL ? (Real)0 : KW_(zint, k + 1);
        return q;
      };
      KWALK_E(5, U, true, ld, GAM_FLUSH4(), {
        const int m = K;
        const Real pm_n = cl.pm(m, r.v[0], r.v[3]);
        KW_(PM, m) = pm_n;
        const Real z_lo = ZL ? A[m * FV3_WAVE] : r.v[4];  // (ZL: before this step's PP goes to slot m - 1)
        const Real dz_n = z_lo - z_top;
        z_top = z_lo;
        const Real dm_n = r.v[0] * rgrav;
        const Real pe_n = fv3_exp(fv3_div((Real)1.0, (Real)1.0 - r.v[1]) * fv3_log(fv3_div(-dm_n, dz_n) * rgas * r.v[2])) - pm_n;
        if (m >= 1) {
          const int k = m - 1;
          const Real g_rat = fv3_div(dm_k, dm_n);
          const Real bb = (Real)2.0 * ((Real)1.0 + g_rat);
          const Real dd = (Real)3.0 * (pe_k + g_rat * pe_n);
          if (k == 0) {
            bet = bb;
            pp_k = fv3_div(dd, bet);
          } else {
            const Real gam = fv3_div(g_prev, bet);
            GAM_PUT_U(k, gam);
            bet = bb - gam;
            pp_k = fv3_div(dd - pp_k, bet);
          }
          A[k * FV3_WAVE] = pp_k;
          g_prev = g_rat;
        }
        dm_k = dm_n;
        pe_k = pe_n;
      })
      {
        const int k = nz - 1;
        const Real bb = (Real)2.0, dd = (Real)3.0 * pe_k;
        const Real gam = fv3_div(g_prev, bet);
        GAM_PUT(k, gam);
        bet = bb - gam;
        pp_k = fv3_div(dd - pp_k, bet);
        A[k * FV3_WAVE] = pp_k;
      }
      pp_nz = pp_k;
    }
    auto ld_gam = [&](int k) {
      KRec<1> q;
      q.v[0] = GL || RA ? (Real)0 : KW_(GAM, k > 0 ? k : 1);
      return q;
    };
    // ---- sweep 2 (down): back substitution, PP(k) = PP(k) - gam(k) PP(k+1), k = nz-1 .. 1
    {
      Real pp_next = pp_nz;
      KWALK(1, UB, false, ld_gam, {
        if (K >= 1) {
          Real gk;
          if constexpr (RA) gk = KREG_GET(rg, K); else gk = GL ? B[K * FV3_WAVE] : r.v[0];
          const Real ppv = A[(K - 1) * FV3_WAVE] - gk * pp_next;
          A[(K - 1) * FV3_WAVE] = ppv;
          pp_next = ppv;
        }
      })
    }
    // ---- sweep 3 (up): forward elimination for w.  W2(k) -> slot k (PP(k+1) has been taken out one level earlier)
    {
      Real pem = ptop, gm_p = (Real)0, dz_p = (Real)0, aa_k = (Real)0, dmp = (Real)0, w1p = (Real)0, pp_lo = (Real)0, pp_lo2 = (Real)0, bet = (Real)0,
           w2_prev = (Real)0;
      Real z_top = KW_(zint, 0);
      auto ld = [&](int k) {
        KRec<4> q;
        q.v[0] = KW_(delp, k);
        q.v[1] = KW_(cappa, k);
        q.v[2] = KW_(zint, k + 1);
        q.v[3] = KW_(w1, k);
        return q;
      };
      KWALK_E(4, U, true, ld, GAM_FLUSH4(), {
        const int m = K;
        const Real gm_n = fv3_div((Real)1.0, (Real)1.0 - r.v[1]);
        const Real dz_n = r.v[2] - z_top;
        z_top = r.v[2];
        const Real pp_hi = A[m * FV3_WAVE];  // PP(m+1)
        if (m >= 1) {
          const Real aa_n = fv3_div(t1g * (Real)0.5 * (gm_p + gm_n), dz_p + dz_n) * (pem + pp_lo);
          const int k = m - 1;
          if (k == 0) {
            bet = dmp - aa_n;
            w2_prev = fv3_div(dmp * w1p + dt * pp_lo, bet);
          } else {
            const Real gam = fv3_div(aa_k, bet);
            GAM_PUT_U(k, gam);
            bet = dmp - (aa_k + aa_n + aa_k * gam);
            w2_prev = fv3_div(dmp * w1p + dt * (pp_lo - pp_lo2) - aa_k * w2_prev, bet);
          }
          A[k * FV3_WAVE] = w2_prev;
          aa_k = aa_n;
        }
        pem = pem + r.v[0];
        gm_p = gm_n;
        dz_p = dz_n;
        dmp = r.v[0] * rgrav;
        w1p = r.v[3];
        pp_lo2 = pp_lo;
        pp_lo = pp_hi;
      })
      {
        const Real p1 = fv3_div(t1g * gm_p, dz_p) * (pem + pp_lo);
        const Real gam = fv3_div(aa_k, bet);
        GAM_PUT(nz - 1, gam);
        bet = dmp - (aa_k + p1 + aa_k * gam);
        w2_prev = fv3_div(dmp * w1p + dt * (pp_lo - pp_lo2) - p1 * ws - aa_k * w2_prev, bet);
        A[(nz - 1) * FV3_WAVE] = w2_prev;
      }
      // ---- sweep 4 (down): back substitution, W2(k) = W2(k) - gam(k+1) W2(k+1), k = nz-2 .. 0
      Real w2_next = w2_prev;
      KWALK(1, UB, false, ld_gam, {
        if (K >= 1) {
          Real gk;
          if constexpr (RA) gk = KREG_GET(rg, K); else gk = GL ? B[K * FV3_WAVE] : r.v[0];
          const Real wv = A[(K - 1) * FV3_WAVE] - gk * w2_next;
          A[(K - 1) * FV3_WAVE] = wv;
          w2_next = wv;
        }
      })
    }
    // ---- sweep 5 (up): new pressure perturbation.  PE(k+1) -> slot k; the new w leaves through wout
    {
      Real pe_run = (Real)0;
      auto ld = [&](int k) {
        KRec<2> q;
        q.v[0] = KW_(delp, k);
        q.v[1] = KW_(w1, k);
        return q;
      };
      KWALK(2, U1, true, ld, {
        const Real w2k = A[K * FV3_WAVE];
        pe_run = pe_run + r.v[0] * rgrav * (w2k - r.v[1]) * rdt;
        A[K * FV3_WAVE] = pe_run;
        if (wout) FV3_ST_NT(KW_(wout, K), w2k);
        cl.out_pe(K + 1, pe_run, r.v[0]);
      })
    }
    // ---- sweep 6 (down): new layer thickness, handed to the caller's finish
    {
      Real p1 = (Real)0, dm_below = (Real)0, pe1 = A[(nz - 1) * FV3_WAVE], pe2 = (Real)0;
      auto ld = [&](int k) {
        KRec<4> q;
        q.v[0] = KW_(delp, k);
        q.v[1] = KW_(pt, k);
        q.v[2] = KW_(cappa, k);
        q.v[3] = KW_(PM, k);
        return q;
      };
      KWALK(4, U, false, ld, {
        const Real dm = r.v[0] * rgrav;
        const Real pe_k = K >= 1 ? A[(K - 1) * FV3_WAVE] : (Real)0;
        if (K == nz - 1) {
          p1 = (pe_k + (Real)2.0 * pe1) * r3;
        } else {
          const Real g_rat = fv3_div(dm, dm_below);
          const Real bb = (Real)2.0 * ((Real)1.0 + g_rat);
          p1 = (pe_k + bb * pe1 + g_rat * pe2) * r3 - g_rat * p1;
        }
        const Real pmk = r.v[3];
        const Real dzn = -dm * rgas * r.v[1] * fv3_exp((r.v[2] - (Real)1.0) * fv3_log(fv3_max(p_fac * pmk, p1 + pmk)));
        cl.finish(K, dzn);
        pe2 = pe1;
        pe1 = pe_k;
        dm_below = dm;
      })
    }
#undef GAM_PUT
#undef GAM_PUT_U
#undef GAM_FLUSH4
  }
};

#ifndef FV3_RIEM_GL
#define FV3_RIEM_GL 0  // 1: gam arrays in a second LDS line (experiment; halves the resident waves)
#endif
// LDS line budget of the wave solver; above it (very deep columns) the callers fall back to the column kernels
inline bool riem_wave_ok(const Geo &g, bool heavy = false) {
  static const char *e = getenv("FV3_RIEM_MODE");
  if (e && !strcmp(e, "columns")) return false;
  const size_t line = (size_t)g.nz * FV3_WAVE * sizeof(Real) * (FV3_RIEM_GL ? 2 : 1);
  // heavy = riem_solver3 (7 exp/log per level): below 4 waves per CU (line > 40 KB: 127 levels in fp64) the
  // bandwidth-bound column form is faster (25.7 vs 29.1 ms at C768 L127 fp64); riem_solver_c still gains (24.7 vs 29.2)
  if (heavy && !(e && !strcmp(e, "wave")) && line > 40 * 1024) return false;
  if ((g.nz + 2) * g.sk * (long)sizeof(Real) >= (1L << 32)) return false;  // (KW_: 32-bit byte offsets inside a sub-domain's field)
  return line <= (FV3_RIEM_GL ? 160 : 64) * 1024 && g.nz >= 3;
}
inline bool riem_gam_lds(const Geo &) { return FV3_RIEM_GL != 0; }
// gam in the accumulation registers (fv3_agpr.h; 80 levels in fp64, 128 in fp32); FV3_RIEM_REGS=0: through the scratch field (A/B, same values).
// (A first attempt let the COMPILER index a register-tuple array: riem_solver_c 9.64 -> 13.84 ms -- DESIGN §7.)
inline bool riem_reg_arrays(const Geo &g) {
  static const bool off = getenv("FV3_RIEM_REGS") && getenv("FV3_RIEM_REGS")[0] == '0';
  return !off && !FV3_RIEM_GL && g.nz < (int)FV3_KREG_LEVELS;  // (level k sits in slot k + 1)
}

#ifndef PG_KC
#define PG_KC 16  // levels one thread of the pressure-gradient kernels walks
#endif

// interface interpolation weights of update_dz_c
struct DzcW {
  Real top_ratio, bot_ratio;
};

}  // namespace

// ---------------------------------------------------------------------------------------------
extern "C" int fv3_update_dz_c(fv3_ctx *c, const fv3_field *zs_, const fv3_field *ut_, const fv3_field *vt_, const fv3_field *gz_, const fv3_field *ws_,
                               double dtd, void *stream) {
  return fv3_update_dz_c_from(c, zs_, ut_, vt_, gz_, gz_, ws_, dtd, stream);
}

// gz_in -> gz: the sequencer passes zh as gz_in on the sub-steps where the reference first copies zh into gz
// (dyn_core: "gz = zh" before update_dz_c), which saves that full-field copy; only cells 0..n+1 of gz are written,
// and nothing downstream (riem_solver_c, p_grad_c) reads gz beyond them.
int fv3_update_dz_c_from(fv3_ctx *c, const fv3_field *zs_, const fv3_field *ut_, const fv3_field *vt_, const fv3_field *gzin_, const fv3_field *gz_,
                         const fv3_field *ws_, double dtd, void *stream) {
  if (!c) return FV3_ERR_ARG;
  FV3_FIELD2D(zs, zs_) FV3_FIELD(ut, ut_) FV3_FIELD(vt, vt_) FV3_FIELD(gz, gz_) FV3_FIELD(gzin, gzin_) FV3_FIELD2D(ws, ws_)
  const Geo g = c->g;
  fv3_stream_t s = (fv3_stream_t)stream;
  const Real dt = (Real)dtd;
  const int nz = g.nz;
  const Real dz_min = (Real)c->cst.dz_min;
  const std::vector<double> &dp = c->dp_ref_h;
  const Real top_ratio = (Real)(dp[0] / (dp[0] + dp[1]));
  const Real bot_ratio = (Real)(dp[nz - 1] / (dp[nz - 2] + dp[nz - 1]));
  Real *gzn = c->scratch[SC_A];
  if (gzin != gz && nz >= 3) {
    // Out of place (the sequencer's zh -> gz form): ONE column kernel walks the interfaces upward from the surface,
    // forms the advected height of interface k in registers and applies the monotonicity scan at once -- the
    // intermediate field is never stored (2 field passes less) and each wind layer is read once (it serves the
    // interface above and the one below).  In place (gz -> gz) the two-kernel form below is required: neighbours'
    // heights are read while columns are rewritten.
    launch2(c, s, Box{0, g.nx + 1, 0, g.ny + 1, 0, 0}, [=] FV3_HD(int t, int i, int j) {
      const int fl = g.flags[t];
      const long tb = t * g.st, m2 = t * g.st2;
      const unsigned pix = IX(i, j), qx1 = IX(i + 1, j), qy1 = IX(i, j + 1);
      const unsigned gw = f4_index<1>(g, fl, i - 1, j), gc1 = f4_index<1>(g, fl, i, j), ge = f4_index<1>(g, fl, i + 1, j);
      const unsigned gs = f4_index<2>(g, fl, i, j - 1), gc2 = f4_index<2>(g, fl, i, j), gn = f4_index<2>(g, fl, i, j + 1);
      const Real ar = (g.area + m2)[pix];
      // layer values at the four faces: a* = layer k-1 (above the interface), c* = layer k (below), p* = layer k+1
      Real ax0, ax1, ay0, ay1, cx0 = (Real)0, cx1 = (Real)0, cy0 = (Real)0, cy1 = (Real)0, px0 = (Real)0, px1 = (Real)0, py0 = (Real)0, py1 = (Real)0;
      {
        const Real *u1 = ut + tb + (long)(nz - 1) * g.sk, *v1 = vt + tb + (long)(nz - 1) * g.sk;
        ax0 = u1[pix];
        ax1 = u1[qx1];
        ay0 = v1[pix];
        ay1 = v1[qy1];
      }
      Real below = (Real)0;
#pragma unroll 1
      for (int k = nz; k >= 0; --k) {
        // layer k-2 (becomes "above" for the next interface); at the bottom it also enters the extrapolation
        Real nx0 = (Real)0, nx1 = (Real)0, ny0 = (Real)0, ny1 = (Real)0;
        if (k >= 2) {
          const Real *u2 = ut + tb + (long)(k - 2) * g.sk, *v2 = vt + tb + (long)(k - 2) * g.sk;
          nx0 = u2[pix];
          nx1 = u2[qx1];
          ny0 = v2[pix];
          ny1 = v2[qy1];
        }
        Real x0, x1, y0, y1;
        if (k == nz) {
          x0 = ax0 + (ax0 - nx0) * bot_ratio;
          x1 = ax1 + (ax1 - nx1) * bot_ratio;
          y0 = ay0 + (ay0 - ny0) * bot_ratio;
          y1 = ay1 + (ay1 - ny1) * bot_ratio;
        } else if (k == 0) {
          x0 = cx0 + (cx0 - px0) * top_ratio;
          x1 = cx1 + (cx1 - px1) * top_ratio;
          y0 = cy0 + (cy0 - py0) * top_ratio;
          y1 = cy1 + (cy1 - py1) * top_ratio;
        } else {
          const Real d1 = g.dp_ref[k], d0 = g.dp_ref[k - 1];
          const Real int_ratio = (Real)1.0 / (d0 + d1);
          x0 = (d1 * ax0 + d0 * cx0) * int_ratio;
          x1 = (d1 * ax1 + d0 * cx1) * int_ratio;
          y0 = (d1 * ay0 + d0 * cy0) * int_ratio;
          y1 = (d1 * ay1 + d0 * cy1) * int_ratio;
        }
        const Real *gg = gzin + tb + (long)k * g.sk;
        const Real fx0 = x0 * (x0 > (Real)0 ? gg[gw] : gg[gc1]);
        const Real fx1 = x1 * (x1 > (Real)0 ? gg[gc1] : gg[ge]);
        const Real fy0 = y0 * (y0 > (Real)0 ? gg[gs] : gg[gc2]);
        const Real fy1 = y1 * (y1 > (Real)0 ? gg[gc2] : gg[gn]);
        const Real zn = (gg[pix] * ar + fx0 - fx1 + fy0 - fy1) / (ar + x0 - x1 + y0 - y1);
        Real v = zn;
        if (k == nz) {
          ws[t * g.st2 + pix] = (zs[t * g.st2 + pix] - zn) / dt;
        } else {
          v = fv3_max(zn, below + dz_min);
        }
        (gz + tb + (long)k * g.sk)[pix] = v;
        below = v;
        // shift: the interface above sees this one's "above" layer as its "below" layer
        px0 = cx0;
        px1 = cx1;
        py0 = cy0;
        py1 = cy1;
        cx0 = ax0;
        cx1 = ax1;
        cy0 = ay0;
        cy1 = ay1;
        if (k == nz) {
          // (interface nz-1 keeps layer nz-1 below it and gets layer nz-2 above it)
        }
        ax0 = nx0;
        ax1 = nx1;
        ay0 = ny0;
        ay1 = ny1;
      }
    });
    return fv3_post(c, s, "update_dz_c");
  }
  launch3(c, s, Box{0, g.nx + 1, 0, g.ny + 1, 0, nz}, [=] FV3_HD(int t, int k, int i, int j) {
    const int fl = g.flags[t];
    const long bt = t * g.st, b = bt + k * g.sk, m2 = t * g.st2;
    auto XI = [&](const Real *f0, int ii, int jj) -> Real {
      const unsigned q = IX(ii, jj);
      const Real *f = f0 + bt;
      if (k == 0) return f[q] + (f[q] - (f + g.sk)[q]) * top_ratio;
      if (k == nz) return (f + (nz - 1) * g.sk)[q] + ((f + (nz - 1) * g.sk)[q] - (f + (nz - 2) * g.sk)[q]) * bot_ratio;
      const Real int_ratio = (Real)1.0 / (g.dp_ref[k - 1] + g.dp_ref[k]);
      return (g.dp_ref[k] * (f + (k - 1) * g.sk)[q] + g.dp_ref[k - 1] * (f + k * g.sk)[q]) * int_ratio;
    };
    const Real *gg = gzin + b;
    const Real x0 = XI(ut, i, j), x1 = XI(ut, i + 1, j), y0 = XI(vt, i, j), y1 = XI(vt, i, j + 1);
    const Real fx0 = x0 * (x0 > (Real)0 ? gg[f4_index<1>(g, fl, i - 1, j)] : gg[f4_index<1>(g, fl, i, j)]);
    const Real fx1 = x1 * (x1 > (Real)0 ? gg[f4_index<1>(g, fl, i, j)] : gg[f4_index<1>(g, fl, i + 1, j)]);
    const Real fy0 = y0 * (y0 > (Real)0 ? gg[f4_index<2>(g, fl, i, j - 1)] : gg[f4_index<2>(g, fl, i, j)]);
    const Real fy1 = y1 * (y1 > (Real)0 ? gg[f4_index<2>(g, fl, i, j)] : gg[f4_index<2>(g, fl, i, j + 1)]);
    const unsigned p = IX(i, j);
    const Real ar = (g.area + m2)[p];
    (gzn + b)[p] = (gg[p] * ar + fx0 - fx1 + fy0 - fy1) / (ar + x0 - x1 + y0 - y1);
  });
  launch2(c, s, Box{0, g.nx + 1, 0, g.ny + 1, 0, 0}, [=] FV3_HD(int t, int i, int j) {
    const long tb = t * g.st;
    const unsigned pix = IX(i, j);
    const long p = tb + pix;
    (void)p;
    Real below = gzn[p + (long)nz * g.sk];
    gz[p + (long)nz * g.sk] = below;
    ws[t * g.st2 + IX(i, j)] = (zs[t * g.st2 + IX(i, j)] - below) / dt;
    for (int k = nz - 1; k >= 0; --k) {
      const Real v = fv3_max(K_(gzn, k), below + dz_min);
      K_(gz, k) = v;
      below = v;
    }
  });
  return fv3_post(c, s, "update_dz_c");
}

// ---------------------------------------------------------------------------------------------
extern "C" int fv3_riem_solver_c(fv3_ctx *c, double dt2d, const fv3_field *cappa_, double ptopd, const fv3_field *phis_, const fv3_field *ws_,
                                 const fv3_field *ptc_, const fv3_field *q_con_, const fv3_field *delpc_, const fv3_field *gz_, const fv3_field *pef_,
                                 const fv3_field *w3_, void *stream) {
  if (!c) return FV3_ERR_ARG;
  FV3_FIELD(cappa, cappa_) FV3_FIELD2D(phis, phis_) FV3_FIELD2D(ws, ws_) FV3_FIELD(ptc, ptc_) FV3_FIELD(q_con, q_con_) FV3_FIELD(delpc, delpc_)
  FV3_FIELD(gz, gz_) FV3_FIELD(pef, pef_) FV3_FIELD(w3, w3_)
  const Geo g = c->g;
  fv3_stream_t s = (fv3_stream_t)stream;
  const Real dt2 = (Real)dt2d, ptop = (Real)ptopd, grav = (Real)c->cst.grav;
  Sim1 sim{g, (Real)c->cst.rdgas, (Real)1.0 / (Real)c->cst.grav, (Real)c->cfg.p_fac, ptop};
  Real *PM = c->scratch[SC_A], *DZ = c->scratch[SC_B], *W2 = c->scratch[SC_C], *PP = c->scratch[SC_D], *GAM = c->scratch[SC_E];
  const int nz = g.nz;
  if (riem_wave_ok(g)) {
    const Sim1W sw{g.sk, nz, sim.rgas, sim.rgrav, sim.p_fac, ptop};
    const int i0 = 0, j0 = 0, ni = g.nx + 2, ncol = ni * (g.ny + 2);
    const long st = g.st, st2 = g.st2, sk = g.sk;
    const int sj32 = g.sj32, go = g.o;
    const bool gl = riem_gam_lds(g);
    auto go_ = [&](auto ra_tag) {
    constexpr bool RA = decltype(ra_tag)::value;
    launch_waves<1>(c, s, (ncol + FV3_WAVE - 1) / FV3_WAVE, 1, g.nsub, sizeof(Real) * nz * FV3_WAVE * (gl ? 2 : 1), [=] FV3_HD(const Blk &blk, char *smem_) {
      const int t = blk.bz;
      const long tb = t * st;
      FV3_LANES(blk, lane, l) {
        const int cidx = blk.bx * FV3_WAVE + lane;
        if (cidx >= ncol) continue;
        const int jr = cidx / ni;
        const unsigned pix = (unsigned)((j0 + jr + go) * sj32 + (i0 + cidx - jr * ni) + go);
        struct Cl {
          long tb, sk;
          unsigned pix;
          Real peg, pem, z, grav;
          Real *pef, *gz;
          FV3_HD Real pm(int, Real dm, Real qc) {
            const Real peg_n = peg + dm * ((Real)1.0 - qc);
            const Real v = fv3_div(peg_n - peg, fv3_log(fv3_div(peg_n, peg)));
            peg = peg_n;
            return v;
          }
          FV3_HD void out_pe(int k1, Real pe, Real dm) {
            pem = pem + dm;
            FV3_ST_NT(KW_(pef, k1), pe + pem);
          }
          FV3_HD void finish(int k, Real dz) {
            z = z - dz * grav;
            FV3_ST_NT(KW_(gz, k), z);
          }
        };
        const Real z_bot = phis[t * st2 + pix];
        Cl cl{tb, sk, pix, ptop, ptop, z_bot, grav, pef, gz};
        KW_(pef, 0) = ptop;
        sw.run<FV3_RIEM_GL != 0, RA>((Real *)smem_ + lane, (Real *)smem_ + nz * FV3_WAVE + lane, tb, pix, dt2, delpc, cappa, ptc, q_con, gz, w3, ws[t * st2 + pix], PM, GAM, (Real *)nullptr, cl);
        KW_(gz, nz) = z_bot;
      }
    });
    };
    if (riem_reg_arrays(g))
      go_(std::true_type{});
    else
      go_(std::false_type{});
    return fv3_post(c, s, "riem_solver_c");
  }
  launch2(c, s, Box{0, g.nx + 1, 0, g.ny + 1, 0, 0}, [=] FV3_HD(int t, int i, int j) {
    const long tb = t * g.st;
    const unsigned pix = IX(i, j);
    const long p = tb + pix;
    (void)p;
    // setup (inside the solver's first sweep): layer-mean pressure without condensate, thickness
    Real peg = ptop;
    auto setup = [&](int k, Real &pm, Real &dz) {
      const Real dm = K_(delpc, k);
      const Real peg_n = peg + dm * ((Real)1.0 - K_(q_con, k));
      pm = (peg_n - peg) / fv3_log(peg_n / peg);
      dz = K_(gz, k + 1) - K_(gz, k);
      K_(PM, k) = pm;
      K_(DZ, k) = dz;
      peg = peg_n;
    };
    // finish (inside the last sweep): geopotential from the new thickness
    Real z = phis[t * g.st2 + IX(i, j)];
    const Real z_bot = z;
    auto finish = [&](int k, Real dz) {
      z = z - dz * grav;
      K_(gz, k) = z;
    };
    sim.run(tb, pix, dt2, delpc, cappa, ptc, w3, ws[t * g.st2 + IX(i, j)], PM, DZ, W2, PP, GAM, pef, (Real *)nullptr, setup, finish);
    K_(gz, nz) = z_bot;  // (after the solve: setup reads the incoming gz[nz])
    // full interface pressure
    Real pem = ptop;
    K_(pef, 0) = ptop;
    for (int k = 0; k < nz; ++k) {
      pem = pem + K_(delpc, k);
      K_(pef, k + 1) = K_(pef, k + 1) + pem;
    }
  });
  return fv3_post(c, s, "riem_solver_c");
}

extern "C" int fv3_riem_solver3(fv3_ctx *c, int last_call, double dtd, const fv3_field *cappa_, double ptopd, const fv3_field *zs_, const fv3_field *wsd_,
                                const fv3_field *delz_, const fv3_field *q_con_, const fv3_field *delp_, const fv3_field *pt_, const fv3_field *zh_,
                                const fv3_field *pe_, const fv3_field *ppe_, const fv3_field *pk3_, const fv3_field *pk_, const fv3_field *peln_,
                                const fv3_field *w_, void *stream) {
  if (!c) return FV3_ERR_ARG;
  FV3_FIELD(cappa, cappa_) FV3_FIELD2D(zs, zs_) FV3_FIELD2D(wsd, wsd_) FV3_FIELD(delz, delz_) FV3_FIELD(q_con, q_con_) FV3_FIELD(delp, delp_)
  FV3_FIELD(pt, pt_) FV3_FIELD(zh, zh_) FV3_FIELD(pe, pe_) FV3_FIELD(ppe, ppe_) FV3_FIELD(pk3, pk3_) FV3_FIELD(pk, pk_) FV3_FIELD(peln, peln_)
  FV3_FIELD(w, w_)
  const Geo g = c->g;
  fv3_stream_t s = (fv3_stream_t)stream;
  const Real dt = (Real)dtd, ptop = (Real)ptopd;
  const Real akap = (Real)(c->cst.rdgas / c->cst.cp_air);
  Sim1 sim{g, (Real)c->cst.rdgas, (Real)1.0 / (Real)c->cst.grav, (Real)c->cfg.p_fac, ptop};
  Real *PM = c->scratch[SC_A], *W2 = c->scratch[SC_C], *PP = c->scratch[SC_D], *GAM = c->scratch[SC_E];
  const int nz = g.nz;
  const bool last = last_call != 0;
  if (riem_wave_ok(g, true)) {
    const Sim1W sw{g.sk, nz, sim.rgas, sim.rgrav, sim.p_fac, ptop};
    const int i0 = 1, j0 = 1, ni = g.nx, ncol = ni * g.ny;
    const long st = g.st, st2 = g.st2, sk = g.sk;
    const int sj32 = g.sj32, go = g.o;
    const bool gl = riem_gam_lds(g);
    // (two instantiations: the last sub-step also stores pe / peln / pk -- three more base pointers in a kernel that spills scalar
    //  registers as it is)
    // Frame-first passes of the sequencer (a transport is present): the columns are numbered frame first (ColumnOrder), pass 1 solves
    // the waves that hold frame columns, the halo updates of zh / pkc start, pass 2 solves the rest beside them.  Columns are
    // independent: the same values in any order.
    const ColumnOrder ord{g.nx, g.ny, c->frame_pass != 0 ? FV3_FRAME_W : 0};
    const int nwave = (ncol + FV3_WAVE - 1) / FV3_WAVE;
    const int nwave_frame = c->frame_pass != 0 ? (ord.n_frame() + FV3_WAVE - 1) / FV3_WAVE : 0;
    const int w_lo = c->frame_pass == 2 ? nwave_frame : 0, w_hi = c->frame_pass == 1 ? nwave_frame : nwave;  // waves [w_lo, w_hi)
    Real *const zn = c->dz_scan_src;  // (update_dz_d left its scan to this call: see fv3_ctx::seq_dz_scan)
    // Round 6: inside the sequencer only the LAST sub-step of an acoustic call stores the layer thickness -- the sub-steps between work from zh, and what reads delz
    // (the heights of the next call, the diffusive heating, the remap) comes after the last one: one field write less in five of six sub-steps.  FV3_SEQ_DELZ=every: A/B.
    const char *sde = getenv("FV3_SEQ_DELZ");
    const bool store_delz = !(c->seq_delz_dead && !(sde && !strcmp(sde, "every")));
    const bool pre = zn != nullptr;
    const Real dzm = c->dz_scan_min;
    auto go_ = [&](auto last_tag, auto ra_tag, auto pre_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    constexpr bool RA = decltype(ra_tag)::value;
    constexpr bool PRE = decltype(pre_tag)::value;
    launch_waves<1>(c, s, w_hi - w_lo, 1, g.nsub, sizeof(Real) * nz * FV3_WAVE * (gl ? 2 : 1), [=] FV3_HD(const Blk &blk, char *smem_) {
      const int t = blk.bz;
      const long tb = t * st;
      FV3_LANES(blk, lane, l) {
        const int cidx = (w_lo + blk.bx) * FV3_WAVE + lane;
        if (cidx >= ncol) continue;
        int ci, cj;
        ord.at(cidx, ci, cj);
        const unsigned pix = (unsigned)((cj + go) * sj32 + ci + go);
        struct Cl {
          long tb, sk;
          unsigned pix;
          Real pem, peg, pelng_k, z, akap;
          Real *pk3, *peln, *pk, *pe, *ppe, *zh, *delz;
          bool sdz;
          FV3_HD Real pm(int k, Real dm, Real qc) {
            pem = pem + dm;
            const Real peg_n = peg + dm * ((Real)1.0 - qc);
            const Real peln_n = fv3_log(pem), pelng_n = fv3_log(peg_n);
            const Real pk3v = fv3_exp(akap * peln_n);
            FV3_ST_NT(KW_(pk3, k + 1), pk3v);
            if constexpr (LAST) {
              FV3_ST_NT(KW_(peln, k + 1), peln_n);
              FV3_ST_NT(KW_(pk, k + 1), pk3v);
              FV3_ST_NT(KW_(pe, k + 1), pem);
            }
            const Real v = fv3_div(peg_n - peg, pelng_n - pelng_k);
            peg = peg_n;
            pelng_k = pelng_n;
            return v;
          }
          FV3_HD void out_pe(int k1, Real pev, Real) { FV3_ST_NT(KW_(ppe, k1), pev); }
          FV3_HD void finish(int k, Real dz) {
            z = z - dz;
            FV3_ST_NT(KW_(zh, k), z);
            if (sdz) FV3_ST_NT(KW_(delz, k), dz);  // (wave-uniform: see store_delz)
          }
        };
        const Real z_bot = zs[t * st2 + pix];
        const Real peln0 = fv3_log(ptop);
        Cl cl{tb, sk, pix, ptop, ptop, peln0, z_bot, akap, pk3, LAST ? peln : nullptr, LAST ? pk : nullptr, LAST ? pe : nullptr, ppe, zh, delz, store_delz};
        KW_(pk3, 0) = fv3_exp(akap * peln0);
        if constexpr (LAST) {
          KW_(peln, 0) = peln0;
          KW_(pk, 0) = KW_(pk3, 0);
          KW_(pe, 0) = ptop;
        }
        KW_(ppe, 0) = (Real)0;
        Real ws_v, z0 = (Real)0;
        if constexpr (PRE) {
          // update_dz_d's last kernel as a pre-sweep (bottom-up): interface k stays dz_min above interface k + 1; the limited heights go to the LDS line for
          // sweep 1 (z(k + 1) in slot k) and -- only where the limit changed them -- back into the marched field, which sweep 3 reads; the surface vertical
          // velocity from the lowest interface.  Same expressions as the scan kernel (fv3_update_dz_d): bitwise the separate form.
          Real *A_ = (Real *)smem_ + lane;
          Real below = KW_(zn, nz);
          A_[(nz - 1) * FV3_WAVE] = below;
          ws_v = (z_bot - below) / dt;
          wsd[t * st2 + pix] = ws_v;
          auto ldz = [&](int k) {
            KRec<1> q;
            q.v[0] = KW_(zn, k);
            return q;
          };
          KWALK(1, FV3_RIEM_U1, false, ldz, {
            const Real zk = r.v[0];
            const Real v = fv3_max(zk, below + dzm);
            if (v != zk) KW_(zn, K) = v;
            if (K >= 1)
              A_[(K - 1) * FV3_WAVE] = v;
            else
              z0 = v;
            below = v;
          })
        } else {
          ws_v = wsd[t * st2 + pix];
        }
        sw.run<FV3_RIEM_GL != 0, RA, PRE>((Real *)smem_ + lane, (Real *)smem_ + nz * FV3_WAVE + lane, tb, pix, dt, delp, cappa, pt, q_con, PRE ? (const Real *)zn : (const Real *)zh, w, ws_v, PM, GAM, w, cl, z0);
        KW_(zh, nz) = z_bot;
      }
    });
    };
    const bool ra = riem_reg_arrays(g);
    if (pre) {  // (the pre-sweep exists in the register-array forms only: fv3_update_dz_d defers under the same predicate)
      if (last)
        go_(std::true_type{}, std::true_type{}, std::true_type{});
      else
        go_(std::false_type{}, std::true_type{}, std::true_type{});
      if (c->frame_pass != 1) c->dz_scan_src = nullptr;  // (frame-first passes: both take the pre-sweep, each on its own columns)
    } else if (last && ra)
      go_(std::true_type{}, std::true_type{}, std::false_type{});
    else if (last)
      go_(std::true_type{}, std::false_type{}, std::false_type{});
    else if (ra)
      go_(std::false_type{}, std::true_type{}, std::false_type{});
    else
      go_(std::false_type{}, std::false_type{}, std::false_type{});
    return fv3_post(c, s, "riem_solver3");
  }
  launch2_pass(c, s, Box{1, g.nx, 1, g.ny, 0, 0}, c->frame_pass, [=] FV3_HD(int t, int i, int j) {
    const long tb = t * g.st;
    const unsigned pix = IX(i, j);
    const long p = tb + pix;
    (void)p;
    Real pem = ptop, peg = ptop;
    Real peln_k = fv3_log(pem), pelng_k = fv3_log(peg);
    K_(pk3, 0) = fv3_exp(akap * peln_k);
    if (last) {
      K_(peln, 0) = peln_k;
      K_(pk, 0) = K_(pk3, 0);
      K_(pe, 0) = pem;
    }
    (void)peln_k;
    // setup (inside the solver's first sweep): interface pressures / Exner functions, layer-mean pressure, thickness
    auto setup = [&](int k, Real &pm, Real &dz) {
      const Real dm = K_(delp, k);
      pem = pem + dm;
      const Real peg_n = peg + dm * ((Real)1.0 - K_(q_con, k));
      const Real peln_n = fv3_log(pem), pelng_n = fv3_log(peg_n);
      const Real pk3v = fv3_exp(akap * peln_n);
      K_(pk3, k + 1) = pk3v;
      if (last) {
        K_(peln, k + 1) = peln_n;
        K_(pk, k + 1) = pk3v;
        K_(pe, k + 1) = pem;
      }
      pm = (peg_n - peg) / (pelng_n - pelng_k);
      dz = K_(zh, k + 1) - K_(zh, k);
      K_(PM, k) = pm;
      K_(delz, k) = dz;  // dz2 lives in the output array
      peg = peg_n;
      pelng_k = pelng_n;
    };
    // finish (inside the last sweep): interface heights from the new thickness
    Real z = zs[t * g.st2 + IX(i, j)];
    const Real z_bot = z;
    auto finish = [&](int k, Real dz) {
      z = z - dz;
      K_(zh, k) = z;
    };
    sim.run(tb, pix, dt, delp, cappa, pt, w, wsd[t * g.st2 + IX(i, j)], PM, delz, W2, PP, GAM, ppe, w, setup, finish);
    K_(zh, nz) = z_bot;
  });
  return fv3_post(c, s, "riem_solver3");
}

// ---------------------------------------------------------------------------------------------
extern "C" int fv3_p_grad_c(fv3_ctx *c, const fv3_field *uc_, const fv3_field *vc_, const fv3_field *delpc_, const fv3_field *pkc_, const fv3_field *gz_,
                            double dt2d, void *stream) {
  if (!c) return FV3_ERR_ARG;
  FV3_FIELD(uc, uc_) FV3_FIELD(vc, vc_) FV3_FIELD(delpc, delpc_) FV3_FIELD(pkc, pkc_) FV3_FIELD(gz, gz_)
  const Geo g = c->g;
  const Real dt2 = (Real)dt2d;
  // A thread walks PG_KC levels keeping the lower-interface values (gz, pkc at k+1, three points each) in
  // registers for the next level: every interface plane is read once instead of twice (9 -> 7 field passes).
  // (rolled loop: an unrolled level loop lets the compiler hoist every level's loads at once)
  const int nz = g.nz, nchunk = (nz + PG_KC - 1) / PG_KC;
  launch3_pass(c, (fv3_stream_t)stream, Box{1, g.nx + 1, 1, g.ny + 1, 0, nchunk - 1}, c->frame_pass, [=] FV3_HD(int t, int kc, int i, int j) {
    const long m2 = t * g.st2;
    const unsigned p = IX(i, j), px = IX(i - 1, j), py = IX(i, j - 1);
    const bool do_u = j <= g.ny, do_v = i <= g.nx;
    const Real rx = (g.rdxc + m2)[p], ry = (g.rdyc + m2)[p];
    const int ka = kc * PG_KC, kb = ka + PG_KC < nz ? ka + PG_KC : nz;
    long b = t * g.st + ka * g.sk;
    Real g0 = (gz + b)[p], gx0 = (gz + b)[px], gy0 = (gz + b)[py];
    Real k0 = (pkc + b)[p], kx0 = (pkc + b)[px], ky0 = (pkc + b)[py];
#pragma unroll 1
    for (int k = ka; k < kb; ++k) {
      const long b1 = b + g.sk;
      const Real g1 = (gz + b1)[p], gx1 = (gz + b1)[px], gy1 = (gz + b1)[py];
      const Real k1 = (pkc + b1)[p], kx1 = (pkc + b1)[px], ky1 = (pkc + b1)[py];
      const Real dpc = (delpc + b)[p];
      if (do_u) (uc + b)[p] = (uc + b)[p] + dt2 * rx / ((delpc + b)[px] + dpc) * ((gx1 - g0) * (k1 - kx0) + (gx0 - g1) * (kx1 - k0));
      if (do_v) (vc + b)[p] = (vc + b)[p] + dt2 * ry / ((delpc + b)[py] + dpc) * ((gy1 - g0) * (k1 - ky0) + (gy0 - g1) * (ky1 - k0));
      g0 = g1;
      gx0 = gx1;
      gy0 = gy1;
      k0 = k1;
      kx0 = kx1;
      ky0 = ky1;
      b = b1;
    }
  });
  return fv3_post(c, (fv3_stream_t)stream, "p_grad_c");
}

extern "C" int fv3_nh_p_grad(fv3_ctx *c, const fv3_field *u_, const fv3_field *v_, const fv3_field *pp_, const fv3_field *gz_, const fv3_field *pk3_,
                             const fv3_field *delp_, double dtd, double ptop, double akap, void *stream) {
  return fv3_nh_p_grad_scaled(c, u_, v_, pp_, gz_, pk3_, delp_, dtd, ptop, akap, 1.0, stream);
}

// gz_scale: the sequencer passes the interface heights zh with gz_scale = g instead of first storing
// gz = g * zh (compute_geopotential): the product is formed where a2b_ord4 reads its input (same bits).
int fv3_nh_p_grad_scaled(fv3_ctx *c, const fv3_field *u_, const fv3_field *v_, const fv3_field *pp_, const fv3_field *gz_, const fv3_field *pk3_,
                         const fv3_field *delp_, double dtd, double ptop, double akap, double gz_scale, void *stream) {
  if (!c) return FV3_ERR_ARG;
  FV3_FIELD(u, u_) FV3_FIELD(v, v_) FV3_FIELD(pp, pp_) FV3_FIELD(gz, gz_) FV3_FIELD(pk3, pk3_) FV3_FIELD(delp, delp_)
  const Geo g = c->g;
  fv3_stream_t s = (fv3_stream_t)stream;
  const Real dt = (Real)dtd;
  const Real top = (Real)std::pow(ptop, akap);
  const int nz = g.nz;
  // corner-interpolated pp / pk3 / gz / delp go to scratch: the reference interpolates pp, pk3 and gz in
  // place, but nothing reads them afterwards (gz and pk3 are rebuilt every sub-step), so the three
  // copy-back passes are not spent; the inputs come back unchanged.
  // product form: the four corner interpolations and the wind update as one marching kernel (fv3_pgf.hip); FV3_NH_PGF=staged keeps
  // the four a2b_ord4 launches + the level-walking update below (A/B reference)
  {
    const char *e = getenv("FV3_NH_PGF");  // (read per call: the A/B parity test flips it in one process)
    if (!(e && !strcmp(e, "staged"))) {
      nh_pgf_fused(c, s, pp, pk3, gz, delp, u, v, dt, top, (Real)gz_scale, c->frame_pass);
      return fv3_post(c, s, "nh_p_grad");
    }
  }
  Real *ppb = c->scratch[SC_B], *pk3b = c->scratch[SC_C], *gzb = c->scratch[SC_D], *wk1 = c->scratch[SC_A];
  if (c->frame_pass != 2) {  // (interior pass of the frame-first form: the corner fields are in scratch already)
    launch2(c, s, Box{1, g.nx + 1, 1, g.ny + 1, 0, 0}, [=] FV3_HD(int t, int i, int j) {
      const long p = t * g.st + IX(i, j);
      ppb[p] = (Real)0;
      pk3b[p] = top;
    });
    a2b_ord4(c, s, pp, ppb, 1, 1, nz, false);
    a2b_ord4(c, s, pk3, pk3b, 1, 1, nz, false);
    a2b_ord4(c, s, gz, gzb, 0, 0, nz + 1, false, (Real)gz_scale);
    a2b_ord4(c, s, delp, wk1, 0, 0, nz, false);
  }
  // A thread walks PG_KC levels keeping the lower-interface corner values (pk3, gz, pp at k+1, three corners each)
  // in registers for the next level: every interface plane is read once instead of twice (11 -> 8 field passes).
  const int nchunk = (nz + PG_KC - 1) / PG_KC;
  launch3_pass(c, s, Box{1, g.nx + 1, 1, g.ny + 1, 0, nchunk - 1}, c->frame_pass, [=] FV3_HD(int t, int kc, int i, int j) {
    const long m2 = t * g.st2;
    const unsigned p = IX(i, j), pe_ = IX(i + 1, j), pn = IX(i, j + 1);
    const bool do_u = i <= g.nx, do_v = j <= g.ny;
    const Real rx = (g.rdx + m2)[p], ry = (g.rdy + m2)[p];
    const int ka = kc * PG_KC, kb = ka + PG_KC < nz ? ka + PG_KC : nz;
    long b = t * g.st + ka * g.sk;
    Real k0 = (pk3b + b)[p], ke0 = (pk3b + b)[pe_], kn0 = (pk3b + b)[pn];
    Real g0 = (gzb + b)[p], ge0 = (gzb + b)[pe_], gn0 = (gzb + b)[pn];
    Real q0 = (ppb + b)[p], qe0 = (ppb + b)[pe_], qn0 = (ppb + b)[pn];
#pragma unroll 1
    for (int k = ka; k < kb; ++k) {
      const long b1 = b + g.sk;
      const Real k1 = (pk3b + b1)[p], ke1 = (pk3b + b1)[pe_], kn1 = (pk3b + b1)[pn];
      const Real g1 = (gzb + b1)[p], ge1 = (gzb + b1)[pe_], gn1 = (gzb + b1)[pn];
      const Real q1 = (ppb + b1)[p], qe1 = (ppb + b1)[pe_], qn1 = (ppb + b1)[pn];
      const Real wkp = k1 - k0, w1p = (wk1 + b)[p];
      if (do_u) {
        const Real du = dt / (wkp + (ke1 - ke0)) * ((g1 - ge0) * (ke1 - k0) + (g0 - ge1) * (k1 - ke0));
        (u + b)[p] = ((u + b)[p] + du + dt / (w1p + (wk1 + b)[pe_]) * ((g1 - ge0) * (qe1 - q0) + (g0 - ge1) * (q1 - qe0))) * rx;
      }
      if (do_v) {
        const Real dv = dt / (wkp + (kn1 - kn0)) * ((g1 - gn0) * (kn1 - k0) + (g0 - gn1) * (k1 - kn0));
        (v + b)[p] = ((v + b)[p] + dv + dt / (w1p + (wk1 + b)[pn]) * ((g1 - gn0) * (qn1 - q0) + (g0 - gn1) * (q1 - qn0))) * ry;
      }
      k0 = k1;
      ke0 = ke1;
      kn0 = kn1;
      g0 = g1;
      ge0 = ge1;
      gn0 = gn1;
      q0 = q1;
      qe0 = qe1;
      qn0 = qn1;
      b = b1;
    }
  });
  return fv3_post(c, s, "nh_p_grad");
}

// ---------------------------------------------------------------------------------------------
extern "C" int fv3_pk3_halo(fv3_ctx *c, const fv3_field *pk3_, const fv3_field *delp_, double ptopd, double akapd, void *stream) {
  if (!c) return FV3_ERR_ARG;
  FV3_FIELD(pk3, pk3_) FV3_FIELD(delp, delp_)
  const Geo g = c->g;
  const Real ptop = (Real)ptopd, akap = (Real)akapd;
  // the two-cell ring around the compute domain only: S / N strips as they lie, W / E strips transposed (lanes along j)
  // -- a launch over the whole padded plane spends its time on threads that exit (0.49 -> 0.2 ms at C768)
  auto column = [=] FV3_HD(int t, int i, int j) {
    const long tb = t * g.st;
    const unsigned pix = IX(i, j);
    Real pei = ptop;
    for (int k = 0; k < g.nz; ++k) {
      pei = pei + K_(delp, k);
      K_(pk3, k + 1) = fv3_exp(akap * fv3_log(pei));
    }
  };
  fv3_stream_t s = (fv3_stream_t)stream;
  const int nx = g.nx, ny = g.ny;
  launch2(c, s, Box{-1, nx + 2, -1, 0, 0, 0}, column);
  launch2(c, s, Box{-1, nx + 2, ny + 1, ny + 2, 0, 0}, column);
  launch2(c, s, Box{1, ny, 0, 1, 0, 0}, [=] FV3_HD(int t, int a, int b_) { column(t, b_ - 1, a); });
  launch2(c, s, Box{1, ny, 0, 1, 0, 0}, [=] FV3_HD(int t, int a, int b_) { column(t, nx + 1 + b_, a); });
  return fv3_post(c, (fv3_stream_t)stream, "pk3_halo");
}

extern "C" int fv3_edge_pe(fv3_ctx *c, const fv3_field *pe_, const fv3_field *delp_, double ptopd, void *stream) {
  if (!c) return FV3_ERR_ARG;
  FV3_FIELD(pe, pe_) FV3_FIELD(delp, delp_)
  const Geo g = c->g;
  const Real ptop = (Real)ptopd;
  launch2(c, (fv3_stream_t)stream, Box{0, g.nx + 1, 0, g.ny + 1, 0, 0}, [=] FV3_HD(int t, int i, int j) {
    if (i >= 1 && i <= g.nx && j >= 1 && j <= g.ny) return;
    const long tb = t * g.st;
    const unsigned pix = IX(i, j);
    const long p = tb + pix;
    (void)p;
    Real pei = ptop;
    K_(pe, 0) = pei;
    for (int k = 0; k < g.nz; ++k) {
      pei = pei + K_(delp, k);
      K_(pe, k + 1) = pei;
    }
  });
  return fv3_post(c, (fv3_stream_t)stream, "edge_pe");
}

// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// edge_profile with the whole column in registers (level count known at compile time): the NZ layer
// values are loaded up front (NZ independent loads in flight per lane), the forward elimination and the
// back substitution run in place on the register array and the NZ+1 interface values are stored once --
// one read and one write per field instead of the two column walks of the generic form below.
// ---------------------------------------------------------------------------------------------
template <int NZ>
static void edge_profile_reg(fv3_ctx *c, fv3_stream_t s, const Real *crx, const Real *xfx, const Real *cry, const Real *yfx, Real *crx_a, Real *xfx_a, Real *cry_a,
                             Real *yfx_a) {
  const Geo g = c->g;
  const int isd = 1 - g.nh, ied = g.nx + g.nh, jsd = 1 - g.nh, jed = g.ny + g.nh;
  const Real *gamd = g.ep_gam;
  launch2(c, s, Box{isd, ied, jsd, jed, 0, 0}, [=] FV3_HD(int t, int i, int j) {
    const long tb = t * g.st;
    const unsigned pix = IX(i, j);
    auto profile = [&](const Real *q, Real *qe) {
      const Real *dp0 = g.dp_ref;
      Real a[NZ + 1];
#pragma unroll
      for (int k = 0; k < NZ; ++k) a[k] = K_(q, k);
      const Real g0 = dp0[1] / dp0[0];
      Real xt1 = (Real)2.0 * g0 * (g0 + (Real)1.0);
      Real bet = g0 * (g0 + (Real)0.5);
      Real q_m = a[0], q_mm = a[0];  // q[k-1], q[k-2] (original layer values)
      a[0] = (xt1 * a[0] + a[1]) / bet;
      Real gam_prev = ((Real)1.0 + g0 * (g0 + (Real)1.5)) / bet;
      Real gk = g0;
#pragma unroll
      for (int k = 1; k < NZ; ++k) {
        gk = dp0[k - 1] / dp0[k];
        bet = (Real)2.0 + (Real)2.0 * gk - gam_prev;
        const Real qk = a[k];
        a[k] = ((Real)3.0 * (q_m + gk * qk) - a[k - 1]) / bet;
        gam_prev = gk / bet;
        q_mm = q_m;
        q_m = qk;
      }
      const Real a_bot = (Real)1.0 + gk * (gk + (Real)1.5);
      xt1 = (Real)2.0 * gk * (gk + (Real)1.0);
      const Real xt2 = gk * (gk + (Real)0.5) - a_bot * gam_prev;
      a[NZ] = (xt1 * q_m + q_mm - a_bot * a[NZ - 1]) / xt2;
#pragma unroll
      for (int k = NZ - 1; k >= 0; --k) a[k] = a[k] - gamd[k] * a[k + 1];
#pragma unroll
      for (int k = 0; k <= NZ; ++k) K_(qe, k) = a[k];
    };
    if (i >= 1 && i <= g.nx + 1) {
      profile(crx, crx_a);
      profile(xfx, xfx_a);
    }
    if (j >= 1 && j <= g.ny + 1) {
      profile(cry, cry_a);
      profile(yfx, yfx_a);
    }
  });
}

// Wave forms of edge_profile: a wave owns 64 columns of ONE of the four fields (flattened column index: no
// idle lanes on the face-field boxes).  The pivots / ratios are level-only tables built at context
// creation, so a level costs one division.
//   NZ > 0: the column lives in registers, all NZ loads in flight at once, code = one unrolled profile
//           (the older edge_profile_reg inlines four profiles with three divisions per level: ~130 KB of
//           code, 2.5 TB/s);
//   NZ = 0: any level count -- rolled loops, the forward-eliminated values sit in the lane's LDS line,
//           inputs prefetched 16 levels ahead (LDS-limited to 4 waves / CU at L79: latency-bound, 3.3 TB/s).
struct EpSet {  // the fields of one launch as element offsets from the first one's pointers
  long din[4], dout[4];
  int n;
};
template <int NZ>
static void edge_profile_wave1(fv3_ctx *c, fv3_stream_t s, const Real *q, Real *qe, bool xf, EpSet set = EpSet{{0, 0, 0, 0}, {0, 0, 0, 0}, 1});
template <int NZ>
static void edge_profile_wave(fv3_ctx *c, fv3_stream_t s, const Real *crx, const Real *xfx, const Real *cry, const Real *yfx, Real *crx_a, Real *xfx_a, Real *cry_a,
                              Real *yfx_a) {
  // one launch per field: the field pointers stay kernel arguments (selecting among them inside the kernel
  // turns every access into a flat load through a scratch copy of the argument block)
  // Round 5 (review item 2): the four solves as ONE launch -- the field of a wave is a wave-uniform ELEMENT OFFSET from the first field's pointer (four
  // integers among the kernel arguments), so every access stays a global load off one base.  Experiment R5-22 (update_dz_d -0.18 ms, but d_sw +0.5 ms in the
  // same processes: left off); re-measured in round 6 on the fused wind stage (R6-8: update_dz_d -0.11 / -0.20 ms, d_sw -0.0 / -0.4): the default now.
  // FV3_EP_ONE_LAUNCH=0: four launches (A/B; bitwise equal).
  static const bool one = !(getenv("FV3_EP_ONE_LAUNCH") && getenv("FV3_EP_ONE_LAUNCH")[0] == '0');
  if (one) {
    const EpSet set{{0, (long)(xfx - crx), (long)(cry - crx), (long)(yfx - crx)}, {0, (long)(xfx_a - crx_a), (long)(cry_a - crx_a), (long)(yfx_a - crx_a)}, 4};
    edge_profile_wave1<NZ>(c, s, crx, crx_a, true, set);
    return;
  }
  edge_profile_wave1<NZ>(c, s, crx, crx_a, true);
  edge_profile_wave1<NZ>(c, s, xfx, xfx_a, true);
  edge_profile_wave1<NZ>(c, s, cry, cry_a, false);
  edge_profile_wave1<NZ>(c, s, yfx, yfx_a, false);
}
template <int NZ>
static void edge_profile_wave1(fv3_ctx *c, fv3_stream_t s, const Real *q0, Real *qe0, bool xf0, EpSet set) {
  const Geo g = c->g;
  const int nz = g.nz;
  const int isd = 1 - g.nh, ied = g.nx + g.nh, jsd = 1 - g.nh, jed = g.ny + g.nh;
  const int nix = g.nx + 1, ncx = nix * (jed - jsd + 1);  // x-face fields: i in [1, nx+1], every j
  const int niy = ied - isd + 1, ncy = niy * (g.ny + 1);  // y-face fields: every i, j in [1, ny+1]
  const int nfield = set.n, nsub = g.nsub;
  const int ncmax = nfield > 1 ? (ncx > ncy ? ncx : ncy) : (xf0 ? ncx : ncy);
  const long st = g.st, sk = g.sk;
  const int sj32 = g.sj32, go = g.o;
  const Real *gkt = g.ep_gk, *bett = g.ep_bet, *gamd = g.ep_gam;
  constexpr int WPE = NZ == 0 ? 1 : (NZ > 100 ? 1 : 2);
  launch_waves<WPE>(c, s, (ncmax + FV3_WAVE - 1) / FV3_WAVE, 1, g.nsub * nfield, NZ == 0 ? sizeof(Real) * nz * FV3_WAVE : 0, [=] FV3_HD(const Blk &blk, char *smem_) {
    // (set of four: fields 0, 1 are x-face fields, 2, 3 y-face fields)
    const int f = nfield > 1 ? blk.bz / nsub : 0;
    const bool xf = nfield > 1 ? f < 2 : xf0;
    const long din = f == 0 ? set.din[0] : f == 1 ? set.din[1] : f == 2 ? set.din[2] : set.din[3];
    const long dout = f == 0 ? set.dout[0] : f == 1 ? set.dout[1] : f == 2 ? set.dout[2] : set.dout[3];
    const Real *const q = q0 + din;
    Real *const qe = qe0 + dout;
    const int ni = xf ? nix : niy, ncol = xf ? ncx : ncy, i0 = xf ? 1 : isd, j0 = xf ? jsd : 1;
    const long tb = (blk.bz - f * nsub) * st;
    FV3_LANES(blk, lane, l) {
      const int cidx = blk.bx * FV3_WAVE + lane;
      if (cidx >= ncol) continue;
      const int jr = cidx / ni;
      const unsigned pix = (unsigned)((j0 + jr + go) * sj32 + (i0 + cidx - jr * ni) + go);
      if constexpr (NZ > 0) {
        (void)smem_;
        Real a[NZ + 1];
#pragma unroll
        for (int k = 0; k < NZ; ++k) a[k] = KW_(q, k);
        const Real g0 = gkt[0];
        Real q_m = a[0], q_mm = a[0];
        a[0] = ((Real)2.0 * g0 * (g0 + (Real)1.0) * a[0] + a[1]) / bett[0];
#pragma unroll
        for (int k = 1; k < NZ; ++k) {
          const Real qk = a[k];
          a[k] = ((Real)3.0 * (q_m + gkt[k] * qk) - a[k - 1]) / bett[k];
          q_mm = q_m;
          q_m = qk;
        }
        const Real gk = gkt[NZ - 1], gam_prev = gamd[NZ - 1];
        const Real a_bot = (Real)1.0 + gk * (gk + (Real)1.5);
        const Real xt1 = (Real)2.0 * gk * (gk + (Real)1.0);
        const Real xt2 = gk * (gk + (Real)0.5) - a_bot * gam_prev;
        a[NZ] = (xt1 * q_m + q_mm - a_bot * a[NZ - 1]) / xt2;
#pragma unroll
        for (int k = NZ - 1; k >= 0; --k) a[k] = a[k] - gamd[k] * a[k + 1];
#pragma unroll
        for (int k = 0; k <= NZ; ++k) KW_(qe, k) = a[k];
      } else {
        Real *A = (Real *)smem_ + lane;
        Real q_m = (Real)0, q_mm = (Real)0, a_prev = (Real)0;
        k_walk<1, 16, true>(
            nz,
            [&](int k) {
              KRec<1> r;
              r.v[0] = KW_(q, k);
              return r;
            },
            [&](int k, const KRec<1> &r) {
              const Real qk = r.v[0];
              if (k == 0) {
                q_m = q_mm = qk;
                return;
              }
              if (k == 1) {
                const Real g0 = gkt[0];
                a_prev = ((Real)2.0 * g0 * (g0 + (Real)1.0) * q_m + qk) / bett[0];
                A[0] = a_prev;
              }
              const Real gk = gkt[k];
              a_prev = ((Real)3.0 * (q_m + gk * qk) - a_prev) / bett[k];
              A[k * FV3_WAVE] = a_prev;
              q_mm = q_m;
              q_m = qk;
            });
        const Real gk = gkt[nz - 1], gam_prev = gamd[nz - 1];
        const Real a_bot = (Real)1.0 + gk * (gk + (Real)1.5);
        const Real xt1 = (Real)2.0 * gk * (gk + (Real)1.0);
        const Real xt2 = gk * (gk + (Real)0.5) - a_bot * gam_prev;
        Real a_next = (xt1 * q_m + q_mm - a_bot * a_prev) / xt2;
        KW_(qe, nz) = a_next;
        for (int k = nz - 1; k >= 0; --k) {
          a_next = A[k * FV3_WAVE] - gamd[k] * a_next;
          KW_(qe, k) = a_next;
        }
      }
    }
  });
}

// returns false when the level count has no register-resident instantiation (the caller then runs the generic form)
static bool edge_profile_columns(fv3_ctx *c, fv3_stream_t s, const Real *crx, const Real *xfx, const Real *cry, const Real *yfx, Real *crx_a, Real *xfx_a, Real *cry_a,
                                 Real *yfx_a) {
  static const bool generic = getenv("FV3_EDGE_PROFILE_GENERIC") != nullptr;  // A/B switches
  static const bool regs = getenv("FV3_EDGE_PROFILE_REG") != nullptr;
  if (generic) return false;
  if (!regs && c->g.nz >= 2) {
    static const bool lds = getenv("FV3_EDGE_PROFILE_LDS") != nullptr;
    if (c->g.nz == 79 && !lds) {
      edge_profile_wave<79>(c, s, crx, xfx, cry, yfx, crx_a, xfx_a, cry_a, yfx_a);
      return true;
    }
    if (c->g.nz == 127 && !lds) {
      edge_profile_wave<127>(c, s, crx, xfx, cry, yfx, crx_a, xfx_a, cry_a, yfx_a);
      return true;
    }
    if (sizeof(Real) * c->g.nz * FV3_WAVE <= 64 * 1024) {
      edge_profile_wave<0>(c, s, crx, xfx, cry, yfx, crx_a, xfx_a, cry_a, yfx_a);
      return true;
    }
  }
  switch (c->g.nz) {
    case 79:
      edge_profile_reg<79>(c, s, crx, xfx, cry, yfx, crx_a, xfx_a, cry_a, yfx_a);
      return true;
    case 127:
      edge_profile_reg<127>(c, s, crx, xfx, cry, yfx, crx_a, xfx_a, cry_a, yfx_a);
      return true;
    case 8:  // exercised by the parity tests
      edge_profile_reg<8>(c, s, crx, xfx, cry, yfx, crx_a, xfx_a, cry_a, yfx_a);
      return true;
    default:
      return false;
  }
}

extern "C" int fv3_update_dz_d(fv3_ctx *c, const fv3_field *zs_, const fv3_field *zh_, const fv3_field *crx_, const fv3_field *cry_, const fv3_field *xfx_,
                               const fv3_field *yfx_, const fv3_field *wsd_, double dtd, void *stream) {
  if (!c) return FV3_ERR_ARG;
  FV3_FIELD2D(zs, zs_) FV3_FIELD(zh, zh_) FV3_FIELD(crx, crx_) FV3_FIELD(cry, cry_) FV3_FIELD(xfx, xfx_) FV3_FIELD(yfx, yfx_) FV3_FIELD2D(wsd, wsd_)
  const Geo g = c->g;
  fv3_stream_t s = (fv3_stream_t)stream;
  const Real dt = (Real)dtd;
  const int nz = g.nz;
  const Real dz_min = (Real)c->cst.dz_min;
  Real *crx_a = c->scratch[SC_A], *xfx_a = c->scratch[SC_B], *cry_a = c->scratch[SC_C], *yfx_a = c->scratch[SC_D];
  Real *fx = c->scratch[SC_E], *fy = c->scratch[SC_F], *fx2 = c->scratch[SC_G], *fy2 = c->scratch[SC_H], *d2 = c->scratch[SC_I];
  const int isd = 1 - g.nh, ied = g.nx + g.nh, jsd = 1 - g.nh, jed = g.ny + g.nh;
  // cubic-spline-like layer -> interface interpolation (FV3 edge_profile, limiter 0)
  if (!edge_profile_columns(c, s, crx, xfx, cry, yfx, crx_a, xfx_a, cry_a, yfx_a)) {
    launch2(c, s, Box{isd, ied, jsd, jed, 0, 0}, [=] FV3_HD(int t, int i, int j) {
      const long tb = t * g.st;
      const unsigned pix = IX(i, j);
      const long p = tb + pix;
      (void)p;
      auto profile = [&](const Real *q, Real *qe) {
        const Real *dp0 = g.dp_ref;
        const Real g0 = dp0[1] / dp0[0];
        Real xt1 = (Real)2.0 * g0 * (g0 + (Real)1.0);
        Real bet = g0 * (g0 + (Real)0.5);
        K_(qe, 0) = (xt1 * K_(q, 0) + K_(q, 1)) / bet;
        Real gam_prev = ((Real)1.0 + g0 * (g0 + (Real)1.5)) / bet;
        Real gk = g0;
        // gam[k] is level-only: recomputed in the backward sweep from the same recurrence
        for (int k = 1; k < nz; ++k) {
          gk = dp0[k - 1] / dp0[k];
          bet = (Real)2.0 + (Real)2.0 * gk - gam_prev;
          K_(qe, k) = ((Real)3.0 * (K_(q, k - 1) + gk * K_(q, k)) - K_(qe, k - 1)) / bet;
          gam_prev = gk / bet;
        }
        const Real a_bot = (Real)1.0 + gk * (gk + (Real)1.5);
        xt1 = (Real)2.0 * gk * (gk + (Real)1.0);
        const Real xt2 = gk * (gk + (Real)0.5) - a_bot * gam_prev;
        K_(qe, nz) = (xt1 * K_(q, nz - 1) + K_(q, nz - 2) - a_bot * K_(qe, nz - 1)) / xt2;
      };
      const bool inx = i >= 1 && i <= g.nx + 1, iny = j >= 1 && j <= g.ny + 1;
      if (inx) {
        profile(crx, crx_a);
        profile(xfx, xfx_a);
      }
      if (iny) {
        profile(cry, cry_a);
        profile(yfx, yfx_a);
      }
    });
    // backward sweep: gam[k] depends on dp_ref only (table built at context creation)
    {
      const Real *gamd = g.ep_gam;
      launch2(c, s, Box{isd, ied, jsd, jed, 0, 0}, [=] FV3_HD(int t, int i, int j) {
        const long tb = t * g.st;
      const unsigned pix = IX(i, j);
      const long p = tb + pix;
      (void)p;
        const bool inx = i >= 1 && i <= g.nx + 1, iny = j >= 1 && j <= g.ny + 1;
        for (int k = nz - 1; k >= 0; --k) {
          const Real gm = gamd[k];
          if (inx) {
            K_(crx_a, k) = K_(crx_a, k) - gm * K_(crx_a, k + 1);
            K_(xfx_a, k) = K_(xfx_a, k) - gm * K_(xfx_a, k + 1);
          }
          if (iny) {
            K_(cry_a, k) = K_(cry_a, k) - gm * K_(cry_a, k + 1);
            K_(yfx_a, k) = K_(yfx_a, k) - gm * K_(yfx_a, k + 1);
          }
        }
      });
    }
  }
  // del-n damping fluxes of the interface heights, then their transport: the advective-form update and the
  // damping term are the transport kernel's epilogue (znew; the height fluxes are never stored)
  int nord_max = 0;
  for (int k = 0; k <= nz; ++k) nord_max = std::max(nord_max, c->nord_v_h[k]);
  // (FV3_ALT=dz_damp_scaled -- DESIGN §2, uncertain restatement 1: the coefficient d_sw's vorticity damping uses,
  //  (damp_vt * da_min_c)^(nord_v + 1), instead of the raw column value; the on / off test stays the raw coefficient)
  const Real *dz_coef = fv3_alt("dz_damp_scaled") ? c->tab.d6_vt : g.damp_vt;
  Deln dn{g.nord_v, dz_coef, g.damp_vt, 0, (Real)0, false, (Real)1.0e-5, nord_max};
  // Interfaces from fd_k0 on (chain of order 2 and switched on: all but the sponge layers) run the chain INSIDE the transport march
  // on the strips away from the W / E tile edges (tp2d_stream_t, TF_FD); del6_stream then only serves the tile-edge strips and
  // the cube-corner patches there.  FV3_DZ_DELN=arrays: the chain of every interface as one del6_stream launch (A/B reference).
  int fd_k0 = nz + 1;
  {
    const char *e = getenv("FV3_DZ_DELN"), *m = getenv("FV3_TP2D_MODE"), *m6 = getenv("FV3_DEL6_MODE");
    const bool off = (e && !strcmp(e, "arrays")) || (m && !strcmp(m, "staged")) || (m6 && !strcmp(m6, "staged"));
    if (!off)
      for (int k = nz; k >= 0; --k) {
        if (!(c->nord_v_h[k] == 2 && c->damp_vt_h[k] > 1.0e-5)) break;
        fd_k0 = k;
      }
  }
  // (the interfaces without the chain inside the march -- the sponge layers: their del-n launch and their small transport launch go
  //  to the auxiliary stream, beside the march of the others; events 2 = fork, 3 = join)
  fv3_stream_t sa = fd_k0 > 0 && fd_k0 <= nz ? fv3_aux(c, s) : s;
  if (sa != s) {
    fv3_signal(c, s, 2);
    fv3_wait(c, sa, 2);
  }
  del6_vt_flux(c, sa, zh, d2, fx2, fy2, dn, false, 0, fd_k0 - 1);
  if (tp2d_fd_lean(c, c->cfg.hord_tm, fd_k0, nz))  // (the round-5 march runs the chain on every strip: only the cube-corner patches come from the staged chain)
    del6_vt_flux_patches(c, s, zh, d2, fx2, fy2, dn, false, fd_k0, nz);
  else
    del6_vt_flux_edge_strips(c, s, zh, d2, fx2, fy2, dn, false, fd_k0, nz);
  // Round 6: inside the sequencer (fv3_ctx::seq_dz_scan) the scan at the end of this operator becomes the pre-sweep of riem_solver3's wave form, which reads the
  // marched heights where this call leaves them (the free slot SC_F: riem_solver3's scratch fields are SC_A / C / D / E).  FV3_DZ_SCAN=separate: the kernel below (A/B).
  const char *zse = getenv("FV3_DZ_SCAN");
  const bool scan_deferred = c->seq_dz_scan && riem_wave_ok(g, true) && riem_reg_arrays(g) && !(zse && !strcmp(zse, "separate"));
  Real *znew = scan_deferred ? fy : fx;
  {
    TpEpi e{znew, nullptr, false, nullptr, nullptr, nullptr, nullptr, nullptr, true, fx2, fy2, g.damp_vt, nullptr, nullptr};
    tp2d(c, sa, zh, crx_a, cry_a, xfx_a, yfx_a, c->scratch[SC_J], c->scratch[SC_K], nullptr, nullptr, nullptr, c->cfg.hord_tm, nullptr, 0, fd_k0 - 1, &e);
    if (sa != s) fv3_signal(c, sa, 3);
    e.fd = 1;
    e.fd_coef = dz_coef;
    tp2d(c, s, zh, crx_a, cry_a, xfx_a, yfx_a, c->scratch[SC_J], c->scratch[SC_K], nullptr, nullptr, nullptr, c->cfg.hord_tm, nullptr, fd_k0, nz, &e);
    if (sa != s) fv3_wait(c, s, 3);
  }
  if (getenv("FV3_DEBUG_FD")) fprintf(stderr, "[update_dz_d] closing scan: %s\n", scan_deferred ? "left to riem_solver3's pre-sweep" : "own kernel");
  if (scan_deferred) {
    c->dz_scan_src = znew;
    c->dz_scan_min = dz_min;
    return fv3_post(c, s, "update_dz_d");
  }
  launch2(c, s, Box{1, g.nx, 1, g.ny, 0, 0}, [=] FV3_HD(int t, int i, int j) {
    const long tb = t * g.st;
    const unsigned pix = IX(i, j);
    const long p = tb + pix;
    (void)p;
    Real below = K_(znew, nz);
    K_(zh, nz) = below;
    wsd[t * g.st2 + IX(i, j)] = (zs[t * g.st2 + IX(i, j)] - below) / dt;
    for (int k = nz - 1; k >= 0; --k) {
      const Real v = fv3_max(K_(znew, k), below + dz_min);
      K_(zh, k) = v;
      below = v;
    }
  });
  return fv3_post(c, s, "update_dz_d");
}

// ---------------------------------------------------------------------------------------------
extern "C" int fv3_ray_fast(fv3_ctx *c, const fv3_field *u_, const fv3_field *v_, const fv3_field *w_, double dt, double ptop, void *stream) {
  if (!c) return FV3_ERR_ARG;
  FV3_FIELD(u, u_) FV3_FIELD(v, v_) FV3_FIELD(w, w_)
  const Geo g = c->g;
  const int nz = g.nz;
  const double rf_cutoff = c->cfg.rf_cutoff;
  const double nudge = rf_cutoff + std::fmin(100.0, 10.0 * ptop);
  const double tau0 = c->cfg.tau * c->cst.seconds_per_day;
  // rf table for this (dt, ptop): host libm, cached in the context
  if (!c->tab_rf || c->rf_dt != dt || c->rf_ptop != ptop) {
    std::vector<Real> rf(nz, (Real)1);
    int nd = 0, nn = 0;
    double dm = 0.0;
    for (int k = 0; k < nz; ++k) {
      const double pf = c->pfull_h[k];
      if (pf < rf_cutoff) {
        const double sn = std::sin(0.5 * c->cst.pi * std::log(rf_cutoff / pf) / std::log(rf_cutoff / ptop));
        rf[k] = (Real)(1.0 / (1.0 + dt / tau0 * (sn * sn)));
        nd = k + 1;
      }
      if (pf < nudge) {
        dm += c->dp_ref_h[k];
        nn = k + 1;
      }
    }
    if (!c->tab_rf) {
      c->tab_rf = fv3_dev_alloc(c, nz * sizeof(Real));
      if (!c->tab_rf) return fv3_fail(c, FV3_ERR_NOMEM, "rf table");
    }
    fv3_h2d(c->tab_rf, rf.data(), nz * sizeof(Real));
    c->rf_dt = dt;
    c->rf_ptop = ptop;
    c->rf_nd = nd;
    c->rf_nn = nn;
    c->rf_dm = dm;
  }
  const int nd = c->rf_nd, nn = c->rf_nn;
  if (nn == 0) return FV3_OK;
  const Real dm = (Real)c->rf_dm;
  const Real *rf = (const Real *)c->tab_rf;
  const bool momentum_fix = !fv3_alt("ray_fast_plain");  // (FV3_ALT: DESIGN §2, uncertain restatement 4)
  launch2_pass(c, (fv3_stream_t)stream, Box{1, g.nx + 1, 1, g.ny + 1, 0, 0}, c->frame_pass, [=] FV3_HD(int t, int i, int j) {
    const long tb = t * g.st;
    const unsigned pix = IX(i, j);
    const long p = tb + pix;
    (void)p;
    auto wind = [&](Real *a) {
      Real dmdir = (Real)0;
      for (int k = 0; k < nd; ++k) {
        dmdir = dmdir + ((Real)1.0 - rf[k]) * g.dp_ref[k] * K_(a, k);
        K_(a, k) = K_(a, k) * rf[k];
      }
      const Real add = dmdir / dm;
      if (momentum_fix)
        for (int k = 0; k < nn; ++k) K_(a, k) = K_(a, k) + add;
    };
    if (i <= g.nx) wind(u);
    if (j <= g.ny) wind(v);
    if (i <= g.nx && j <= g.ny)
      for (int k = 0; k < nd; ++k) K_(w, k) = K_(w, k) * rf[k];
  });
  return fv3_post(c, (fv3_stream_t)stream, "ray_fast");
}

// the three cells at every cube corner of a sub-domain become their mean (del2_cubed, before each iteration), in place
void del2_fill_corners(fv3_ctx *c, fv3_stream_t s, Real *qin) {
  const Geo g = c->g;
  const int nz1 = g.nz - 1;
  launch3(c, s, Box{1, 1, 1, 1, 0, nz1}, [=] FV3_HD(int t, int k, int, int) {
    const int fl = g.flags[t];
    const bool W = fl & FV3_W, E = fl & FV3_E, S = fl & FV3_S, N = fl & FV3_N;
    Real *qq = qin + t * g.st + k * g.sk;
    const int npx = g.npx, npy = g.npy, ie = g.nx, je = g.ny;
    const Real r3 = (Real)(1.0 / 3.0);
    if (W && S) {
      const Real a = (qq[IX(1, 1)] + qq[IX(0, 1)] + qq[IX(1, 0)]) * r3;
      qq[IX(1, 1)] = a; qq[IX(0, 1)] = a; qq[IX(1, 0)] = a;
    }
    if (E && S) {
      const Real a = (qq[IX(ie, 1)] + qq[IX(npx, 1)] + qq[IX(ie, 0)]) * r3;
      qq[IX(ie, 1)] = a; qq[IX(npx, 1)] = a; qq[IX(ie, 0)] = a;
    }
    if (E && N) {
      const Real a = (qq[IX(ie, je)] + qq[IX(npx, je)] + qq[IX(ie, npy)]) * r3;
      qq[IX(ie, je)] = a; qq[IX(npx, je)] = a; qq[IX(ie, npy)] = a;
    }
    if (W && N) {
      const Real a = (qq[IX(1, je)] + qq[IX(0, je)] + qq[IX(1, npy)]) * r3;
      qq[IX(1, je)] = a; qq[IX(0, je)] = a; qq[IX(1, npy)] = a;
    }
  });
}

// ---------------------------------------------------------------------------------------------
extern "C" int fv3_del2_cubed(fv3_ctx *c, const fv3_field *q_, double cdd, int nmax, void *stream) {
  if (!c) return FV3_ERR_ARG;
  FV3_FIELD(q, q_)
  const Geo g = c->g;
  fv3_stream_t s = (fv3_stream_t)stream;
  const Real cd = (Real)cdd;
  const int ntimes = nmax < 3 ? nmax : 3;
  const int nz1 = g.nz - 1;
  // One launch per iteration, out of place (q -> B -> A -> q for three iterations): the x / y fluxes of a cell's four
  // faces are formed in registers from the five neighbouring values instead of being stored and read back (7 -> 2
  // field passes per iteration), FV3_KC levels per thread with the five metric terms loaded once.  Every launch
  // covers the whole padded plane -- cells outside the iteration's update box are copied -- so each buffer is
  // complete, as the in-place reference field is.
  Real *bufA = c->scratch[SC_A], *bufB = c->scratch[SC_B];
  const int isd = 1 - g.nh, ied = g.nx + g.nh, jsd = 1 - g.nh, jed = g.ny + g.nh;
  const int nkc = (nz1 + FV3_KC) / FV3_KC;
  Real *in = q;
  for (int n = 1; n <= ntimes; ++n) {
    const int nt = ntimes - n;
    Real *out = nt == 0 ? (ntimes == 1 ? bufA : q) : (nt % 2 ? bufA : bufB);
    if (out == in) out = out == bufA ? bufB : bufA;
    Real *qin = in;
    del2_fill_corners(c, s, qin);
    launch3(c, s, Box{isd, ied, jsd, jed, 0, nkc - 1}, [=] FV3_HD(int t, int kp, int i_, int j_) {
      const int fl = g.flags[t];
      const long m2 = t * g.st2;
      int i = i_, j = j_;
      const bool upd = i >= 1 - nt && i <= g.nx + nt && j >= 1 - nt && j <= g.ny + nt;
      const unsigned p0 = IX(i, j);
      Real mvx0 = (Real)0, mvx1 = (Real)0, muy0 = (Real)0, muy1 = (Real)0, cra = (Real)0;
      // the six points of the stencil as in-plane offsets, formed ONCE (they do not depend on the level): the corner-halo remaps
      // (cc_index: a dozen compares and selects per point) were evaluated per level inside the loop -- the kernel ran at 2.8 TB/s on its
      // 5.5 GB because of that index arithmetic, not because of its bytes
      unsigned pxw = p0, pxc = p0, pxe = p0, pys = p0, pyc = p0, pyn = p0;
      if (upd) {
        mvx0 = (g.del6_v + m2)[p0];
        mvx1 = (g.del6_v + m2)[IX(i + 1, j)];
        muy0 = (g.del6_u + m2)[p0];
        muy1 = (g.del6_u + m2)[IX(i, j + 1)];
        cra = cd * (g.rarea + m2)[p0];
        if (nt > 0) {
          pxw = cc_index<1>(g, fl, i - 1, j);
          pxc = cc_index<1>(g, fl, i, j);
          pxe = cc_index<1>(g, fl, i + 1, j);
          pys = cc_index<2>(g, fl, i, j - 1);
          pyc = cc_index<2>(g, fl, i, j);
          pyn = cc_index<2>(g, fl, i, j + 1);
        } else {
          pxw = IX(i - 1, j);
          pxe = IX(i + 1, j);
          pys = IX(i, j - 1);
          pyn = IX(i, j + 1);
        }
      }
#ifndef FV3_DEL2_UNROLL
#define FV3_DEL2_UNROLL 2
#endif
      // (no early exit inside the loop: a level past the last one is clamped and its store masked, so that the unrolled body has the
      //  loads of several levels in flight -- with `break` every level waited for its own seven loads)
#pragma unroll FV3_DEL2_UNROLL
      for (int kk = 0; kk < FV3_KC; ++kk) {
        const int k_ = FV3_KC * kp + kk;
        const bool live = k_ <= nz1;
        const int k = live ? k_ : nz1;
        const long b = t * g.st + k * g.sk;
        const Real *qq = qin + b;
        Real v = qq[p0];
        if (upd) {
          const Real xw = qq[pxw], xc = qq[pxc], xe = qq[pxe], ys = qq[pys], yc = qq[pyc], yn = qq[pyn];
          v = v + cra * (mvx0 * (xw - xc) - mvx1 * (xc - xe) + muy0 * (ys - yc) - muy1 * (yc - yn));
        }
        if (live) (out + b)[p0] = v;
      }
    });
    in = out;
  }
  if (in != q) {
    Real *src = in;
    launch3<4>(c, s, Box{isd, ied, jsd, jed, 0, nz1}, [=] FV3_HD(int t, int k, int i, int j) {
      const long p = t * g.st + k * g.sk + IX(i, j);
      q[p] = src[p];
    });
  }
  return fv3_post(c, s, "del2_cubed");
}

extern "C" int fv3_apply_diffusive_heating(fv3_ctx *c, const fv3_field *delp_, const fv3_field *delz_, const fv3_field *cappa_, const fv3_field *hs_,
                                           const fv3_field *pt_, double delt, void *stream) {
  if (!c) return FV3_ERR_ARG;
  FV3_FIELD(delp, delp_) FV3_FIELD(delz, delz_) FV3_FIELD(cappa, cappa_) FV3_FIELD(hs, hs_) FV3_FIELD(pt, pt_)
  const Geo g = c->g;
  const Real rdg = (Real)(-c->cst.rdgas / c->cst.grav), cv_air = (Real)(c->cst.cp_air - c->cst.rdgas), lim0 = (Real)delt;
  launch3(c, (fv3_stream_t)stream, Box{1, g.nx, 1, g.ny, 0, g.nz - 1}, [=] FV3_HD(int t, int k, int i, int j) {
    const long p = t * g.st + k * g.sk + IX(i, j);
    const Real cp = cappa[p];
    const Real pkz = fv3_exp(cp / ((Real)1.0 - cp) * fv3_log(rdg * delp[p] / delz[p] * pt[p]));
    const Real dtmp = hs[p] / (cv_air * delp[p]);
    Real lim = lim0;
    if (k == 0) lim = lim * (Real)0.1;
    if (k == 1) lim = lim * (Real)0.5;
    const Real mag = fv3_min(lim, fabs(dtmp));
    const Real sg = dtmp > (Real)0 ? (Real)1 : (dtmp < (Real)0 ? (Real)-1 : (Real)0);
    pt[p] = pt[p] + sg * mag / pkz;
  });
  return fv3_post(c, (fv3_stream_t)stream, "apply_diffusive_heating");
}

// ---------------------------------------------------------------------------------------------
// Self-test of the hand-written arithmetic / register tables of this file ON THE DEVICE (tests/test_device_math.py): the claims "fv3_div
// is bit for bit `/`", "log / exp are the host emulation's" and "every slot of the accumulation-register column holds what was put
// there" are checked directly, not only through the solvers.  x, y, out: device pointers to n doubles.
//   which 0: out = fv3_div(x, y)    1: out = fv3_log(x)    2: out = fv3_exp(x)
//   which 3: n = 80 * 64: one wave puts x[k * 64 + lane] into slot k (k = 0 .. 79; slots 4g .. 4g+3 through the group table when
//            g is odd, one by one when even), then reads the slots back in descending order into out[k * 64 + lane]
// ---------------------------------------------------------------------------------------------
#ifndef FV3_HOST_EMU
__global__ void __launch_bounds__(256) fv3_selftest_math_kernel(int which, const double *x, const double *y, double *out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = which == 0 ? fv3_div(x[i], y[i]) : which == 1 ? fv3_log(x[i]) : fv3_exp(x[i]);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) fv3_selftest_agpr_kernel(const double *x, double *out) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass parses kernel bodies too; the register tables exist in the device pass only)
  const int lane = threadIdx.x;
  for (int g = 0; g < FV3_AGPR_LEVELS / 4; ++g) {
    const double v0 = x[(4 * g + 0) * 64 + lane], v1 = x[(4 * g + 1) * 64 + lane], v2 = x[(4 * g + 2) * 64 + lane], v3 = x[(4 * g + 3) * 64 + lane];
    if (g & 1) {
      fv3_agpr_set4(g, v0, v1, v2, v3);
    } else {
      fv3_agpr_set(4 * g + 0, v0);
      fv3_agpr_set(4 * g + 1, v1);
      fv3_agpr_set(4 * g + 2, v2);
      fv3_agpr_set(4 * g + 3, v3);
    }
  }
  for (int k = FV3_AGPR_LEVELS - 1; k >= 0; --k) out[k * 64 + lane] = fv3_agpr_get(k);
#else
  (void)x;
  (void)out;
#endif
}
#endif
extern "C" int fv3_selftest_math(fv3_ctx *c, int which, const double *x, const double *y, double *out, int64_t n, void *stream) {
  if (!c || !x || !out || n < 0 || which < 0 || which > 3 || (which == 0 && !y)) return FV3_ERR_ARG;
  if (which == 3 && n != (int64_t)FV3_AGPR_LEVELS * 64) return fv3_fail(c, FV3_ERR_ARG, "fv3_selftest_math(3): n must be 80 * 64");
  if (n == 0) return FV3_OK;
#ifdef FV3_HOST_EMU
  (void)stream;
  for (int64_t i = 0; i < n; ++i) out[i] = which == 0 ? fv3_div(x[i], y[i]) : which == 1 ? fv3_log(x[i]) : which == 2 ? fv3_exp(x[i]) : x[i];
  return FV3_OK;
#else
  if (which == 3)
    hipLaunchKernelGGL(fv3_selftest_agpr_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, x, out);
  else
    hipLaunchKernelGGL(fv3_selftest_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, which, x, y, out, n);
  return fv3_post(c, (fv3_stream_t)stream, "selftest_math");
#endif
}

#ifdef FV3_HOST_EMU
// test hook of the host-emulation library only (tests/test_fast_math.py): the solvers' log on an array
extern "C" void fv3_hostemu_log(const double *x, double *y, long n) {
  for (long i = 0; i < n; ++i) y[i] = fv3_log(x[i]);
}
extern "C" void fv3_hostemu_exp(const double *x, double *y, long n) {
  for (long i = 0; i < n; ++i) y[i] = fv3_exp(x[i]);
}
#endif
