// fv3_ops.h -- internal operator building blocks shared between translation units.
#pragma once
#include "fv3_common.h"

// scratch slot map (indices into ctx->scratch); operators on one stream never overlap in time.
enum {
  SC_TP_FY2 = 0,  // fv_tp_2d inner y flux
  SC_TP_FX2 = 1,  // fv_tp_2d inner x flux
  SC_TP_QI = 2,
  SC_TP_QJ = 3,
  SC_DN_D2 = 4,   // del-n work array
  SC_DN_FX = 5,   // del-n x flux
  SC_DN_FY = 6,   // del-n y flux
  SC_A = 7,       // operator-level temporaries (d_sw, c_sw, nh, a2b ...)
  SC_B = 8,
  SC_C = 9,
  SC_D = 10,
  SC_E = 11,
  SC_F = 12,
  SC_G = 13,
  SC_H = 14,
  SC_I = 15,
  SC_J = 16,
  SC_K = 17,
  SC_L = 18,  // del-n corner windows (private fluxes)
  SC_M = 19,
  SC_N = 20,  // d_sw flux-form updates formed by the transport epilogues (delp_new, delp*w, delp*q_con, delp*pt)
  SC_O = 21,
  SC_P = 22,
  SC_Q = 23,
  SC_R = 24,  // d_sw absolute vorticity
  SC_S = 25,  // second pair of del-n damping fluxes (d_sw pipelines the chains one transport ahead)
  SC_T = 26,
  SC_U = 27,  // d_sw: y damping flux of w (kept until the post-transport kernel forms dw and its heat)
  SC_COUNT = 28
};

// Per-level del-n control.  Level k uses order nord_k[k] (or nord_u), coefficient damp_k[k]
// (or damp_u) and is active when on_k[k] > on_thr (or on_u).
struct Deln {
  const int *nord_k;
  const Real *damp_k;
  const Real *on_k;
  int nord_u;
  Real damp_u;
  bool on_u;
  Real on_thr;
  int nord_max;  // host side: max order over the levels of the call
};
FV3_HD inline int deln_nord(const Deln &d, int k) { return d.nord_k ? d.nord_k[k] : d.nord_u; }
FV3_HD inline Real deln_damp(const Deln &d, int k) { return d.damp_k ? d.damp_k[k] : d.damp_u; }
FV3_HD inline bool deln_on(const Deln &d, int k) { return d.on_k ? d.on_k[k] > d.on_thr : d.on_u; }

#define FV3_D6_PATCH 8  // the faces within this distance of a cube corner come from the staged del-n chain (corner patches)

// Optional epilogue of fv_tp_2d: the flux-form update the callers apply right after the transport,
//   out = (mult ? mult * q : q) + (fx - fx[i+1] + fy - fy[j+1]) * rarea       on the compute cells,
// formed in the transport kernel itself (the fluxes of the neighbouring faces are already in the
// wave) instead of a second pass over fx / fy.  write_flux = false: fx / fy are not stored at all.
// acc_x / acc_y (optional): acc += flux on the owned faces (d_sw's mfx / mfy accumulation).
// wind_u / wind_v / wind_ke (optional): d_sw's wind update from the vorticity fluxes, in place,
//   u = u * dx + ke - ke[i+1] + fy   (i in 1..nx, j in 1..ny+1),   v = v * dy + ke - ke[j+1] - fx   (i in 1..nx+1, j in 1..ny).
struct TpEpi {
  Real *out;
  const Real *mult;
  bool write_flux;
  Real *acc_x, *acc_y;
  Real *wind_u, *wind_v;
  const Real *wind_ke;
  // area_form: out = (q * area + fx - fx[i+1] + fy - fy[j+1]) / (ra_x + ra_y - area) with ra_x = area + xfx - xfx[i+1],
  // ra_y = area + yfx - yfx[j+1]  (update_dz_d's advective-form height update) instead of the flux form above
  bool area_form;
  // area form only (optional): out += (zfx - zfx[i+1] + zfy - zfy[j+1]) * rarea on the levels with zon[k] > 1e-5
  // (update_dz_d's del-n damping of the interface heights; zfx / zfy are del6_vt_flux outputs)
  const Real *zfx, *zfy, *zon;
  // damping fluxes of q already computed by the caller (del6_vt_flux into these arrays): fv_tp_2d then does
  // not run the del-n chain itself (d_sw computes them one transport ahead on the auxiliary stream)
  const Real *damp_fx, *damp_fy;
  // wind epilogue only (optional): d_sw's vorticity damping applied in the same store,
  //   u = (u * dx + ke - ke[i+1] + fy) + wind_du,   v = (v * dy + ke - ke[j+1] - fx) - wind_dv     on the levels with wind_don[k] > 1e-5,
  // while the pre-damping values (what the damping-heat kernel differentiates) go to wind_u_pre / wind_v_pre
  const Real *wind_du, *wind_dv, *wind_don;
  Real *wind_u_pre, *wind_v_pre;
  // area form with zfx / zfy (update_dz_d): fd != 0 = the del-n chain of q (order 2 on every level of the call, d2 of iteration 0 =
  // fd_coef[k] * q) runs INSIDE the march on the strips away from the W / E tile edges; zfx / zfy then only hold the tile-edge
  // strips and the cube-corner patches (del6_vt_flux_edge_strips).  See dsw_scalars_t (fv3_tp4.hip) for the pipeline.
  // Wind form with wind_du / wind_dv: the same for the relative vorticity q, whose chain gives the damping increments (stored
  // into wind_du / wind_dv for the damping-heat kernel); fd_add (2-D, optional) is added to q where the transport loads it
  // (absolute vorticity = q + f0 without a field of its own).
  int fd = 0;
  const Real *fd_coef = nullptr;
  const Real *fd_add = nullptr;
  // wind form, fd != 0, round-5 march only: the damping heat formed in the march (TpHeat, below); ignored by the other forms
  const struct TpHeat *heat = nullptr;
};

// fv_tp_2d on levels k0..k1.  mfx/mfy/mass may be null; dn may be null (no damping).
void tp2d(fv3_ctx *c, fv3_stream_t s, const Real *q, const Real *crx, const Real *cry, const Real *xfx, const Real *yfx, Real *fx, Real *fy,
          const Real *mfx, const Real *mfy, const Real *mass, int hord, const Deln *dn, int k0, int k1, const TpEpi *epi = nullptr);

// d_sw's four scalar transports (delp, w, q_con, pt) and the division by the new air mass as ONE march (fv3_tp2d.hip):
// the Courant numbers / area fluxes / cell areas are read once for the four tracers, the air-mass fluxes the three
// mass-weighted transports ride on never leave the wave (they are only accumulated into mfx / mfy), and the
// flux-form updates, the divisions and w's damping increment + dissipated heat are the epilogue of the cell.
// Inputs are read-only; the new fields go to o_* (out of place: neighbouring waves still read the old halo values).
struct DswScalars {
  const Real *delp, *w, *q_con, *pt;
  Real *o_delp, *o_w, *o_q_con, *o_pt, *heat;
  const Real *crx, *cry, *xfx, *yfx;
  Real *mfx, *mfy;                  // accumulated: += air-mass flux
  Real *fx, *fy;                    // air-mass fluxes (work fields of the two-launch form)
  const Real *dpx, *dpy;            // del-n damping fluxes of delp (plain), q_con and pt (mass-weighted), w (applied as an increment)
  const Real *dqx, *dqy, *dtx, *dty, *dwx, *dwy;
  int hord_dp, hord_vt, hord_tm;
  Deln dn_vt, dn_t;
  Real dt;
  Deln dn_w;  // w's chain (fused form: the march runs it; the other forms only test its switch through g.damp_w)
  int fd_k0;  // levels >= fd_k0 run the del-n chains inside the marches (every chain of order 2 there); nz = never
  // first sub-step of an acoustic call inside the sequencer (fv3_ctx::seq_acc_first): mfx / mfy hold nothing yet -- the marches add the flux to a zero
  // read from `zeros` (one level plane, read with the field's own in-plane offsets; it stays in L2) instead of reading the fields
  bool acc_first = false;
  const Real *zeros = nullptr;
};
// mode 0: one march for the four tracers (one wave per SIMD); 1: delp + w, then q_con + pt on the stored air-mass fluxes
void dsw_scalars_stream(fv3_ctx *c, fv3_stream_t s, const DswScalars &a, int mode);
bool dsw_honors_acc_first(const fv3_ctx *c);  // (fv3_dsw.hip)
bool dsw_can_defer_acc(const fv3_ctx *c);     // (fv3_dsw.hip)
void del2_fill_corners(fv3_ctx *c, fv3_stream_t s, Real *qin);  // (fv3_nh.hip)
// fv3_del2x.hip: del2_cubed (three iterations) + apply_diffusive_heating as one pass; 1 = no fused form for this configuration
int fv3_del2_heat_fused(fv3_ctx *c, const fv3_field *q, double cd, int nmax, const fv3_field *delp, const fv3_field *delz, const fv3_field *cappa,
                        const fv3_field *pt, double delt, bool keep_q, void *stream);

bool tp2d_fd_lean(const fv3_ctx *c, int hord, int k0, int k1);  // (fv3_tp2d.hip) will tp2d's FD forms run the round-5 march?  (then only the corner patches of the chain's fluxes are needed)
// kind 1: d_sw's vorticity transport + wind update, 2: update_dz_d's interface-height transport (TpEpi as for tp2d with fd = 1); PPM order 6
struct TpEpi;
// d_sw's damping heat as the epilogue of the vorticity march (kind 1): heat_src += ndelp * (heat_s - 0.25 d_con rsin2 (...)) on the levels with d_con > 1e-5,
// heat_src += heat_s on the others; vdamp = the corner damping field.  Null: the winds before the damping and the increments are stored for the
// damping-heat kernel instead (TpEpi::wind_u_pre ...).
struct TpHeat {
  const Real *vdamp, *ndelp, *heat_s, *dcon;
  Real *heat_src;
  const Real *zeros = nullptr;  // non-null: first sub-step of a call inside the sequencer -- the accumulated heat is read as zero from this block, not from heat_src
  // set by the dispatch that actually launches the HEAT march (tp2d_single_march): the caller derives the level range its damping-heat kernel still has to serve
  // from THIS, not from a prediction of the dispatch (the two predicates -- d_sw's and tp2d_stream's -- could drift apart silently otherwise)
  mutable bool consumed = false;
  int side_from = 1 << 30;  // levels >= side_from already have their side copies (the fused wind stage stored them): sx_side_copy serves the levels under it
};
void tp2d_single_march(fv3_ctx *c, fv3_stream_t s, int kind, const Real *q, const Real *crx, const Real *cry, const Real *xfx, const Real *yfx, int k0, int k1,
                       const TpEpi *epi, const TpHeat *heat = nullptr);
// the round-5 two-tracer march on the tiles without a cube corner (fv3_tp4x.hip); role 1 = delp + w, 2 = q_con + pt; levels k_lo .. k_hi must
// all run their del-n chains inside the march (>= fd_k0), PPM order 6
void dsw_pair_march(fv3_ctx *c, fv3_stream_t s, const DswScalars &a, int role, int k_lo, int k_hi);

// d_sw's wind-branch stage kernels as one march (fv3_wind.hip): reads u, v, uc, vc, divgd (the un-iterated corner divergence); writes wk (cell-mean relative
// vorticity, every cell of the padded plane), ke (corner kinetic energy + damping) and vdamp (the corner damping field) on the corners that take the interior
// formulas (>= 4 from a cube-tile edge), and exports the iterated divergence to dnew on the other corners of the compute domain (the per-point launch of the
// caller finishes those; store_dn: on every corner).  dnew must already hold the staged chain's values on the WS_PATCH^2 corners next to a cube corner.
// Levels k0 .. k1 must all run the damping chain (0 < nord <= 3).
#define WS_PATCH 8
struct WindStage {
  const Real *u, *v, *uc, *vc, *divgd;
  Real *ke, *vdamp, *wk, *dnew;
  const Real *dd8;  // per-level (da_min_c * d4_bg)^(nord+1)
  Real dt, dddmp;
  int hord;
  bool store_dn;
  int k0, k1;
  // side copies for the damping-heat epilogue of the vorticity march (fv3_tp2x.hip: sx_side_copy): u on the first row of every row segment of THAT march but
  // the first (rows 1 + m * side_seg) and v on the first column of every strip but the first (columns 1 + b * 58), before the march updates the winds in place.
  // This march holds both rows anyway: side_seg > 0 makes it store them (null / 0: the copies stay a launch of their own).
  Real *u_side = nullptr, *v_side = nullptr;
  int side_seg = 0;
};
void wind_stage_march(fv3_ctx *c, fv3_stream_t s, const WindStage &a);
int sx_march_seg(const fv3_ctx *c, int nk);  // (fv3_tp2x.hip) rows per segment of the single-tracer marches on nk levels

// two tracers riding on given mass fluxes (the TRC march alone: tracer_2d_1l): a.q_con / a.pt = the two tracers, a.delp = the
// old air mass, a.o_delp = the new one, a.fx / a.fy = the mass fluxes, outputs a.o_q_con / a.o_pt; no damping (dn_* off)
void tracer_pair_stream(fv3_ctx *c, fv3_stream_t s, const DswScalars &a);

// del6_vt_flux: fx2, fy2 (work d2) of q.  q_raw: d2 starts as q instead of damp*q.
// del6_vt_flux_patches: only the faces on the FV3_D6_PATCH^2 patches at the cube corners (staged chain; orders > 0) -- for callers that
// run the chain themselves everywhere else (the fused scalar marches of d_sw)
#ifndef FV3_Q4_KB_DEFAULT
#define FV3_Q4_KB_DEFAULT 16  // levels of one tile an XCD walks back to back in the transport marches (0: plane-major launches)
#endif
// del6_vt_flux_edge_strips: the strips that touch a W / E cube-tile edge (+ the corner patches) -- for tp2d's fused form (TpEpi::fd)
void del6_vt_flux_edge_strips(fv3_ctx *c, fv3_stream_t s, const Real *q, Real *d2, Real *fx2, Real *fy2, const Deln &dn, bool q_raw, int k0, int k1);
void del6_vt_flux_patches(fv3_ctx *c, fv3_stream_t s, const Real *q, Real *d2, Real *fx2, Real *fy2, const Deln &dn, bool q_raw, int k0, int k1);
void del6_vt_flux(fv3_ctx *c, fv3_stream_t s, const Real *q, Real *d2, Real *fx2, Real *fy2, const Deln &dn, bool q_raw, int k0, int k1);

// a2b_ord4: qout levels kout0.. from qin levels kin0.. (nk levels); replace writes back into qin
void a2b_ord4(fv3_ctx *c, fv3_stream_t s, Real *qin, Real *qout, int kin0, int kout0, int nk, bool replace, Real scale = (Real)1);
// fv3_d_sw with the new delp / pt / w / q_con written to o_* instead of in place (all four, or all null = fv3_d_sw)
int fv3_d_sw_out(fv3_ctx *c, const fv3_field *delpc, const fv3_field *delp, const fv3_field *pt, const fv3_field *u, const fv3_field *v, const fv3_field *w,
                 const fv3_field *uc, const fv3_field *vc, const fv3_field *ua, const fv3_field *va, const fv3_field *divgd, const fv3_field *mfx,
                 const fv3_field *mfy, const fv3_field *cx, const fv3_field *cy, const fv3_field *crx, const fv3_field *cry, const fv3_field *xfx,
                 const fv3_field *yfx, const fv3_field *q_con, const fv3_field *zh, const fv3_field *heat_source, const fv3_field *diss_est, double dt,
                 void *stream, const fv3_field *o_delp, const fv3_field *o_pt, const fv3_field *o_w, const fv3_field *o_q_con, int (*after_scalars)(void *) = nullptr,
                 void *after_user = nullptr);  // after_scalars: called once the four new scalars are final (the winds still to come)
int fv3_csw_join(fv3_ctx *c, void *stream);  // (fv3_csw.hip) the join of c_sw's deferred stage D / E windows (fv3_ctx::seq_csw_defer)
// nh_p_grad as one marching kernel (fv3_pgf.hip); pass = the sequencer's frame-first pass (0 all, 1 sub-domain frames, 2 the rest)
void nh_pgf_fused(fv3_ctx *c, fv3_stream_t s, const Real *pp, const Real *pk3, const Real *gz, const Real *delp, Real *u, Real *v, Real dt, Real top, Real gz_scale,
                  int pass);
int fv3_nh_p_grad_scaled(fv3_ctx *c, const fv3_field *u, const fv3_field *v, const fv3_field *pp, const fv3_field *gz, const fv3_field *pk3, const fv3_field *delp,
                         double dt, double ptop, double akap, double gz_scale, void *stream);
int fv3_update_dz_c_from(fv3_ctx *c, const fv3_field *zs, const fv3_field *ut, const fv3_field *vt, const fv3_field *gz_in, const fv3_field *gz, const fv3_field *ws,
                         double dt, void *stream);

