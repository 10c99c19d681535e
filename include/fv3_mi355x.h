/*
 * fv3_mi355x.h -- C ABI of the MI355X-native FV3 acoustic dynamics.
 *
 * Drop-in boundary for the hot path pace reaches through
 *   Driver._critical_path_step_all -> DynamicalCore.step_dynamics -> AcousticDynamics
 *   [REF driver/pace/driver/driver.py:494-504, 641]
 * One entry point per operator object the reference constructs through
 * StencilFactory / QuantityFactory [REF driver/pace/driver/driver.py:744-765;
 * examples/notebooks/functions.py:877-891, 935-951].  pyFV3 itself is an
 * un-vendored submodule, so each entry point cites the reference evidence for
 * the operator it replaces (class name / checkpoint variables / config field).
 *
 * Conventions
 *  - plain pointers and sizes only; no torch / numpy types cross this ABI.
 *  - every 3-D field of a context shares ONE layout: [n_sub][nk_alloc][nj_alloc][ni_alloc],
 *    i fastest (stride 1).  Logical index order is the reference's (i, j, k)
 *    [REF tests/main/fv3core/test_init_from_geos.py:94-113]; fields are padded to the
 *    interface shape (nx+2h+1, ny+2h+1, nz+1) like NDSL storages.
 *  - n_sub sub-domains (ranks of the reference) are co-resident in one context and
 *    processed by the same launches ("6 tiles on 1 GPU", BASELINE cfg-1).
 *  - all calls are asynchronous on the given hipStream_t (passed as void*); the
 *    library never allocates at call time (scratch is owned by the context), mirroring
 *    the reference invariants [REF tests/main/fv3core/test_dycore_call.py:193-211].
 *  - return 0 on success, negative fv3_status otherwise; fv3_last_error() has the text.
 *    Nothing throws across the boundary.
 */
#ifndef FV3_MI355X_H
#define FV3_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FV3_MAX_SUB 32
#define FV3_ABI_VERSION 2

typedef enum {
  FV3_OK = 0,
  FV3_ERR_ARG = -1,         /* bad pointer / shape / stride / dtype ("validate_args") */
  FV3_ERR_HIP = -2,         /* a HIP runtime call failed */
  FV3_ERR_UNSUPPORTED = -3, /* configuration outside the specialised set */
  FV3_ERR_NOMEM = -4
} fv3_status;

typedef enum { FV3_F64 = 0, FV3_F32 = 1 } fv3_dtype;

/* edge flag bits per sub-domain == GridIndexing.{west,east,south,north}_edge
 * [REF tests/main/fv3core/test_grid.py:56-101] */
#define FV3_EDGE_W 1
#define FV3_EDGE_E 2
#define FV3_EDGE_S 4
#define FV3_EDGE_N 8

typedef struct fv3_ctx fv3_ctx;

/* A borrowed view of one Quantity's storage (all sub-domains). */
typedef struct {
  void *ptr;
  int64_t shape[3];   /* logical (i, j, k) extents of the allocation */
  int64_t stride[3];  /* element strides for (i, j, k); stride[0] must be 1 */
  int64_t sub_stride; /* elements between consecutive sub-domains */
  int32_t n_sub;
  int32_t dtype;      /* fv3_dtype */
} fv3_field;

/* GridIndexing / sizer [REF driver/pace/driver/driver.py:744-765] */
typedef struct {
  int32_t nx, ny, nz, n_halo, n_sub;
  int32_t edge_flags[FV3_MAX_SUB];
} fv3_gridspec;

/* GridData + DampingCoefficients, device pointers, 2-D fields laid out
 * [n_sub][nj_alloc][ni_alloc]  [REF tests/mpi_54rank/test_grid_init.py:33-120] */
typedef struct {
  const void *dx, *dy, *dxa, *dya, *dxc, *dyc;
  const void *rdx, *rdy, *rdxa, *rdya, *rdxc, *rdyc;
  const void *area, *rarea, *area_c, *rarea_c;
  const void *cosa, *sina, *rsina, *cosa_u, *cosa_v, *cosa_s;
  const void *sina_u, *sina_v, *rsin_u, *rsin_v, *rsin2;
  const void *sin_sg1, *sin_sg2, *sin_sg3, *sin_sg4;
  const void *cos_sg1, *cos_sg2, *cos_sg3, *cos_sg4;
  const void *fC, *f0;
  const void *del6_u, *del6_v, *divg_u, *divg_v;
  const void *edge_w, *edge_e; /* [n_sub][nj_alloc] */
  const void *edge_s, *edge_n; /* [n_sub][ni_alloc] */
  /* host arrays (copied at create) */
  const double *corner_extrap; /* [n_sub][4][3] a2b_ord4 cube-corner factors */
  const double *ak, *bk;       /* [nz+1] */
  double da_min, da_min_c;
  const void *sin_sg5; /* (ABI 2) sine of the grid angle at the cell centre: tracer_2d_1l's Courant bound; may be NULL if unused */
} fv3_griddata;

/* dycore_config fields the acoustic path reads
 * [REF driver/examples/configs/baroclinic_c12.yaml:43-93] */
typedef struct {
  int32_t n_split, k_split;
  int32_t hord_dp, hord_mt, hord_tm, hord_vt;
  int32_t nord, n_sponge;
  int32_t do_vort_damp, rf_fast, hydrostatic, use_logp, grid_type;
  double a_imp, beta, p_fac;
  double d2_bg, d2_bg_k1, d2_bg_k2, d4_bg, dddmp, d_con, d_ext, delt_max, ke_bg, vtdm4;
  double rf_cutoff, tau;
} fv3_acoustic_config;

/* PACE_CONSTANTS set [REF README.md:88-93] */
typedef struct {
  double radius, omega, grav, rdgas, rvgas, cp_air, dz_min, pi, seconds_per_day;
} fv3_constants;

/* DycoreState fields on the path [REF tests/main/fv3core/test_init_from_geos.py:128-199;
 * driver/pace/driver/state.py:131-139] */
typedef struct {
  fv3_field u, v, w, ua, va, uc, vc, delp, delz, pt, pe, pk, peln, pkz, q_con, omga, cappa;
  fv3_field mfxd, mfyd, cxd, cyd, diss_estd;
  fv3_field phis; /* 2-D: shape[2] == 1 */
} fv3_state;

int fv3_version(void);
const char *fv3_last_error(const fv3_ctx *ctx); /* ctx may be NULL: last create error */
const char *fv3_backend(void);                  /* "hip:gfx950" (product) or "hostemu" (test build) */
/* sha256 (hex, first 16 digits) of the kernel sources + headers this library was built from (pace_amd/build.py: src_hash): the counter
 * files bench.py quotes (profiles/traffic_d_sw.json, valu_d_sw.json) carry the hash of the tree they were measured on, and a line built from
 * another tree reports them as null instead of a stale number */
const char *fv3_build_id(void);

int fv3_ctx_create(fv3_ctx **out, const fv3_gridspec *spec, const fv3_griddata *grid,
                   const fv3_acoustic_config *cfg, const fv3_constants *consts, int device, int dtype);
int fv3_ctx_destroy(fv3_ctx *ctx);
/* bytes of device scratch the context owns (for memory accounting) */
int64_t fv3_ctx_scratch_bytes(const fv3_ctx *ctx);
/* device_sync: true semantics [REF .jenkins/driver_configs/baroclinic_c192_6ranks.yaml:7] */
int fv3_ctx_set_device_sync(fv3_ctx *ctx, int on);

/* ---- operators (argument order = the reference operator's __call__) ---- */

/* CGridShallowWaterDynamics.__call__ [REF tests/savepoint/thresholds/fv_dynamics.yaml:2-75].
 * delpc / ptc are outputs (the reference returns them). */
int fv3_c_sw(fv3_ctx *, const fv3_field *delp, const fv3_field *pt, const fv3_field *u, const fv3_field *v,
             const fv3_field *w, const fv3_field *uc, const fv3_field *vc, const fv3_field *ua,
             const fv3_field *va, const fv3_field *ut, const fv3_field *vt, const fv3_field *divgd,
             const fv3_field *omga, const fv3_field *delpc, const fv3_field *ptc, double dt2, void *stream);

/* UpdateGeopotentialHeightOnCGrid.__call__(dp_ref, zs, ut, vt, gz, ws, dt) (dp_ref lives in the ctx) */
int fv3_update_dz_c(fv3_ctx *, const fv3_field *zs, const fv3_field *ut, const fv3_field *vt,
                    const fv3_field *gz, const fv3_field *ws, double dt, void *stream);

/* RiemannSolverC.__call__(dt2, cappa, ptop, phis, ws, ptc, q_con, delpc, gz, pef, w3)
 * [REF tests/main/fv3core/test_config.py:13] */
int fv3_riem_solver_c(fv3_ctx *, double dt2, const fv3_field *cappa, double ptop, const fv3_field *phis,
                      const fv3_field *ws, const fv3_field *ptc, const fv3_field *q_con,
                      const fv3_field *delpc, const fv3_field *gz, const fv3_field *pef,
                      const fv3_field *w3, void *stream);

/* p_grad_c(rdxc, rdyc, uc, vc, delpc, pkc, gz, dt2) (rdxc/rdyc live in the ctx) */
int fv3_p_grad_c(fv3_ctx *, const fv3_field *uc, const fv3_field *vc, const fv3_field *delpc,
                 const fv3_field *pkc, const fv3_field *gz, double dt2, void *stream);

/* FiniteVolumeFluxPrep.__call__(uc, vc, crx, cry, xfx, yfx, ut, vt, dt)
 * [REF examples/notebooks/functions.py:877-891] */
int fv3_fxadv(fv3_ctx *, const fv3_field *uc, const fv3_field *vc, const fv3_field *crx, const fv3_field *cry,
              const fv3_field *xfx, const fv3_field *yfx, const fv3_field *ut, const fv3_field *vt, double dt,
              void *stream);

/* FiniteVolumeTransport.__call__(q, crx, cry, xfx, yfx, fx, fy[, mfx, mfy, mass])
 * [REF examples/notebooks/functions.py:935-951].  mfx/mfy/mass may be NULL.
 * nord < 0 or damp_c <= 1e-4 disables the del-n damping fluxes. */
int fv3_fv_tp_2d(fv3_ctx *, const fv3_field *q, const fv3_field *crx, const fv3_field *cry,
                 const fv3_field *xfx, const fv3_field *yfx, const fv3_field *fx, const fv3_field *fy,
                 const fv3_field *mfx, const fv3_field *mfy, const fv3_field *mass, int hord, int nord,
                 double damp_c, void *stream);

/* AGrid2BGridFourthOrder.__call__(qin, qout) ; replace != 0 writes qout back into qin */
int fv3_a2b_ord4(fv3_ctx *, const fv3_field *qin, const fv3_field *qout, int kstart, int nk, int replace,
                 void *stream);

/* DGridShallowWaterLagrangianDynamics.__call__ [REF tests/savepoint/thresholds/fv_dynamics.yaml:76-170;
 * tests/main/fv3core/test_config.py:14] */
int fv3_d_sw(fv3_ctx *, const fv3_field *delpc, const fv3_field *delp, const fv3_field *pt, const fv3_field *u,
             const fv3_field *v, const fv3_field *w, const fv3_field *uc, const fv3_field *vc,
             const fv3_field *ua, const fv3_field *va, const fv3_field *divgd, const fv3_field *mfx,
             const fv3_field *mfy, const fv3_field *cx, const fv3_field *cy, const fv3_field *crx,
             const fv3_field *cry, const fv3_field *xfx, const fv3_field *yfx, const fv3_field *q_con,
             const fv3_field *zh, const fv3_field *heat_source, const fv3_field *diss_est, double dt,
             void *stream);

/* UpdateHeightOnDGrid.__call__(zs, zh, crx, cry, xfx, yfx, wsd, dt) */
int fv3_update_dz_d(fv3_ctx *, const fv3_field *zs, const fv3_field *zh, const fv3_field *crx,
                    const fv3_field *cry, const fv3_field *xfx, const fv3_field *yfx, const fv3_field *wsd,
                    double dt, void *stream);

/* RiemannSolver3.__call__(last_call, dt, cappa, ptop, zs, wsd, delz, q_con, delp, pt, zh, pe, ppe, pk3, pk, peln, w) */
int fv3_riem_solver3(fv3_ctx *, int last_call, double dt, const fv3_field *cappa, double ptop,
                     const fv3_field *zs, const fv3_field *wsd, const fv3_field *delz, const fv3_field *q_con,
                     const fv3_field *delp, const fv3_field *pt, const fv3_field *zh, const fv3_field *pe,
                     const fv3_field *ppe, const fv3_field *pk3, const fv3_field *pk, const fv3_field *peln,
                     const fv3_field *w, void *stream);

/* PK3Halo.__call__(pk3, delp, ptop, akap) and the edge_pe stencil */
int fv3_pk3_halo(fv3_ctx *, const fv3_field *pk3, const fv3_field *delp, double ptop, double akap, void *stream);
int fv3_edge_pe(fv3_ctx *, const fv3_field *pe, const fv3_field *delp, double ptop, void *stream);

/* NonHydrostaticPressureGradient.__call__(u, v, pp, gz, pk3, delp, dt, ptop, akap) */
int fv3_nh_p_grad(fv3_ctx *, const fv3_field *u, const fv3_field *v, const fv3_field *pp, const fv3_field *gz,
                  const fv3_field *pk3, const fv3_field *delp, double dt, double ptop, double akap, void *stream);

/* RayleighDamping.__call__(u, v, w, dp, pfull, dt, ptop) (dp_ref / pfull live in the ctx)
 * [REF driver/examples/configs/baroclinic_c12.yaml:73-75] */
int fv3_ray_fast(fv3_ctx *, const fv3_field *u, const fv3_field *v, const fv3_field *w, double dt, double ptop,
                 void *stream);

/* HyperdiffusionDamping.__call__(q, cd) and apply_diffusive_heating */
int fv3_del2_cubed(fv3_ctx *, const fv3_field *q, double cd, int nmax, void *stream);
int fv3_apply_diffusive_heating(fv3_ctx *, const fv3_field *delp, const fv3_field *delz, const fv3_field *cappa,
                                const fv3_field *heat_source, const fv3_field *pt, double delt_time_factor,
                                void *stream);

/* small glue stencils of dyn_core: set_gz, copy, zero, gz = zh * grav */
int fv3_set_gz(fv3_ctx *, const fv3_field *zs, const fv3_field *delz, const fv3_field *gz, void *stream);
int fv3_copy(fv3_ctx *, const fv3_field *src, const fv3_field *dst, void *stream);
int fv3_zero(fv3_ctx *, const fv3_field *dst, void *stream);
int fv3_compute_geopotential(fv3_ctx *, const fv3_field *zh, const fv3_field *gz, void *stream);

/* ---- halo exchange (HaloUpdater pack / unpack) [REF docs/util/communication.rst:43-109] ----
 * A plan is a gather list built on the host from the partitioner geometry:
 * value[dst] = sign * source[src].  The same kernel packs (source = field, dst = buffer),
 * unpacks (source = buffer) and does the device-local copy between co-resident sub-domains. */
typedef struct fv3_gather_plan fv3_gather_plan;
/* dst_off / src_off: element offsets of the (sub, j, i) column base inside the respective
 * allocation (k stride is supplied per call); n entries; nk levels are moved per entry. */
int fv3_gather_plan_create(fv3_ctx *, fv3_gather_plan **out, int64_t n, const int64_t *dst_off,
                           const int64_t *src_off, const int8_t *sign);
int fv3_gather_plan_destroy(fv3_gather_plan *);
int fv3_gather_run(fv3_ctx *, const fv3_gather_plan *, void *dst, int64_t dst_kstride, const void *src,
                   int64_t src_kstride, int nk, void *stream);

/* ---- halo updaters behind the ABI (fv3_halo_plan_create / start / wait) ----------------------------
 * One plan = one HaloUpdater of the reference [REF docs/util/communication.rst:100-109,169-176]: a list of gather
 * operations on borrowed field pointers -- copies between co-resident sub-domains, packs into and unpacks out of one
 * message buffer per peer process -- and the peers with their message sizes.  start(): pack, post every message of the
 * update (RCCL: one ncclGroupStart/End of ncclSend/ncclRecv), local copies; wait(): unpack.  With a communicator
 * the exchange runs on the context's communication stream and overlaps what the caller enqueues in between. */
typedef struct fv3_halo_plan fv3_halo_plan;
enum { FV3_HALO_LOCAL = 0, FV3_HALO_PACK = 1, FV3_HALO_UNPACK = 2 };
typedef struct {
  const fv3_gather_plan *plan;
  void *dst;           /* LOCAL / UNPACK: destination field (all sub-domains); PACK: ignored (the peer's send buffer) */
  const void *src;     /* LOCAL / PACK: source field; UNPACK: ignored (the peer's receive buffer) */
  int64_t dst_kstride; /* elements between levels of the field side(s) */
  int64_t src_kstride;
  int64_t buf_off;     /* PACK / UNPACK: element offset of this piece inside the message buffer ... */
  int64_t buf_kstride; /* ... and the elements between its levels there */
  int32_t peer;        /* PACK / UNPACK: index into the plan's peer list */
  int32_t kind;        /* FV3_HALO_LOCAL / PACK / UNPACK */
  int32_t nk;          /* levels moved per gather entry */
  int32_t reserved;
} fv3_halo_op;
typedef struct {
  int32_t rank;        /* communicator rank of the peer process */
  int32_t reserved;
  int64_t send_elems, recv_elems;
  void *send_buf, *recv_buf; /* optional caller-owned message buffers (NULL: allocated by the plan, device memory) */
} fv3_halo_peer;
int fv3_halo_plan_create(fv3_ctx *, fv3_halo_plan **out, int n_ops, const fv3_halo_op *ops, int n_peers,
                         const fv3_halo_peer *peers);
int fv3_halo_plan_start(fv3_ctx *, fv3_halo_plan *, void *stream);
int fv3_halo_plan_wait(fv3_ctx *, fv3_halo_plan *, void *stream);
int fv3_halo_plan_destroy(fv3_halo_plan *);
int fv3_halo_plan_buffer(fv3_halo_plan *, int peer_index, int recv, void **ptr, int64_t *elems);

/* Transport.  RCCL point-to-point over xGMI: the 128-byte id is made on rank 0 (fv3_comm_unique_id), distributed by the
 * launcher's own channel, and every process calls fv3_ctx_comm_init (= ncclCommInitRank) once, before its first
 * exchange; librccl.so is bound at run time.  Host-driven transport (tests, gloo): the plan packs, calls
 * fn(user, plan, 0) [post the messages], later fn(user, plan, 1) [complete them], then unpacks. */
typedef struct { char internal[128]; } fv3_nccl_id;
int fv3_rccl_available(void); /* 1 when librccl.so could be bound in this process (what every rank checks BEFORE the collective calls) */
int fv3_comm_unique_id(fv3_nccl_id *id);
int fv3_ctx_comm_init(fv3_ctx *, const fv3_nccl_id *id, int world, int rank);
int fv3_ctx_comm_destroy(fv3_ctx *);
typedef int (*fv3_xfer_fn)(void *user, fv3_halo_plan *plan, int phase);
int fv3_ctx_set_xfer(fv3_ctx *, fv3_xfer_fn fn, void *user);
/* run the exchanges on the context's second stream (default: on with a communicator, off without) */
int fv3_ctx_set_comm_stream(fv3_ctx *, int on);
/* the stream the exchanges run on when that is switched on (hipStream_t as void*), else NULL: a host-driven transport enqueues its
 * own copies there so that they stay ordered with the plan's pack / unpack kernels */
void *fv3_ctx_get_comm_stream(fv3_ctx *);

/* ---- whole acoustic call [REF AcousticDynamics.__call__; SURVEY §3.3] ----------------------------
 * Temporaries AcousticDynamics owns (allocated by the host's QuantityFactory, borrowed here). */
typedef struct {
  fv3_field gz, zh, pkc, pk3;             /* interface fields (nz+1) */
  fv3_field crx, cry, xfx, yfx;           /* Courant numbers / area fluxes of d_sw */
  fv3_field divgd, ut, vt;                /* corner divergence, contravariant C-grid winds */
  fv3_field delpc, ptc;                   /* c_sw outputs */
  fv3_field dsw_delpc;                    /* d_sw work field (its "delpc" argument) */
  fv3_field heat_source;
  fv3_field ws3, wsd, zs;                 /* 2-D */
} fv3_workspace;

/* The 11 halo updaters of AcousticDynamics + the D-grid interface synchronisation. */
enum fv3_halo_update {
  FV3_HALO_Q_CON__CAPPA = 0,
  FV3_HALO_DELP__PT,
  FV3_HALO_U__V,
  FV3_HALO_W,
  FV3_HALO_GZ,
  FV3_HALO_DIVGD,
  FV3_HALO_UC__VC,
  FV3_HALO_DELP__PT__Q_CON,
  FV3_HALO_ZH,
  FV3_HALO_PKC,
  FV3_HALO_HEAT_SOURCE,
  FV3_HALO_INTERFACE_U__V,
  FV3_HALO_COUNT
};
/* Host-provided transport: phase 0 = start (pack + post), 1 = wait (complete + unpack), both
 * enqueued on `stream`.  Returns 0 on success.  The library never moves halos on its own. */
typedef int (*fv3_halo_fn)(void *user, int update /* fv3_halo_update */, int phase, void *stream);

/* The updaters as plans, indexed by fv3_halo_update: fv3_acoustic_step then needs no callback (halo = NULL) and no
 * host code runs between its operators. */
int fv3_ctx_set_halo_plans(fv3_ctx *, fv3_halo_plan *const *plans, int n);

/* n_split acoustic sub-steps (+ the once-per-call diffusive heating when d_con > 1e-5);
 * timestep = dt_atmos / k_split, n_map = 1..k_split.  halo == NULL: the registered plans (fv3_ctx_set_halo_plans).
 * The flux accumulators of the state (mfxd, mfyd, cxd, cyd) hold this call's fluxes on return [dyn_core.F90: emptied at the start of
 * every call]; the library zeroes each array in full the first time the context sees its pointer and afterwards has the first
 * sub-step store into the cells d_sw writes ("Flux accumulators").  With a halo callback the first sub-step asks for
 * FV3_HALO_ZH where the reference's sequence updates gz (the heights of a call are computed straight into zh). */
int fv3_acoustic_step(fv3_ctx *, const fv3_state *state, const fv3_workspace *work, double timestep, int n_map,
                      fv3_halo_fn halo, void *halo_user, void *stream);

/* ---- SURVEY §8f-3 (next row): sub-cycled tracer advection ---------------------------------------------------
 * TracerAdvection.__call__(tracers, dp1, mfxd, mfyd, cxd, cyd) [REF examples/notebooks/functions.py:916-951,
 * 1037-1044; savepoint Tracer2D1L, tests/savepoint/thresholds/fv_dynamics.yaml:328-360].  The one global quantity of
 * the operator is the Courant-number bound: fv3_tracer_2d_1l_cmax returns this process's maximum of
 * max(|cx|, |cy|) + 1 - sin_sg5 (it synchronises the stream), the host all-reduces it (MAX) and passes
 * n_split = (int)(1 + cmax).  cxd / cyd / mfxd / mfyd are scaled by 1 / n_split in place and dp1 ends as the air mass
 * before the last sub-cycle, like the reference's fields (Tracer2D1L-Out).  tracer_halo (may be NULL when n_split == 1):
 * the halo plan of the tracers, run between sub-cycles.  hord: 5 / 6, or 8 = PPM with the fast monotone constraint
 * (hord_tr of the reference configs [REF driver/examples/configs/baroclinic_c12.yaml:60]). */
int fv3_tracer_2d_1l_cmax(fv3_ctx *, const fv3_field *cxd, const fv3_field *cyd, double *cmax, void *stream);
int fv3_tracer_2d_1l(fv3_ctx *, int n_tracers, const fv3_field *const *tracers, const fv3_field *dp1,
                     const fv3_field *mfxd, const fv3_field *mfyd, const fv3_field *cxd, const fv3_field *cyd,
                     int n_split, int hord, fv3_halo_plan *tracer_halo, void *stream);

/* LagrangianToEulerian (the vertical remap that closes DynamicalCore.step_dynamics) [REF driver/pace/driver/driver.py:494-504,
 * 639-644; savepoint Remapping: tests/savepoint/thresholds/fv_dynamics.yaml:227-326; kord_tm -9, kord_mt / kord_tr / kord_wz 9,
 * consv_te 0: driver/examples/configs/baroclinic_c12.yaml:45,65-68].  In place: pt (the loop's theta_v / pkz form), delp, delz,
 * u, v, w and the tracers go from the Lagrangian layers (interfaces = pe / peln as the last acoustic sub-step left them,
 * pe's one-cell halo ring by edge_pe) to the Eulerian ones ak + bk * ps; pe, peln, pk, pkz and ps (2-D) are rebuilt.
 * Configuration of the reference configs: non-hydrostatic, T_v remapped in log(p), kord 9 everywhere, moist-cappa pkz with
 * the given cappa field, no energy fixer, no saturation adjustment, no fillz, omga untouched; nz >= 5. */
int fv3_remap(fv3_ctx *, int n_tracers, const fv3_field *const *tracers, const fv3_field *pt, const fv3_field *delp,
              const fv3_field *delz, const fv3_field *peln, const fv3_field *pe, const fv3_field *pk, const fv3_field *pkz,
              const fv3_field *u, const fv3_field *v, const fv3_field *w, const fv3_field *cappa, const fv3_field *ps,
              const fv3_field *wsd, void *stream);

/* ---- per-operator timing (HIP events on the operators' stream) --------------------------------- */
enum fv3_op {
  FV3_OP_C_SW = 0,
  FV3_OP_UPDATE_DZ_C,
  FV3_OP_RIEM_SOLVER_C,
  FV3_OP_P_GRAD_C,
  FV3_OP_D_SW,
  FV3_OP_UPDATE_DZ_D,
  FV3_OP_RIEM_SOLVER3,
  FV3_OP_PK3_HALO,
  FV3_OP_NH_P_GRAD,
  FV3_OP_RAY_FAST,
  FV3_OP_DIFFUSIVE_HEATING,
  FV3_OP_GLUE,
  FV3_OP_HALO,
  FV3_OP_COUNT
};
int fv3_ctx_set_profiling(fv3_ctx *, int on); /* 1: record an event pair around every operator of fv3_acoustic_step; 2: around d_sw only; 0: off */
const char *fv3_op_name(int op);
/* accumulated milliseconds and call counts per fv3_op (arrays of FV3_OP_COUNT); waits for the events */
int fv3_profile_read(fv3_ctx *, double *ms_sum, int64_t *calls, int reset);

/* Device self-test of the hand-written fp64 arithmetic of the Riemann solvers (no reference counterpart; tests/test_device_math.py).
 * x, y, out: device pointers to n doubles.  which 0: out = x / y by the solvers' division sequence; 1: their log; 2: their exp;
 * 3: round trip through the 80 accumulation-register slots of one wave (n = 80 * 64).  fp64 values in both builds. */
int fv3_selftest_math(fv3_ctx *, int which, const double *x, const double *y, double *out, int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FV3_MI355X_H */
