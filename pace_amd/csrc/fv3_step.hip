// fv3_step.hip -- fv3_acoustic_step: the sequencing of AcousticDynamics.__call__ behind one C entry
// (n_split sub-steps: c_sw, update_dz_c, riem_solver_c, p_grad_c, d_sw, update_dz_d, riem_solver3,
// pk3_halo / edge_pe, nh_p_grad, ray_fast, then the once-per-call diffusive heating), with the 11
// halo updates requested through a host callback (the host owns the transport: device-local gather
// plans + RCCL point-to-point).  Python twin: pace_amd/dyn_core.py (same order, used when a
// checkpointer is attached).  [SURVEY §3.3, §8 a1 / b; reference AcousticDynamics.__call__]
//
// Also here: per-operator timing with HIP events recorded on the stream the operators run on
// (fv3_ctx_set_profiling / fv3_profile_read) -- what bench.py reports per operator.
#include <chrono>

#include "fv3_ops.h"

namespace {

struct OpTimer {
  fv3_ctx *c;
  fv3_stream_t s;
  int id;
  bool on;
#ifdef FV3_HOST_EMU
  std::chrono::steady_clock::time_point t0;
#else
  hipEvent_t e0, e1;
#endif
  // profiling 1: every operator; 2: d_sw only (and what is nested in it: the halo start its time is reduced by) -- the roofline kernel of the bench line
  // at two event pairs per sub-step instead of ~50 (the multi-GPU runs, where a sub-step is 12 - 45 ms and the pairs cost up to 2 % of it)
  OpTimer(fv3_ctx *c_, fv3_stream_t s_, int id_)
      : c(c_), s(s_), id(id_), on(c_->profiling == 1 || (c_->profiling == 2 && (id_ == FV3_OP_D_SW || c_->prof_parent == FV3_OP_D_SW))) {
    if (!on) return;
#ifdef FV3_HOST_EMU
    t0 = std::chrono::steady_clock::now();
#else
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
#endif
  }
  ~OpTimer() {
    if (!on) return;
#ifdef FV3_HOST_EMU
    const double ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c->prof_ms[id] += ms_;
    c->prof_n[id] += 1;
    if (c->prof_parent >= 0) c->prof_ms[c->prof_parent] -= ms_;
#else
    (void)hipEventRecord(e1, s);
    c->prof_events.push_back({id, (void *)e0, (void *)e1, c->prof_parent});
#endif
  }
};

// part = 0: every plane cell of levels 0..nz-1 (the level nz of the allocation -- interface padding of a cell-centred field -- is the
// caller's: the alternate buffers never hold it); 1: the compute cells of a cell-centred field only; 2: the frame around them
int copy_part(fv3_ctx *c, const fv3_field *src, const fv3_field *dst, int part, void *stream) {
  FV3_FIELD(a, src) FV3_FIELD(b, dst)
  const Geo g = c->g;
  auto cp = [=] FV3_HD(int t, int k, int i, int j) {
    const long p = t * g.st + k * g.sk + IX(i, j);
    b[p] = a[p];
  };
  fv3_stream_t s = (fv3_stream_t)stream;
  const int ia = -g.o, ib = g.ni - 1 - g.o, ja = -g.o, jb = g.nj - 1 - g.o;
  if (part == 0) {
    launch3<4>(c, s, Box{ia, ib, ja, jb, 0, g.nz - 1}, cp);
  } else if (part == 1) {
    launch3<4>(c, s, Box{1, g.nx, 1, g.ny, 0, g.nz - 1}, cp);
  } else {
    launch3<4>(c, s, Box{ia, ib, ja, 0, 0, g.nz - 1}, cp);
    launch3<4>(c, s, Box{ia, ib, g.ny + 1, jb, 0, g.nz - 1}, cp);
    launch3<4>(c, s, Box{ia, 0, 1, g.ny, 0, g.nz - 1}, cp);
    launch3<4>(c, s, Box{g.nx + 1, ib, 1, g.ny, 0, g.nz - 1}, cp);
  }
  return fv3_post(c, s, "copy_part");
}

// copy_part(.., 2) of up to four fields in ONE launch, a thread per frame cell and level (the frame = every plane cell outside the compute
// cells: 4 x 391 + ... of 391^2 at C768 layout 2 x 2).  As four launches per field -- row bands 64 lanes wide, column bands with 4 of 64 lanes
// active -- the sixteen launches of a call took 3.2 ms for 90 MB; this one takes what its bytes cost.
int copy_frames(fv3_ctx *c, int n, const fv3_field *const *src, const fv3_field *const *dst, void *stream) {
  const Real *a[4] = {nullptr, nullptr, nullptr, nullptr};
  Real *b[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int q = 0; q < n && q < 4; ++q) {
    a[q] = fv3_chk(c, src[q], "copy_frames src");
    b[q] = fv3_chk(c, dst[q], "copy_frames dst");
    if (!a[q] || !b[q]) return FV3_ERR_ARG;
  }
  const Real *a0 = a[0], *a1 = a[1], *a2 = a[2], *a3 = a[3];
  Real *b0 = b[0], *b1 = b[1], *b2 = b[2], *b3 = b[3];
  const Geo g = c->g;
  const int ia = -g.o, ja = -g.o;
  const int nlo = 1 - ja, nhi = g.nj - 1 - g.o - g.ny;     // rows below / above the compute rows
  const int wlo = 1 - ia, whi = g.ni - 1 - g.o - g.nx;     // columns left / right of the compute columns
  const int n_rows = (nlo + nhi) * g.ni, n_cols = (wlo + whi) * g.ny, n_frame = n_rows + n_cols;
  const int W = (n_frame + 3) / 4;
  launch3<4>(c, (fv3_stream_t)stream, Box{1, W, 1, 4, 0, g.nz - 1}, [=] FV3_HD(int t, int k, int ii, int jj) {
    const int idx = (jj - 1) * W + (ii - 1);
    if (idx >= n_frame) return;
    int i, j;
    if (idx < n_rows) {  // the row bands: full width
      const int r = idx / g.ni;
      i = ia + (idx - r * g.ni);
      j = r < nlo ? ja + r : g.ny + 1 + (r - nlo);
    } else {  // the column bands beside the compute rows
      const int m = idx - n_rows, w = wlo + whi, r = m / w, cidx = m - r * w;
      j = 1 + r;
      i = cidx < wlo ? ia + cidx : g.nx + 1 + (cidx - wlo);
    }
    const long p = t * g.st + k * g.sk + IX(i, j);
    b0[p] = a0[p];
    if (b1) b1[p] = a1[p];
    if (b2) b2[p] = a2[p];
    if (b3) b3[p] = a3[p];
  });
  return fv3_post(c, (fv3_stream_t)stream, "copy_frames");
}

// What the first-sub-step form of the accumulators (fv3_ctx::seq_acc_first / seq_heat_first) leaves to the sequencer: the first d_sw of a call STORES 0 + flux on
// every face / cell it owns, so the only cells of mfx / mfy / cx / cy / heat_source that still need the reference's zero_data are the ones d_sw never writes -- the
// frame of every plane outside the operator's write set and the padding level nz.  They are zeroed here on EVERY call (a thread per frame cell and level, as
// copy_frames; ~4 % of a field for the five of them together), so the result does not depend on what the arrays held before the call: no "zeroed once" flag keyed on a
// pointer (round 5), which a host write into those cells or an allocator re-using the address went past unseen [REF tests/main/fv3core/test_dycore_call.py:169-190].
// Write sets (first sub-step; fxadv, the AIR march, the two heat sites): cx [1, nx+1] x [jsd, jed], cy [isd, ied] x [1, ny+1], mfx [1, nx+1] x [1, ny],
// mfy [1, nx] x [1, ny+1], heat_source [1, nx] x [1, ny].  A null field takes no part (it gets the full zero instead).
int zero_unwritten(fv3_ctx *c, const fv3_field *mfx_, const fv3_field *mfy_, const fv3_field *cx_, const fv3_field *cy_, const fv3_field *heat_, void *stream) {
  Real *f[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  const fv3_field *in[5] = {mfx_, mfy_, cx_, cy_, heat_};
  bool any = false;
  for (int q = 0; q < 5; ++q)
    if (in[q]) {
      f[q] = fv3_chk(c, in[q], "zero_unwritten");
      if (!f[q]) return FV3_ERR_ARG;
      any = true;
    }
  if (!any) return FV3_OK;
  Real *mfx = f[0], *mfy = f[1], *cx = f[2], *cy = f[3], *heat = f[4];
  const Geo g = c->g;
  const int ia = -g.o, ja = -g.o;
  const int nlo = 1 - ja, nhi = g.nj - 1 - g.o - g.ny;  // rows below / above the compute rows
  const int wlo = 1 - ia, whi = g.ni - 1 - g.o - g.nx;  // columns left / right of the compute columns
  const int n_rows = (nlo + nhi) * g.ni, n_cols = (wlo + whi) * g.ny, n_frame = n_rows + n_cols;
  const int W = (n_frame + 3) / 4;
  const int isd = 1 - g.nh, ied = g.nx + g.nh, jsd = 1 - g.nh, jed = g.ny + g.nh;
  launch3<4>(c, (fv3_stream_t)stream, Box{1, W, 1, 4, 0, g.nz - 1}, [=] FV3_HD(int t, int k, int ii, int jj) {
    const int idx = (jj - 1) * W + (ii - 1);
    if (idx >= n_frame) return;
    int i, j;
    if (idx < n_rows) {  // the row bands: full width
      const int r = idx / g.ni;
      i = ia + (idx - r * g.ni);
      j = r < nlo ? ja + r : g.ny + 1 + (r - nlo);
    } else {  // the column bands beside the compute rows
      const int m = idx - n_rows, w = wlo + whi, r = m / w, cidx = m - r * w;
      j = 1 + r;
      i = cidx < wlo ? ia + cidx : g.nx + 1 + (cidx - wlo);
    }
    const long p = t * g.st + k * g.sk + IX(i, j);
    const bool xf = i >= 1 && i <= g.nx + 1, yf = j >= 1 && j <= g.ny + 1;  // on an owned x face column / y face row
    const bool xc = i >= 1 && i <= g.nx, yc = j >= 1 && j <= g.ny;
    if (mfx && !(xf && yc)) mfx[p] = (Real)0;
    if (mfy && !(xc && yf)) mfy[p] = (Real)0;
    if (cx && !(xf && j >= jsd && j <= jed)) cx[p] = (Real)0;
    if (cy && !(yf && i >= isd && i <= ied)) cy[p] = (Real)0;
    if (heat) heat[p] = (Real)0;  // (every frame cell lies outside [1, nx] x [1, ny])
  });
  launch3<4>(c, (fv3_stream_t)stream, Box{ia, g.ni - 1 - g.o, ja, g.nj - 1 - g.o, g.nz, g.nz}, [=] FV3_HD(int t, int k, int i, int j) {
    const long p = t * g.st + k * g.sk + IX(i, j);
    if (mfx) mfx[p] = (Real)0;
    if (mfy) mfy[p] = (Real)0;
    if (cx) cx[p] = (Real)0;
    if (cy) cy[p] = (Real)0;
    if (heat) heat[p] = (Real)0;
  });
  return fv3_post(c, (fv3_stream_t)stream, "zero_unwritten");
}


}  // namespace

extern "C" int fv3_ctx_set_profiling(fv3_ctx *c, int on) {
  if (!c) return FV3_ERR_ARG;
  c->profiling = on;
  return FV3_OK;
}

extern "C" const char *fv3_op_name(int op) {
  static const char *names[FV3_OP_COUNT] = {"c_sw", "update_dz_c", "riem_solver_c", "p_grad_c", "d_sw", "update_dz_d", "riem_solver3",
                                            "pk3_halo_edge_pe", "nh_p_grad", "ray_fast", "diffusive_heating", "glue", "halo"};
  return op >= 0 && op < FV3_OP_COUNT ? names[op] : "";
}

extern "C" int fv3_profile_read(fv3_ctx *c, double *ms_sum, int64_t *calls, int reset) {
  if (!c || !ms_sum || !calls) return FV3_ERR_ARG;
#ifndef FV3_HOST_EMU
  for (auto &e : c->prof_events) {
    hipEvent_t e0 = (hipEvent_t)e.e0, e1 = (hipEvent_t)e.e1;
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) {
      c->prof_ms[e.op] += ms;
      c->prof_n[e.op] += 1;
      if (e.parent >= 0) c->prof_ms[e.parent] -= ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  c->prof_events.clear();
#endif
  for (int i = 0; i < FV3_OP_COUNT; ++i) {
    ms_sum[i] = c->prof_ms[i];
    calls[i] = c->prof_n[i];
    if (reset) {
      c->prof_ms[i] = 0.0;
      c->prof_n[i] = 0;
    }
  }
  return FV3_OK;
}

#define RUN(op, call)                   \
  do {                                  \
    OpTimer tm_(c, s, op);              \
    const int st_ = (call);             \
    if (st_ != FV3_OK) return st_;      \
  } while (0)
#define HALO(id, phase)                                                                                  \
  do {                                                                                                   \
    OpTimer tm_(c, s, FV3_OP_HALO);                                                                      \
    const int st_ = halo ? halo(halo_user, id, phase, stream) : fv3_halo_step(c, id, phase, stream);     \
    if (st_ != 0) return halo ? fv3_fail(c, FV3_ERR_ARG, "acoustic_step: the halo callback reported an error") : st_; \
  } while (0)

extern "C" int fv3_acoustic_step(fv3_ctx *c, const fv3_state *st, const fv3_workspace *ws, double timestep, int n_map, fv3_halo_fn halo, void *halo_user,
                                 void *stream) {
  if (!c || !st || !ws) return FV3_ERR_ARG;
  if (!halo) {
    bool any = false;
    for (int i = 0; i < FV3_HALO_COUNT; ++i) any = any || c->halo_plans[i] != nullptr;
    if (!any) return fv3_fail(c, FV3_ERR_ARG, "acoustic_step: a halo-exchange callback or registered halo plans are required (sub-domain halos are never implicit)");
  }
  fv3_stream_t s = (fv3_stream_t)stream;
  const fv3_acoustic_config &cf = c->cfg;
  const int n_split = cf.n_split;
  const double dt = timestep / n_split, dt2 = 0.5 * dt;
  const double ptop = c->ptop, akap = c->cst.rdgas / c->cst.cp_air;

  // Ping-pong of delp / pt / w / q_con (see fv3_ctx::pp_buf): d_sw writes the new fields into the other half of the pair and
  // the operators that follow -- and the registered halo plans, through the pointer translation -- use that half; after an
  // even number of sub-steps the state is back in the caller's arrays, after an odd number one copy brings it there.
  // Only with the registered plans (a host callback moves the caller's own arrays) and when the buffers exist.
  const bool pingpong = !halo && fv3_pp_ensure(c);
  fv3_field f_delp[2] = {st->delp, st->delp}, f_pt[2] = {st->pt, st->pt}, f_w[2] = {st->w, st->w}, f_qc[2] = {st->q_con, st->q_con};
  if (pingpong) {
    f_delp[1].ptr = c->pp_buf[0];
    f_pt[1].ptr = c->pp_buf[1];
    f_w[1].ptr = c->pp_buf[2];
    f_qc[1].ptr = c->pp_buf[3];
    c->pp_from[0] = st->delp.ptr;
    c->pp_from[1] = st->pt.ptr;
    c->pp_from[2] = st->w.ptr;
    c->pp_from[3] = st->q_con.ptr;
    for (int n = 0; n < 4; ++n) c->pp_to[n] = c->pp_buf[n];
  }
  c->pp_n = 0;  // 4 while the state lives in the alternate buffers
  int cur = 0;
  // Frame-first: with messages to other processes, the operators that feed a halo update compute the frame of every sub-domain
  // first, the update starts, the interior follows while the messages travel (uc / vc behind p_grad_c; u / v / w behind
  // nh_p_grad + ray_fast).  Same values either way; without a transport the split only costs launches, so it stays off
  // (FV3_FRAME_FIRST=1 / 0 forces it).
  bool frame_first = !halo && (c->nccl_comm != nullptr || c->xfer != nullptr);
  if (const char *e = getenv("FV3_FRAME_FIRST")) frame_first = !halo && e[0] == '1';
  bool w_started = false;
  c->frame_pass = 0;
  struct PpGuard {  // (no early return leaves the halo translation switched on, or c_sw's deferred windows unjoined)
    fv3_ctx *c;
    void *stream;
    ~PpGuard() {
      (void)fv3_csw_join(c, stream);
      c->pp_n = 0;
      c->frame_pass = 0;
      c->seq_divgd_dead = false;
      c->seq_acc_first = false;
      c->seq_heat_first = false;
      c->seq_csw_defer = false;
      c->seq_dz_scan = false;
      c->seq_delz_dead = false;
      c->seq_uava_thin = false;
      c->seq_acc_defer = false;
      c->seq_acc_sum_n = 0;
      c->dz_scan_src = nullptr;
    }
  } pp_guard{c, stream};

  if (pingpong) {
    // cells no operator and no halo update ever writes (the 3 x 3 blocks beyond a cube corner, the allocation padding) keep the
    // caller's values in both halves of a pair, so that nothing -- not even a discarded corner value -- depends on the half
    const fv3_field *fs[4] = {&st->delp, &st->pt, &st->w, &st->q_con}, *fd[4] = {&f_delp[1], &f_pt[1], &f_w[1], &f_qc[1]};
    RUN(FV3_OP_GLUE, copy_frames(c, 4, fs, fd, stream));
  }
  HALO(FV3_HALO_Q_CON__CAPPA, 0);
  HALO(FV3_HALO_DELP__PT, 0);
  HALO(FV3_HALO_U__V, 0);
  HALO(FV3_HALO_Q_CON__CAPPA, 1);
  // "Empty the flux capacitors" (dyn_core.F90): the accumulated mass fluxes / Courant numbers cover ONE call -- the tracer
  // advection that follows each call consumes exactly them (dp2 = dp1 + div(mfx) must be the air mass after this call)
  (void)n_map;
  // Round 5: the first sub-step's d_sw STORES 0 + flux instead of accumulating into zeroed fields (fv3_ctx::seq_acc_first: the zero is read from a one-plane
  // block, not from the field): four 2.3 GB zero launches and four field reads less per call.  Round 6: the cells d_sw never writes (frame of every plane, padding
  // level) are zeroed on every call by zero_unwritten -- whatever the arrays held before the call, every cell ends up with what zero + accumulate leaves there.
  // FV3_ACC_STORE=0: zero + accumulate on every sub-step (A/B; same bits: 0 + x is what the accumulation computes on a zeroed field).
  const char *acc_env = getenv("FV3_ACC_STORE");  // (read per call: the parity test flips it in one process)
  const bool acc_store = !(acc_env && acc_env[0] == '0') && n_split > 0 && dsw_honors_acc_first(c);
  if (!acc_store) {
    const fv3_field *acc[4] = {&st->mfxd, &st->mfyd, &st->cxd, &st->cyd};
    for (int a = 0; a < 4; ++a) RUN(FV3_OP_GLUE, fv3_zero(c, acc[a], stream));
  }
  // (the accumulated damping heat the same way: the first sub-step's d_sw forms 0 + heat without reading it)
  const bool heat_reset = n_map == 1 || !fv3_alt("heat_zero_first_call");  // (FV3_ALT: DESIGN §2, uncertain restatement 5)
  // (not under the FV3_ALT: there the smoothed heat is copied back into the field after every call, never-read corner-halo cells included, and the full
  //  zero of a step's first call is what keeps those cells from drifting)
  const bool heat_store = acc_store && heat_reset && cf.d_con > 1.0e-5 && !fv3_alt("heat_zero_first_call");
  if (heat_reset && !heat_store) RUN(FV3_OP_GLUE, fv3_zero(c, &ws->heat_source, stream));
  // Round 6: cx / cy are formed once, at the end of the last d_sw of the call, from the sub-steps' own Courant-number arrays (fv3_dsw.hip: acc_sum; fv3_ctx::acc_slots).
  // FV3_ACC_DEFER=0 (or FV3_ACC_STORE=0, or a failed allocation of the 2 x n_split arrays): read-modify-write in every sub-step (A/B; same bits).
  const bool acc_defer = acc_store && n_split >= 2 && n_split <= FV3_ACC_MAXSTEPS && dsw_can_defer_acc(c) && fv3_acc_slots_ensure(c, n_split);
  if (getenv("FV3_DEBUG_FD")) fprintf(stderr, "[acoustic_step] accumulators: %s\n", acc_defer ? "cx / cy formed once per call from the sub-steps' Courant numbers, mfx / mfy read-modify-write" : acc_store ? "read-modify-write, first sub-step stores" : "zeroed, read-modify-write");
  if (acc_store)
    RUN(FV3_OP_GLUE, zero_unwritten(c, &st->mfxd, &st->mfyd, &st->cxd, &st->cyd, heat_store ? &ws->heat_source : nullptr, stream));
  RUN(FV3_OP_GLUE, fv3_zero(c, &st->diss_estd, stream));
  const char *gz_env = getenv("FV3_GZ_FIRST");  // (read per call: the parity test flips it in one process)
  const bool gz_direct = !(gz_env && !strcmp(gz_env, "copy"));
  for (int it = 0; it < n_split; ++it) {
    const int remap_step = it == n_split - 1;
    if (!w_started) HALO(FV3_HALO_W, 0);
    w_started = false;
    if (it == 0) {
      // Round 5: the heights of the call go straight into zh (with zh's halo plan) and update_dz_c runs its zh -> gz form, as in every other sub-step:
      // no gz -> zh copy (0.8 ms) and no in-place update_dz_c (two kernels, 4.6 ms, instead of one, 2.6).  The reference fills gz, updates its halo, copies
      // it to zh and updates gz in place -- the same values in every cell an operator reads (gz's outer halo cells, which only the copy would define, are
      // read by none).  FV3_GZ_FIRST=copy: the reference's order (A/B; same bits).
      RUN(FV3_OP_GLUE, fv3_set_gz(c, &ws->zs, &st->delz, gz_direct ? &ws->zh : &ws->gz, stream));
      HALO(gz_direct ? FV3_HALO_ZH : FV3_HALO_GZ, 0);
      HALO(FV3_HALO_DELP__PT, 1);
    }
    HALO(FV3_HALO_U__V, 1);
    HALO(FV3_HALO_W, 1);
    // (its last window launches may still run on the auxiliary stream beside update_dz_c: joined before riem_solver_c.  Only beside the one-kernel zh -> gz form of
    //  update_dz_c: the in-place form of the reference's first sub-step order works in the scratch field c_sw keeps its kinetic energy in)
    c->seq_csw_defer = (gz_direct || it > 0) && c->g.nz >= 3;
    c->seq_uava_thin = it < n_split - 1;  // (ua / va: outputs of the call -- the last sub-step's; before that only d_sw's divergence and c_sw's own windows read them: fv3_csw.hip, sua)
    RUN(FV3_OP_C_SW, fv3_c_sw(c, &f_delp[cur], &f_pt[cur], &st->u, &st->v, &f_w[cur], &st->uc, &st->vc, &st->ua, &st->va, &ws->ut, &ws->vt, &ws->divgd, &st->omga,
                              &ws->delpc, &ws->ptc, dt2, stream));
    c->seq_csw_defer = false;
    c->seq_uava_thin = false;
    if (cf.nord > 0) HALO(FV3_HALO_DIVGD, 0);
    if (it == 0 && gz_direct) {
      HALO(FV3_HALO_ZH, 1);
      RUN(FV3_OP_UPDATE_DZ_C, fv3_update_dz_c_from(c, &ws->zs, &ws->ut, &ws->vt, &ws->zh, &ws->gz, &ws->ws3, dt2, stream));
    } else if (it == 0) {
      HALO(FV3_HALO_GZ, 1);
      RUN(FV3_OP_GLUE, fv3_copy(c, &ws->gz, &ws->zh, stream));
      RUN(FV3_OP_UPDATE_DZ_C, fv3_update_dz_c(c, &ws->zs, &ws->ut, &ws->vt, &ws->gz, &ws->ws3, dt2, stream));
    } else {
      // (the reference copies zh into gz first; update_dz_c reads zh directly instead)
      RUN(FV3_OP_UPDATE_DZ_C, fv3_update_dz_c_from(c, &ws->zs, &ws->ut, &ws->vt, &ws->zh, &ws->gz, &ws->ws3, dt2, stream));
    }
    RUN(FV3_OP_GLUE, fv3_csw_join(c, stream));  // (what is left of c_sw's deferred windows when update_dz_c is done: timed as glue)
    RUN(FV3_OP_RIEM_SOLVER_C,
        fv3_riem_solver_c(c, dt2, &st->cappa, ptop, &st->phis, &ws->ws3, &ws->ptc, &f_qc[cur], &ws->delpc, &ws->gz, &ws->pkc, &st->omga, stream));
    if (frame_first) {
      c->frame_pass = 1;
      RUN(FV3_OP_P_GRAD_C, fv3_p_grad_c(c, &st->uc, &st->vc, &ws->delpc, &ws->pkc, &ws->gz, dt2, stream));
      HALO(FV3_HALO_UC__VC, 0);
      c->frame_pass = 2;
      RUN(FV3_OP_P_GRAD_C, fv3_p_grad_c(c, &st->uc, &st->vc, &ws->delpc, &ws->pkc, &ws->gz, dt2, stream));
      c->frame_pass = 0;
    } else {
      RUN(FV3_OP_P_GRAD_C, fv3_p_grad_c(c, &st->uc, &st->vc, &ws->delpc, &ws->pkc, &ws->gz, dt2, stream));
      HALO(FV3_HALO_UC__VC, 0);
    }
    if (cf.nord > 0) HALO(FV3_HALO_DIVGD, 1);
    HALO(FV3_HALO_UC__VC, 1);
    // (deferred accumulation: this sub-step's Courant numbers go to -- and are read from, by d_sw and update_dz_d -- arrays of its own)
    fv3_field f_crx = ws->crx, f_cry = ws->cry;
    if (acc_defer) {
      f_crx.ptr = c->acc_slots[2 * it];
      f_cry.ptr = c->acc_slots[2 * it + 1];
      c->seq_acc_defer = true;
      c->seq_acc_sum_n = it == n_split - 1 ? n_split : 0;  // (the last d_sw of the call ends with the sum of the sub-steps' arrays: fv3_dsw.hip, acc_sum)
    }
    if (pingpong) {
      // d_sw writes the new delp / pt / w / q_con into the other half of the pair.  Their halo update STARTS inside the operator, as
      // soon as the scalar marches are done, and is waited for after it: with a communicator the exchange (communication stream)
      // overlaps d_sw's whole wind part [REF docs/util/communication.rst:100-109,169-176: start() ... compute ... wait()]
      const int nxt = 1 - cur;
      struct Mid {
        fv3_ctx *c;
        int pp_n_next;
        void *stream;
        fv3_stream_t s;
      } mid{c, nxt ? 4 : 0, stream, s};
      auto start_halo = [](void *u) -> int {
        Mid *m = (Mid *)u;
        fv3_ctx *c = m->c;
        fv3_stream_t s = m->s;
        m->c->pp_n = m->pp_n_next;  // delp / pt / q_con now live in the half d_sw has just written
        c->prof_parent = FV3_OP_D_SW;  // (timed as halo, not as d_sw)
        int st_;
        {
          OpTimer tm_(c, s, FV3_OP_HALO);
          st_ = fv3_halo_step(m->c, FV3_HALO_DELP__PT__Q_CON, 0, m->stream);
        }
        c->prof_parent = -1;
        return st_;
      };
      c->seq_divgd_dead = getenv("FV3_SEQ_KEEP_DIVGD") == nullptr;
      c->seq_acc_first = acc_store && it == 0;
      c->seq_heat_first = heat_store && it == 0;
      RUN(FV3_OP_D_SW, fv3_d_sw_out(c, &ws->dsw_delpc, &f_delp[cur], &f_pt[cur], &st->u, &st->v, &f_w[cur], &st->uc, &st->vc, &st->ua, &st->va, &ws->divgd, &st->mfxd,
                                    &st->mfyd, &st->cxd, &st->cyd, &f_crx, &f_cry, &ws->xfx, &ws->yfx, &f_qc[cur], &ws->zh, &ws->heat_source, &st->diss_estd, dt,
                                    stream, &f_delp[nxt], &f_pt[nxt], &f_w[nxt], &f_qc[nxt], +start_halo, &mid));
      c->seq_divgd_dead = false;
      c->seq_acc_first = false;
      c->seq_heat_first = false;
      cur = nxt;
      c->pp_n = cur ? 4 : 0;
    } else {
      c->seq_acc_first = acc_store && it == 0;
      c->seq_heat_first = heat_store && it == 0;
      RUN(FV3_OP_D_SW, fv3_d_sw(c, &ws->dsw_delpc, &f_delp[cur], &f_pt[cur], &st->u, &st->v, &f_w[cur], &st->uc, &st->vc, &st->ua, &st->va, &ws->divgd, &st->mfxd, &st->mfyd,
                                &st->cxd, &st->cyd, &f_crx, &f_cry, &ws->xfx, &ws->yfx, &f_qc[cur], &ws->zh, &ws->heat_source, &st->diss_estd, dt, stream));
      c->seq_acc_first = false;
      c->seq_heat_first = false;
      HALO(FV3_HALO_DELP__PT__Q_CON, 0);
    }
    c->seq_acc_defer = false;
    c->seq_acc_sum_n = 0;
    HALO(FV3_HALO_DELP__PT__Q_CON, 1);
    c->seq_dz_scan = true;  // (its closing scan becomes riem_solver3's pre-sweep: nothing between the two reads zh or wsd)
    RUN(FV3_OP_UPDATE_DZ_D, fv3_update_dz_d(c, &ws->zs, &ws->zh, &f_crx, &f_cry, &ws->xfx, &ws->yfx, &ws->wsd, dt, stream));
    c->seq_dz_scan = false;
    c->seq_delz_dead = it < n_split - 1;  // (delz is read after the last sub-step of a call only: fv3_nh.hip, store_delz)
    if (frame_first) {
      // the frame columns of the new zh / pkc first, their updates start, the interior columns follow beside the messages
      c->frame_pass = 1;
      RUN(FV3_OP_RIEM_SOLVER3, fv3_riem_solver3(c, remap_step, dt, &st->cappa, ptop, &ws->zs, &ws->wsd, &st->delz, &f_qc[cur], &f_delp[cur], &f_pt[cur], &ws->zh, &st->pe,
                                                &ws->pkc, &ws->pk3, &st->pk, &st->peln, &f_w[cur], stream));
      HALO(FV3_HALO_ZH, 0);
      HALO(FV3_HALO_PKC, 0);
      c->frame_pass = 2;
      RUN(FV3_OP_RIEM_SOLVER3, fv3_riem_solver3(c, remap_step, dt, &st->cappa, ptop, &ws->zs, &ws->wsd, &st->delz, &f_qc[cur], &f_delp[cur], &f_pt[cur], &ws->zh, &st->pe,
                                                &ws->pkc, &ws->pk3, &st->pk, &st->peln, &f_w[cur], stream));
      c->frame_pass = 0;
    } else {
      RUN(FV3_OP_RIEM_SOLVER3, fv3_riem_solver3(c, remap_step, dt, &st->cappa, ptop, &ws->zs, &ws->wsd, &st->delz, &f_qc[cur], &f_delp[cur], &f_pt[cur], &ws->zh, &st->pe,
                                                &ws->pkc, &ws->pk3, &st->pk, &st->peln, &f_w[cur], stream));
      HALO(FV3_HALO_ZH, 0);
      HALO(FV3_HALO_PKC, 0);
    }
    c->seq_delz_dead = false;
    if (remap_step) RUN(FV3_OP_PK3_HALO, fv3_edge_pe(c, &st->pe, &f_delp[cur], ptop, stream));
    RUN(FV3_OP_PK3_HALO, fv3_pk3_halo(c, &ws->pk3, &f_delp[cur], ptop, akap, stream));
    HALO(FV3_HALO_ZH, 1);
    HALO(FV3_HALO_PKC, 1);
    if (frame_first) {
      // the frame of the new u / v / w first (pressure gradient + Rayleigh damping), the updates start, the interior follows
      c->frame_pass = 1;
      RUN(FV3_OP_NH_P_GRAD, fv3_nh_p_grad_scaled(c, &st->u, &st->v, &ws->pkc, &ws->zh, &ws->pk3, &f_delp[cur], dt, ptop, akap, c->cst.grav, stream));
      if (cf.rf_fast) RUN(FV3_OP_RAY_FAST, fv3_ray_fast(c, &st->u, &st->v, &f_w[cur], dt, ptop, stream));
      if (it != n_split - 1) {
        HALO(FV3_HALO_U__V, 0);
        HALO(FV3_HALO_W, 0);  // (w is final too: the next sub-step finds its update in flight)
        w_started = true;
      } else {
        HALO(FV3_HALO_INTERFACE_U__V, 0);
      }
      c->frame_pass = 2;
      RUN(FV3_OP_NH_P_GRAD, fv3_nh_p_grad_scaled(c, &st->u, &st->v, &ws->pkc, &ws->zh, &ws->pk3, &f_delp[cur], dt, ptop, akap, c->cst.grav, stream));
      if (cf.rf_fast) RUN(FV3_OP_RAY_FAST, fv3_ray_fast(c, &st->u, &st->v, &f_w[cur], dt, ptop, stream));
      c->frame_pass = 0;
      if (it == n_split - 1) HALO(FV3_HALO_INTERFACE_U__V, 1);
    } else {
      // (the reference stores gz = g * zh first -- compute_geopotential; here the corner interpolation reads zh and scales it)
      RUN(FV3_OP_NH_P_GRAD, fv3_nh_p_grad_scaled(c, &st->u, &st->v, &ws->pkc, &ws->zh, &ws->pk3, &f_delp[cur], dt, ptop, akap, c->cst.grav, stream));
      if (cf.rf_fast) RUN(FV3_OP_RAY_FAST, fv3_ray_fast(c, &st->u, &st->v, &f_w[cur], dt, ptop, stream));
      if (it != n_split - 1) {
        HALO(FV3_HALO_U__V, 0);
      } else {
        HALO(FV3_HALO_INTERFACE_U__V, 0);
        HALO(FV3_HALO_INTERFACE_U__V, 1);
      }
    }
  }
  if (pingpong) {
    // Bring the state home exactly as the in-place sequence leaves it.  delp / pt / q_con: their halos were exchanged after the
    // last d_sw, in the buffer that holds them.  w: its halo dates from the top of the last sub-step, i.e. it sits in the
    // buffer d_sw READ from -- the other one.
    if (cur == 1) {  // odd number of sub-steps
      RUN(FV3_OP_GLUE, copy_part(c, &f_delp[1], &st->delp, 0, stream));
      RUN(FV3_OP_GLUE, copy_part(c, &f_pt[1], &st->pt, 0, stream));
      RUN(FV3_OP_GLUE, copy_part(c, &f_qc[1], &st->q_con, 0, stream));
      RUN(FV3_OP_GLUE, copy_part(c, &f_w[1], &st->w, 1, stream));
    } else if (n_split > 0) {
      const fv3_field *fs[1] = {&f_w[1]}, *fd[1] = {&st->w};
      RUN(FV3_OP_GLUE, copy_frames(c, 1, fs, fd, stream));
    }
    cur = 0;
    c->pp_n = 0;
  }
  if (cf.d_con > 1.0e-5) {
    HALO(FV3_HALO_HEAT_SOURCE, 0);
    HALO(FV3_HALO_HEAT_SOURCE, 1);
    const double cd = 0.20 * c->g.da_min;
    int nmax = cf.nord + 1;
    if (nmax > 3) nmax = 3;
    const double delt = (fv3_alt("heat_dt_full") ? timestep : dt) * cf.delt_max;  // (FV3_ALT: DESIGN §2, uncertain restatement 2)
    // Round 5: the three smoothing iterations and the heating of pt as one pass (fv3_del2x.hip; the smoothed field is stored only when the heat
    // is kept over calls -- FV3_ALT heat_zero_first_call); 1 = no fused form for this configuration (fewer iterations, tiny sub-domains,
    // FV3_DEL2_FUSED=0): the two staged operators, same bits.
    int fused_st;
    {
      OpTimer tm_(c, s, FV3_OP_DIFFUSIVE_HEATING);
      fused_st = fv3_del2_heat_fused(c, &ws->heat_source, cd, nmax, &st->delp, &st->delz, &st->cappa, &st->pt, delt < 0 ? -delt : delt, fv3_alt("heat_zero_first_call"), stream);
    }
    if (fused_st != FV3_OK && fused_st != 1) return fused_st;
    if (fused_st == 1) {
      RUN(FV3_OP_DIFFUSIVE_HEATING, fv3_del2_cubed(c, &ws->heat_source, cd, nmax, stream));
      RUN(FV3_OP_DIFFUSIVE_HEATING, fv3_apply_diffusive_heating(c, &st->delp, &st->delz, &st->cappa, &ws->heat_source, &st->pt, delt < 0 ? -delt : delt, stream));
    }
  }
  return FV3_OK;
}
