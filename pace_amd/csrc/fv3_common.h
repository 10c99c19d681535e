// fv3_common.h -- shared device/host definitions of the MI355X FV3 acoustic kernels.
//
// Layout: every 3-D field is [n_sub][nk_alloc][nj_alloc][ni_alloc], i fastest, so the 64
// lanes of a wavefront read 64 consecutive i (512 B rows for fp64).  Kernels are written in
// the reference's local Fortran numbering (first compute cell 1, npx = nx+1) through the
// IX() macro so that every loop bound can be checked against the oracle / FV3 source.
//
// The stage bodies are lambdas handed to launch3()/launch2(); under hipcc they run as
// gfx950 kernels, and the SAME source can be compiled with g++ (-DFV3_HOST_EMU) into a
// test-only library that executes the lambdas as host loops, so kernel logic is checked
// against the oracle in the GPU-less build container.  The host-emulation build is test
// infrastructure: the product loader refuses it (see pace_amd/lib.py).
#pragma once
#include <type_traits>
#include <algorithm>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fv3_mi355x.h"

#ifdef FV3_HOST_EMU
#define FV3_HD
#define FV3_DEV
typedef void *fv3_stream_t;
#else
#include <hip/hip_runtime.h>
#define FV3_HD __host__ __device__
#define FV3_DEV __device__
typedef hipStream_t fv3_stream_t;
#endif

#ifndef FV3_REAL
#define FV3_REAL double
#endif
typedef FV3_REAL Real;

// ---------------------------------------------------------------------------------------------
// In-kernel phase stamps (diagnostic builds only: -DFV3_STAMPS, never the product library).  rocprofv3's thread trace needs a
// decoder library this image does not ship, so the marches are timed from inside: s_memtime (shader clock) at the phase
// boundaries of a step, pinned with sched_barriers, the per-phase sums of a wave written to a record at its end.
// Record = {kernel id, steps, sum[0..5]} (32-bit cycle sums); word 0 of the buffer counts the records, word 1 selects one kernel
// id (0 = all: the buffer then fills with the first launches of a call).
// ---------------------------------------------------------------------------------------------
#if defined(FV3_STAMPS) && !defined(FV3_HOST_EMU)
#define FV3_STAMP_RECS 16384
unsigned long long *fv3_stamp_buf();  // (fv3_ctx.hip) device buffer, allocated on first use
#if defined(__HIP_DEVICE_COMPILE__)
#define FV3_STAMP_STATE unsigned st_t = (unsigned)__builtin_amdgcn_s_memtime(), st_a[6] = {0u, 0u, 0u, 0u, 0u, 0u}, st_n = 0u
#define FV3_STAMP(i)                                            \
  do {                                                          \
    __builtin_amdgcn_sched_barrier(0);                          \
    const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); \
    st_a[i] += t_ - st_t;                                       \
    st_t = t_;                                                  \
    if ((i) == 0) ++st_n;                                       \
    __builtin_amdgcn_sched_barrier(0);                          \
  } while (0)
#define FV3_STAMP_USE(x) asm volatile("" ::"v"(x))
#define FV3_STAMP_FLUSH(buf, kid, tid)                                          \
  do {                                                                          \
    if ((tid) == 0 && ((buf)[1] == 0ull || (buf)[1] == (kid))) {                \
      const unsigned long long slot_ = atomicAdd((buf), 1ull);                  \
      if (slot_ < FV3_STAMP_RECS) {                                             \
        unsigned long long *r_ = (buf) + 8 + slot_ * 8;                         \
        r_[0] = (kid);                                                          \
        r_[1] = st_n;                                                           \
        for (int q_ = 0; q_ < 6; ++q_) r_[2 + q_] = st_a[q_];                   \
      }                                                                         \
    }                                                                           \
  } while (0)
#else
// host pass of hipcc: the kernel lambda must capture exactly what the device pass captures (the closure is the kernel argument)
inline void fv3_stamp_touch(unsigned long long *, unsigned long long, int) {}
#define FV3_STAMP_STATE unsigned st_n = 0u
#define FV3_STAMP(i) ((void)0)
#define FV3_STAMP_USE(x) ((void)0)
#define FV3_STAMP_FLUSH(buf, kid, tid) (fv3_stamp_touch((buf), (kid), (tid)), (void)st_n)
#endif
#else
#define FV3_STAMP_STATE ((void)0)
#define FV3_STAMP(i) ((void)0)
#define FV3_STAMP_USE(x) ((void)0)
#define FV3_STAMP_FLUSH(buf, kid, tid) ((void)0)
#endif

// Stores of the marching kernels: `if (owned) field[p] = v;`.  Because the block is conditional the compiler's wait for the rows
// requested BEFORE the stores of a step (one in-order memory counter) has to assume none was issued: the wait at the top of every
// step is `s_waitcnt vmcnt(0)`.  Round 4 measured the alternative -- every store issued, unowned lanes / rows writing to a per-wave
// row of a sink buffer (c->trash), so that the compiler counts and the wait becomes vmcnt(number of stores) (-DFV3_USTORE; checked
// in the ISA: vmcnt(7) / vmcnt(2) at the top of the two-tracer marches): d_sw 52.6 vs 52.6 ms, update_dz_d 11.2 vs 11.1 on the same
// box -- the stores are acknowledged long before the rows arrive, the drain costs nothing.  The conditional form is the default.
#define FV3_TRASH_SLOTS 4096
// element at a wave-uniform base + a per-lane byte offset of 32 bits (scalar-base addressing: see KW_ in fv3_nh.hip)
template <class T>
FV3_HD inline T *fv3_at(T *base, unsigned byte_off) {
  return (T *)((char *)base + byte_off);
}
template <class T>
FV3_HD inline const T *fv3_at(const T *base, unsigned byte_off) {
  return (const T *)((const char *)base + byte_off);
}

// element `idx` (unsigned, 32 bits) of a WAVE-UNIFORM pointer, addressed as scalar base + 32-bit byte offset.  `ptr[idx]` makes the
// compiler extend idx to 64 bits and shift it (it cannot prove idx * 8 fits 32 bits): a 64-bit vector add per access and a live
// 64-bit address; with the byte offset formed in 32 bits the access becomes `global_load v, v_off, s[base:base+1]`.  A plane of a
// sub-domain is far below 4 GB (the marches index inside one plane; the level / sub-domain offset is part of the uniform base).
#ifdef FV3_EL_PLAIN  // (A/B: the plain element access, per translation unit)
#define FV3_EL(ptr, idx) ((ptr)[idx])
#else
#define FV3_EL(ptr, idx) (*fv3_at((ptr), (unsigned)(idx) * (unsigned)sizeof(*(ptr))))
#endif

// streaming store: an output no kernel reads back soon bypasses the L2's normal allocation, so that the rows the kernel re-reads (its
// neighbours' cells, the metric terms) stay resident (c_sw's march gained 6 % from it in round 2).  -DFV3_NO_NT: plain stores (A/B).
#if !defined(FV3_HOST_EMU) && !defined(FV3_NO_NT)
#define FV3_ST_NT(lhs, val) __builtin_nontemporal_store((Real)(val), &(lhs))
#else
#define FV3_ST_NT(lhs, val) ((lhs) = (val))
#endif

FV3_HD inline void fv3_store_sel(Real *owned_dst, Real *sink, bool owned, Real v) {
#ifdef FV3_USTORE
  Real *d = owned ? owned_dst : sink;
  *d = v;
#else
  if (owned) FV3_ST_NT(*owned_dst, v);
  (void)sink;
#endif
}

// ... the same with the destination as (wave-uniform base, 32-bit element index): scalar-base addressing (FV3_EL)
FV3_HD inline void fv3_store_el(Real *base, unsigned idx, Real *sink, bool owned, Real v) {
#ifdef FV3_USTORE
  Real *d = owned ? fv3_at(base, idx * (unsigned)sizeof(Real)) : sink;
  *d = v;
#else
  if (owned) FV3_EL(base, idx) = v;
  (void)sink;
#endif
}

// scheduling fence (no instruction): the compiler may not move code across it.  The branch-free marches (PPM order as a constant)
// otherwise overlap the sweeps of a step until the register file overflows (12 - 14 spilled VGPRs); fenced at the points where the
// branchy form had its basic-block ends they fit again.
#if !defined(FV3_HOST_EMU) && defined(__HIP_DEVICE_COMPILE__)
#define FV3_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define FV3_SCHED_FENCE() ((void)0)
#endif

#define FV3_W 1
#define FV3_E 2
#define FV3_S 4
#define FV3_N 8

// Geometry + metric terms, captured by value in every stage lambda.
// 2-D metric terms (immutable after context creation).  Addressing them through the constant
// address space (address_space(4)) lets the compiler hoist them out of the level loop of launch3,
// but it then hoists ALL of them at once: the metric-heavy kernels (c_sw B/D, fxadv, the KE kernel)
// went to 155-256 VGPRs and ran 1.5-4x slower on MI355X, so the plain generic pointer stays.
#if defined(FV3_CONST_METRICS) && !defined(FV3_HOST_EMU)
typedef const Real __attribute__((address_space(4))) * MPtr;
#else
typedef const Real *MPtr;
#endif

struct Geo {
  int nx, ny, nz, nh, nsub;
  int npx, npy;
  int ni, nj, nkA;  // allocation extents
  int o;            // nh - 1: Fortran-local index -> storage index offset
  int sj32;         // j stride as 32-bit (in-plane offsets are computed in 32 bits: IX)
  long sj, sk, st;  // strides of 3-D fields: j, k, sub
  long st2;         // sub stride of 2-D fields
  unsigned char flags[FV3_MAX_SUB];
  MPtr dx, dy, dxa, dya, dxc, dyc, rdx, rdy, rdxa, rdya, rdxc, rdyc;
  MPtr area, rarea, area_c, rarea_c;
  MPtr cosa, sina, rsina, cosa_u, cosa_v, cosa_s, sina_u, sina_v, rsin_u, rsin_v, rsin2;
  MPtr sin_sg1, sin_sg2, sin_sg3, sin_sg4, cos_sg1, cos_sg2, cos_sg3, cos_sg4;
  MPtr sin_sg5;  // may be null (only tracer_2d_1l reads it)
  MPtr fC, f0, del6_u, del6_v, divg_u, divg_v;
  const Real *edge_w, *edge_e, *edge_s, *edge_n;
  const Real *corner_extrap;  // [nsub][4][3]
  const Real *dp_ref, *pfull; // [nz]
  const Real *ep_gam;         // [nz+1] edge_profile back-substitution factors (function of dp_ref)
  const Real *ep_gk, *ep_bet; // [nz+1] edge_profile forward-elimination ratios dp(k-1)/dp(k) and pivots (level-only)
  // per-level parameters (get_column_namelist), device arrays [nz+1]
  const int *nord, *nord_v, *nord_w, *nord_t;
  const Real *damp_vt, *damp_w, *damp_t, *d2_divg, *d_con, *ke_bg;
  Real da_min, da_min_c;
};

// storage offset of Fortran-local (i, j) inside one k-plane
#define IX(i, j) ((unsigned)(((j) + g.o) * g.sj32 + ((i) + g.o)))

struct Box {
  int i0, i1, j0, j1, k0, k1;  // inclusive; i, j Fortran-local; k 0-based
};

// per-level coefficient tables precomputed on the host at context creation (device arrays [nz+1])
struct DampTables {
  const Real *tp_vt;  // (damp_vt * da_min)^(nord_v+1)   fv_tp_2d of delp, pt
  const Real *tp_t;   // (damp_t  * da_min)^(nord_t+1)   fv_tp_2d of q_con
  const Real *d6_w;   // (damp_w  * da_min_c)^(nord_w+1) del6_vt_flux of w
  const Real *d6_vt;  // (damp_vt * da_min_c)^(nord_v+1) del6_vt_flux of vorticity
  const Real *dd8;    // (da_min_c * d4_bg)^(nord+1)     divergence damping
};

struct fv3_ctx {
  Geo g;
  const Geo *g_dev = nullptr;  // device-resident copy of g: rarely taken kernel paths read geometry through it
                               // instead of keeping dozens of kernel-argument scalars live in SGPRs
  DampTables tab;
  fv3_acoustic_config cfg;
  fv3_constants cst;
  int device;
  int dtype;
  int device_sync;
  double ptop;
  std::vector<double> ak, bk, dp_ref_h, pfull_h;
  std::vector<int> nord_h, nord_v_h, nord_w_h, nord_t_h;
  std::vector<double> damp_vt_h, damp_w_h, damp_t_h, d2_divg_h, d_con_h, ke_bg_h;
  std::vector<void *> owned;   // device allocations owned by the context
  std::vector<Real *> scratch; // 3-D scratch fields (full layout)
  int64_t scratch_bytes;
  // Rayleigh-damping table cache (ray_fast)
  void *tab_rf = nullptr;
  double rf_dt = 0.0, rf_ptop = 0.0, rf_dm = 0.0;
  int rf_nd = 0, rf_nn = 0;
  std::string err;
  // auxiliary stream: bandwidth-bound helper kernels (the del-n chains of d_sw) run there, one
  // operator ahead of the issue-bound transport kernels on the caller's stream (fv3_aux_* below)
  void *aux_stream = nullptr;
  void *aux_events[8] = {nullptr};
  int aux_on = 1;
  // halo exchange behind the ABI (fv3_halo.hip): communicator, communication stream + its two ordering events,
  // host-driven transport, the updaters registered for fv3_acoustic_step
  void *nccl_comm = nullptr;
  int comm_world = 1, comm_rank = 0;
  void *comm_stream = nullptr;
  void *comm_ev[2] = {nullptr, nullptr};
  int comm_stream_on = 0;
  fv3_xfer_fn xfer = nullptr;
  void *xfer_user = nullptr;
  fv3_halo_plan *halo_plans[FV3_HALO_COUNT] = {nullptr};
  Real *ak_dev = nullptr, *bk_dev = nullptr;  // hybrid-coordinate tables on the device (fv3_remap.hip)
  // Sink for the stores of lanes / rows a marching wave does not own (-DFV3_USTORE experiment builds only; see fv3_store_sel)
  Real *trash = nullptr;
  // Ping-pong of the four scalars d_sw rewrites (delp, pt, w, q_con; fv3_step.hip): d_sw's marches read the old fields through
  // their halo columns / rows while they produce the new ones, so they write beside them.  Inside fv3_acoustic_step the new
  // fields simply BECOME the state for the operators that follow (no copy-back); every second sub-step lands in the caller's
  // arrays again.  pp_buf: the alternate buffers (allocated by the first eligible fv3_acoustic_step call: fv3_pp_ensure; pp_state
  // 0 = not tried yet, 1 = allocated, -1 = switched off or the allocation failed); pp_from / pp_to: the pointer translation the
  // registered halo plans apply while the state lives in the alternates.
  Real *pp_buf[4] = {nullptr, nullptr, nullptr, nullptr};
  int pp_state = 0;
  // set by fv3_acoustic_step around its d_sw call: the workspace divgd is dead after the operator (the next c_sw overwrites it)
  bool seq_divgd_dead = false;
  // set by fv3_acoustic_step around the d_sw call of a call's FIRST sub-step: the four accumulators of the tracer sub-cycling (mfx, mfy, cx, cy) hold
  // nothing yet, so d_sw forms 0 + flux with the zero read from `zeros` (one level plane of zeros, 1.2 MB at C768: it stays in L2) instead of the field, and the sequencer
  // zeroes only the cells d_sw never writes (zero_unwritten, fv3_step.hip: every call, no cached state; FV3_ACC_STORE=0: zero + accumulate on every sub-step, as the reference does -- same values).
  bool seq_acc_first = false;
  // set by fv3_acoustic_step around c_sw: the operator may leave its last boundary-window launches (stages D and E: they write window cells of delpc / ptc / wc /
  // ke / vort / uc / vc, which update_dz_c does not touch) RUNNING on the auxiliary stream when it returns; csw_pending then says that the join (event 7) is still
  // owed -- the sequencer pays it before riem_solver_c, the first reader (fv3_csw_join).  The stand-alone operator always joins before it returns.
  bool seq_csw_defer = false;
  bool csw_pending = false;
  // Round 6: inside the sequencer update_dz_d leaves its last kernel -- the bottom-up scan that keeps the marched interface heights dz_min apart and forms
  // the surface vertical velocity -- to riem_solver3, whose wave form runs it as a pre-sweep of each column (fv3_nh.hip: PRE).  seq_dz_scan: the sequencer
  // allows it for the update_dz_d call it is about to make; dz_scan_src: the marched heights of a scan still to be done (null: none); dz_scan_min: dz_min.
  bool seq_dz_scan = false;
  // Round 6, deferred accumulation of the Courant numbers (fv3_step.hip): inside the sequencer crx / cry of every sub-step of a call stay in their own arrays
  // (acc_slots: crx, cry per sub-step -- fields fxadv writes anyway) and cx / cy are formed ONCE, at the end of the last d_sw of the call, as ((0 + s1) + s2) + ...
  // in the order the read-modify-write of every sub-step would have used.  seq_acc_defer: the sequencer asks the d_sw call it is about to make not to touch cx / cy.
  std::vector<Real *> acc_slots;
  int acc_state = 0;  // -1: the slots could not be allocated (or FV3_ACC_DEFER=0): read-modify-write in every sub-step
  bool seq_acc_defer = false;
  int seq_acc_sum_n = 0;  // > 0: this d_sw call is the last of the acoustic call -- it ends with the sum of that many sub-steps' arrays into the accumulators
  bool seq_uava_thin = false;  // the sequencer: this c_sw call is not the last of the acoustic call -- its A-grid winds are read by d_sw's divergence on the levels without a damping chain and by c_sw's own boundary windows next to the march's rectangle, nowhere else
  bool seq_delz_dead = false;  // the sequencer: this riem_solver3 call is not the last of the acoustic call -- nobody reads the layer thickness it would store (the next sub-step works from zh)
  Real *dz_scan_src = nullptr;
  Real dz_scan_min = (Real)0;
  bool seq_heat_first = false;  // ... the same for the accumulated damping heat (heat_source): d_sw's two heat sites form 0 + heat on the call's first sub-step
  Real *zeros = nullptr;
  const void *pp_from[4] = {nullptr, nullptr, nullptr, nullptr};
  void *pp_to[4] = {nullptr, nullptr, nullptr, nullptr};
  int pp_n = 0;
  // Frame-first passes (fv3_step.hip): the operators that feed a halo update (p_grad_c -> uc / vc; nh_p_grad + ray_fast -> u / v / w)
  // run their last kernel on the frame of every sub-domain first (the cells a neighbour's halo is packed from), the update starts,
  // and the interior follows while the messages travel.  0: whole box (default); 1: frame only; 2: interior only.
  int frame_pass = 0;
  // per-operator profiling (fv3_step.hip)
  int profiling = 0;
  struct ProfEvent {
    int op;
    void *e0, *e1;
    int parent;  // >= 0: recorded inside that operator's own event pair (its time is taken out of the parent's)
  };
  int prof_parent = -1;
  std::vector<ProfEvent> prof_events;
  double prof_ms[16] = {0};
  int64_t prof_n[16] = {0};
};

struct fv3_gather_plan {
  int64_t n;
  int64_t *dst_off;
  int64_t *src_off;
  signed char *sign;
};
// one gather of a batch (fv3_gather_run_jobs: the pack / local / unpack gathers of ONE halo update as one launch per FV3_GATHER_MAXOPS of them)
struct fv3_gather_job {
  const fv3_gather_plan *plan;
  void *dst;
  const void *src;
  int64_t dks, sks;
  int nk;
};
int fv3_gather_run_jobs(fv3_ctx *c, const fv3_gather_job *jobs, int n, void *stream);  // fv3_ctx.hip

// ---------------------------------------------------------------------------------------------
#define FV3_ACC_MAXSTEPS 12  // sub-steps per call the deferred accumulation handles (pointer table of its sum kernel)
// Two-stream helpers.  fv3_aux(c, s): the stream helper kernels go to (the caller's stream itself when
// the auxiliary stream is off or in the host emulation -- everything then runs in program order).
// fv3_signal(c, from, e) / fv3_wait(c, to, e): event e recorded on `from`, later awaited by `to`.
// ---------------------------------------------------------------------------------------------
fv3_stream_t fv3_aux(fv3_ctx *c, fv3_stream_t s);
void fv3_signal(fv3_ctx *c, fv3_stream_t from, int e);
void fv3_wait(fv3_ctx *c, fv3_stream_t to, int e);

// ---------------------------------------------------------------------------------------------
// error helpers
// ---------------------------------------------------------------------------------------------
extern std::string g_fv3_create_error;
int fv3_fail(fv3_ctx *c, int code, const std::string &msg);
int fv3_halo_step(fv3_ctx *c, int update, int phase, void *stream);  // fv3_halo.hip
void *fv3_dev_alloc(fv3_ctx *c, size_t bytes);
// FV3_ALT="name[,name...]": the named alternatives of the restatements DESIGN §2 lists as uncertain -- the same variable and names the
// oracle reads (oracle/fv3_oracle/util.py: alt), so that one run against reference savepoints can try them.  Read per call.
inline bool fv3_alt(const char *name) {
  const char *e = getenv("FV3_ALT");
  if (!e) return false;
  const size_t n = strlen(name);
  for (const char *p = e; (p = strstr(p, name)) != nullptr; p += n)
    if ((p == e || p[-1] == ',' || p[-1] == ' ') && (p[n] == 0 || p[n] == ',' || p[n] == ' ')) return true;
  return false;
}
bool fv3_pp_ensure(fv3_ctx *c);  // (fv3_ctx.hip) the ping-pong buffers of fv3_acoustic_step, allocated on first use
bool fv3_acc_slots_ensure(fv3_ctx *c, int n_sub_steps);  // (fv3_ctx.hip) the per-sub-step flux arrays of the deferred accumulation
void fv3_h2d(void *dst, const void *src, size_t bytes);
int fv3_post(fv3_ctx *c, fv3_stream_t s, const char *what);
// validate one field against the context layout; returns typed base pointer or nullptr
Real *fv3_chk(fv3_ctx *c, const fv3_field *f, const char *name, bool is2d = false);

#define FV3_FIELD(var, f)                \
  Real *var = fv3_chk(c, f, #f);         \
  if (!var) return FV3_ERR_ARG;
#define FV3_FIELD2D(var, f)              \
  Real *var = fv3_chk(c, f, #f, true);   \
  if (!var) return FV3_ERR_ARG;

// ---------------------------------------------------------------------------------------------
// launch: f(t, k, i, j) over a box for every sub-domain; f2(t, i, j) for column kernels
// ---------------------------------------------------------------------------------------------
#ifndef FV3_HOST_EMU
// XCD-aware workgroup order.  The 8 XCDs of an MI355X take workgroups round-robin in linear
// dispatch order, and each XCD has its own L2.  Kernels are launched on a grid
// (8, tiles-per-plane, ceil(planes / 8)): blockIdx.x is then exactly the XCD a workgroup lands on,
// every XCD walks the tiles of its OWN (sub-domain, level) plane in order, and the rows / columns
// neighbouring tiles share are served by that XCD's L2 instead of being fetched once per XCD.
// Few planes (the 2-D kernels and the column solvers have one per sub-domain: 3 on the per-GPU share of an 8-GPU run, 6 of a 4-GPU
// run, 12 of a 2-GPU run): with whole planes per XCD, 3 planes keep 3 of the 8 XCDs busy and 12 planes give four XCDs twice the
// work of the others.  Such launches cut every plane into 2^sshift contiguous runs of `tps` tiles ("sub-planes"), enough of
// them for a multiple of 8: an XCD still walks neighbouring tiles of one plane, and all eight get the same share.
struct GridMap {
  int gx;       // tiles per plane along i
  float rgx;    // 1 / gx  (tile index -> (bx, by) without an integer division)
  int nplanes;  // sub-domains * levels
  int sshift;   // log2 of the sub-planes per plane (0: whole planes)
  int tps;      // tiles per sub-plane (= tiles per plane when sshift == 0)
  int tpp;      // tiles per plane
};
__device__ inline bool fv3_tile(const GridMap &m, int &bx, int &by, int &bz) {
  const int v = (int)(blockIdx.z * 8 + blockIdx.x);
  bz = v >> m.sshift;
  if (bz >= m.nplanes) return false;
  const int y = (int)blockIdx.y + (v - (bz << m.sshift)) * m.tps;
  if (y >= m.tpp) return false;
  by = (int)(((float)y + 0.5f) * m.rgx);
  bx = y - by * m.gx;
  return true;
}
// KCH = levels walked by one thread.  1 for stencil kernels (a level loop makes the compiler hoist
// every level-invariant index / metric expression at once: the big stencils went to 155-256 VGPRs
// and ran up to 4x slower on MI355X); 4 for trivially pointwise kernels, where fewer, longer
// workgroups stream ~25% faster.
template <int KCH, class F>
__global__ void __launch_bounds__(256) fv3_k3(Box b, int nkc, GridMap m, F f) {
  int bx, by, kz;
  if (!fv3_tile(m, bx, by, kz)) return;
  const int i = b.i0 + (int)(bx * 64 + threadIdx.x);
  const int j = b.j0 + (int)(by * 4 + threadIdx.y);
  const int t = kz / nkc;
  if constexpr (KCH == 1) {
    const int k = b.k0 + (kz - t * nkc);
    if (i <= b.i1 && j <= b.j1) f(t, k, i, j);
  } else if constexpr (KCH == 2) {
    // two levels per thread as straight-line code (no loop: nothing to hoist); with the metric
    // terms in the constant address space the second body reuses the first one's metric reads
    const int k = b.k0 + (kz - t * nkc) * 2;
    if (i <= b.i1 && j <= b.j1) {
      f(t, k, i, j);
      if (k + 1 <= b.k1) f(t, k + 1, i, j);
    }
  } else {
    const int ka = b.k0 + (kz - t * nkc) * KCH;
    if (i <= b.i1 && j <= b.j1) {
#pragma unroll
      for (int kk = 0; kk < KCH; ++kk) {
        const int k = ka + kk;
        if (k > b.k1) break;
        f(t, k, i, j);
      }
    }
  }
}
template <class F>
__global__ void __launch_bounds__(256) fv3_k2(Box b, GridMap m, F f) {
  int bx, by, t;
  if (!fv3_tile(m, bx, by, t)) return;
  const int i = b.i0 + (int)(bx * 64 + threadIdx.x);
  const int j = b.j0 + (int)(by * 4 + threadIdx.y);
  if (i <= b.i1 && j <= b.j1) f(t, i, j);
}
// host side: grid + map for gx x gy tiles on nplanes planes (the float decode is checked once per shape)
inline GridMap fv3_grid(int gx, int gy, int nplanes, dim3 *grid) {
  const int tpp = gx * gy;
  int sshift = 0;
  static const bool no_split = getenv("FV3_GRID_SPLIT") && getenv("FV3_GRID_SPLIT")[0] == '0';  // A/B switch
  if (!no_split && nplanes < 64 && (nplanes & 7) != 0) {
    while (((nplanes << sshift) & 7) != 0) ++sshift;  // 8 / gcd(nplanes, 8) sub-planes per plane
    while (sshift > 0 && (1 << sshift) > tpp) --sshift;  // (never more sub-planes than tiles)
  }
  const int tps = (tpp + (1 << sshift) - 1) >> sshift;
  GridMap m{gx, 1.0f / (float)gx, nplanes, sshift, tps, tpp};
  static thread_local int ok_gx = 0, ok_n = 0;
  if (gx != ok_gx || tpp > ok_n) {
    for (int y = 0; y < tpp; ++y)
      if ((int)(((float)y + 0.5f) * m.rgx) != y / gx) abort();
    ok_gx = gx;
    ok_n = tpp;
  }
  *grid = dim3(8, tps, ((nplanes << sshift) + 7) / 8);
  return m;
}
#endif

// Levels one thread of a metric-heavy stage kernel walks: the 2-D metric terms of the point are loaded once,
// by hand, ahead of a ROLLED level loop (unrolled, the compiler hoists every level's loads at once).
// FV3_LAUNDER(i): inside such a level loop the point indices are passed through an empty asm so that the index /
// predicate arithmetic derived from them is recomputed per level instead of being hoisted out of the loop as
// hundreds of live registers (c_sw stage B went to 255 VGPRs, 1 wave / SIMD, without it).
#if !defined(FV3_HOST_EMU) && defined(__HIP_DEVICE_COMPILE__)
#define FV3_LAUNDER(x) asm volatile("" : "+v"(x))
// FV3_LANDED(x): x is the result of a load issued inside a RARE, wave-uniform branch of a marching kernel's hot loop (cube-corner
// remaps, tile-edge metric terms, corner-patch fluxes).  Without it the wait for that load is placed where the value is first used
// -- behind the point where the branch rejoins the hot path -- and, the memory counter being one in-order counter, the hot path
// then waits THERE for everything older than the rare load: the rows it has just requested for the next steps (seen in the ISA of
// every march: `s_waitcnt vmcnt(1)` at the head of phase 1, `vmcnt(0)` in phase 2: the prefetch distance was zero).  Reading the
// value through an empty asm inside the rare block makes the compiler put the wait there.
#ifdef FV3_NO_LANDED  // (A/B build)
#define FV3_LANDED(x) ((void)0)
#else
#define FV3_LANDED(x) asm volatile("" : "+v"(x))
#endif
#else
#define FV3_LAUNDER(x) ((void)0)
#define FV3_LANDED(x) ((void)0)
#endif
#ifndef FV3_KC
#define FV3_KC 8  // measured at C768: 2 -> 105.6, 4 -> 107.5, 8 -> 108.0, 16 -> 108.4 SDPD (8 keeps more workgroups for the multi-GPU loads)
#endif
#ifndef FV3_KCH_DEFAULT
#define FV3_KCH_DEFAULT 1
#endif
template <int KCH = FV3_KCH_DEFAULT, class F>
inline void launch3(const fv3_ctx *c, fv3_stream_t s, Box b, F f) {
  const int ni = b.i1 - b.i0 + 1, nj = b.j1 - b.j0 + 1, nk = b.k1 - b.k0 + 1;
  if (ni <= 0 || nj <= 0 || nk <= 0) return;
#ifdef FV3_HOST_EMU
  (void)s;
  const int nsub = c->g.nsub;
#pragma omp parallel for collapse(2) schedule(static)
  for (int t = 0; t < nsub; ++t)
    for (int k = b.k0; k <= b.k1; ++k)
      for (int j = b.j0; j <= b.j1; ++j)
        for (int i = b.i0; i <= b.i1; ++i) f(t, k, i, j);
#else
  dim3 block(64, 4, 1), grid;
  const int nkc = (nk + KCH - 1) / KCH;
  const GridMap m = fv3_grid((ni + 63) / 64, (nj + 3) / 4, c->g.nsub * nkc, &grid);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(fv3_k3<KCH, F>), grid, block, 0, s, b, nkc, m, f);
#endif
}

// ---------------------------------------------------------------------------------------------
// The frame of a sub-domain (the windows along its W / E / S / N boundary that the interior marching kernels leave to the
// generic per-point stage kernels): f(t, k, i, j) on w[0], w[1] (W / E: a few columns wide: 8 x 32 workgroups, fv3_k3n) and on
// w[2], w[3] (S / N: the usual 64 x 4 workgroups).  k runs over b.k0 .. b.k1 of
// w[0] (the level chunks).  ONE launch for the four windows (round 6); FV3_FRAME_LAUNCH=split: one launch per window (rounds 3 - 5, A/B).
// ---------------------------------------------------------------------------------------------
struct Frame {
  Box w[4];
};
#ifndef FV3_HOST_EMU
// narrow windows (a few columns wide): a workgroup is 8 columns x 32 rows, so a wave reads 8 rows of 64 contiguous bytes (lanes
// along j -- the transposed form -- made every lane of a load its own 128-byte line: 6 x the time of an S / N window of equal size)
template <class F>
__global__ void __launch_bounds__(256) fv3_k3n(Box b, int nkc, GridMap m, F f) {
  int bx, by, kz;
  if (!fv3_tile(m, bx, by, kz)) return;
  const int t = kz / nkc;
  const int k = b.k0 + (kz - t * nkc);
  const int i = b.i0 + (int)(bx * 8 + (threadIdx.x & 7)), j = b.j0 + (int)(by * 32 + (threadIdx.x >> 3));
  if (i <= b.i1 && j <= b.j1) f(t, k, i, j);
}
// the four windows in ONE launch (round 6): the workgroups of a plane are the 8 x 32 tiles of W, those of E, then the 64 x 4 tiles of S and of N -- each window
// in the thread layout its separate launch has.  (The merged launch of round 3 -- every window as 64 x 4 with the W / E lanes along j -- was slower than four.)
struct FrameMap {
  Box w[4];
  int first[4];  // first workgroup of the window within a plane
  int gx[4];     // the window's workgroups along i
};
inline int fv3_frame_map(const Frame &fr, FrameMap *fm) {
  int n = 0;
  for (int w = 0; w < 4; ++w) {
    Box b = fr.w[w];
    b.k0 = fr.w[0].k0;
    b.k1 = fr.w[0].k1;
    const int ni = b.i1 - b.i0 + 1, nj = b.j1 - b.j0 + 1;
    const int tx = ni <= 0 || nj <= 0 ? 0 : w < 2 ? (ni + 7) / 8 : (ni + 63) / 64, ty = ni <= 0 || nj <= 0 ? 0 : w < 2 ? (nj + 31) / 32 : (nj + 3) / 4;
    fm->w[w] = b;
    fm->first[w] = n;
    fm->gx[w] = tx > 0 ? tx : 1;
    n += tx * ty;
  }
  return n;
}
template <bool WIDX, class F>
__global__ void __launch_bounds__(256) fv3_kfr(FrameMap fm, int nkc, GridMap m, F f) {
  int bx, by, kz;
  if (!fv3_tile(m, bx, by, kz)) return;
  const int t = kz / nkc;
  const int k = fm.w[0].k0 + (kz - t * nkc);
  const int wi = bx >= fm.first[2] ? (bx >= fm.first[3] ? 3 : 2) : (bx >= fm.first[1] ? 1 : 0);
  const int lt = bx - fm.first[wi];
  const int ty = lt / fm.gx[wi], tx = lt - ty * fm.gx[wi];
  const Box b = fm.w[wi];
  const int tid = (int)threadIdx.x;
  const int i = wi < 2 ? b.i0 + tx * 8 + (tid & 7) : b.i0 + tx * 64 + (tid & 63);
  const int j = wi < 2 ? b.j0 + ty * 32 + (tid >> 3) : b.j0 + ty * 4 + (tid >> 6);
  if (i <= b.i1 && j <= b.j1) {
    if constexpr (WIDX)
      f(wi, t, k, i, j);
    else
      f(t, k, i, j);
  }
}
// FV3_FRAME_LAUNCH=split: one launch per window (A/B; read once)
inline bool fv3_frame_split() {
  static const bool v = getenv("FV3_FRAME_LAUNCH") && !strcmp(getenv("FV3_FRAME_LAUNCH"), "split");
  return v;
}
#endif
template <class F>
inline void launch_frame(const fv3_ctx *c, fv3_stream_t s, const Frame &fr, F f) {
#ifdef FV3_HOST_EMU
  for (int w = 0; w < 4; ++w) {
    Box b = fr.w[w];
    b.k0 = fr.w[0].k0;
    b.k1 = fr.w[0].k1;
    launch3(c, s, b, f);
  }
#else
  if (!fv3_frame_split()) {
    FrameMap fm;
    const int nt = fv3_frame_map(fr, &fm), nkc = fr.w[0].k1 - fr.w[0].k0 + 1;
    if (nt <= 0 || nkc <= 0) return;
    dim3 grid;
    const GridMap m = fv3_grid(nt, 1, c->g.nsub * nkc, &grid);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(fv3_kfr<false, F>), grid, dim3(256, 1, 1), 0, s, fm, nkc, m, f);
    return;
  }
  for (int w = 0; w < 4; ++w) {
    Box b = fr.w[w];
    b.k0 = fr.w[0].k0;
    b.k1 = fr.w[0].k1;
    if (w < 2) {
      const int ni = b.i1 - b.i0 + 1, nj = b.j1 - b.j0 + 1, nkc = b.k1 - b.k0 + 1;
      if (ni <= 0 || nj <= 0 || nkc <= 0) continue;
      dim3 grid;
      const GridMap m = fv3_grid((ni + 7) / 8, (nj + 31) / 32, c->g.nsub * nkc, &grid);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(fv3_k3n<F>), grid, dim3(256, 1, 1), 0, s, b, nkc, m, f);
    } else {
      launch3(c, s, b, f);
    }
  }
#endif
}

// launch_frame with the window index handed to the body: f(w, t, k, i, j), w = 0 (W), 1 (E), 2 (S), 3 (N) -- for bodies that
// test the sub-domain's tile-edge flag of their side and leave the corner cells to the column windows.  KCH 1 (k = level).
template <class F>
inline void launch_frame_w(const fv3_ctx *c, fv3_stream_t s, const Frame &fr, F f) {
#ifndef FV3_HOST_EMU
  if (!fv3_frame_split()) {
    FrameMap fm;
    const int nt = fv3_frame_map(fr, &fm), nkc = fr.w[0].k1 - fr.w[0].k0 + 1;
    if (nt <= 0 || nkc <= 0) return;
    dim3 grid;
    const GridMap m = fv3_grid(nt, 1, c->g.nsub * nkc, &grid);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(fv3_kfr<true, F>), grid, dim3(256, 1, 1), 0, s, fm, nkc, m, f);
    return;
  }
#endif
  for (int w = 0; w < 4; ++w) {
    Box b = fr.w[w];
    b.k0 = fr.w[0].k0;
    b.k1 = fr.w[0].k1;
    auto fw = [=] FV3_HD(int t, int k, int i, int j) { f(w, t, k, i, j); };
#ifdef FV3_HOST_EMU
    launch3(c, s, b, fw);
#else
    if (w < 2) {
      const int ni = b.i1 - b.i0 + 1, nj = b.j1 - b.j0 + 1, nkc = b.k1 - b.k0 + 1;
      if (ni <= 0 || nj <= 0 || nkc <= 0) continue;
      dim3 grid;
      const GridMap m = fv3_grid((ni + 7) / 8, (nj + 31) / 32, c->g.nsub * nkc, &grid);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(fv3_k3n<decltype(fw)>), grid, dim3(256, 1, 1), 0, s, b, nkc, m, fw);
    } else {
      launch3(c, s, b, fw);
    }
#endif
  }
}

// ---------------------------------------------------------------------------------------------
// launch3 restricted to up to four windows (the cube-corner patches of the marching kernels) in
// ONE launch: f(t, k, i, j) runs on natural ∩ window for every window.  The windows must be
// pairwise disjoint when f writes (callers check); n == 0 means "no restriction".
// ---------------------------------------------------------------------------------------------
struct Wins {
  int n;
  Box w[4];
};
FV3_HD inline Box fv3_clip(Box a, const Box &w) {
  Box r = a;
  r.i0 = a.i0 > w.i0 ? a.i0 : w.i0;
  r.i1 = a.i1 < w.i1 ? a.i1 : w.i1;
  r.j0 = a.j0 > w.j0 ? a.j0 : w.j0;
  r.j1 = a.j1 < w.j1 ? a.j1 : w.j1;
  return r;
}
inline bool fv3_wins_disjoint(const Wins &ws) {
  for (int a = 0; a < ws.n; ++a)
    for (int b = a + 1; b < ws.n; ++b)
      if (ws.w[a].i0 <= ws.w[b].i1 && ws.w[b].i0 <= ws.w[a].i1 && ws.w[a].j0 <= ws.w[b].j1 && ws.w[b].j0 <= ws.w[a].j1) return false;
  return true;
}
#ifndef FV3_HOST_EMU
template <class F>
__global__ void __launch_bounds__(256) fv3_k3w(Box nat, Wins ws, int nk, GridMap m, F f) {
  int bx, by, kz;
  if (!fv3_tile(m, bx, by, kz)) return;
  const int wi = kz % ws.n;
  const int tk = kz / ws.n;
  const int t = tk / nk;
  const int k = nat.k0 + (tk - t * nk);
  const Box b = fv3_clip(nat, ws.w[wi]);
  const int i = b.i0 + (int)(bx * 64 + threadIdx.x);
  const int j = b.j0 + (int)(by * 4 + threadIdx.y);
  if (i <= b.i1 && j <= b.j1) f(t, k, i, j);
}
#endif
template <class F>
inline void launch3w(const fv3_ctx *c, fv3_stream_t s, Box nat, const Wins &ws, F f) {
  if (ws.n == 0) {
    launch3(c, s, nat, f);
    return;
  }
#ifdef FV3_HOST_EMU
  for (int w = 0; w < ws.n; ++w) {
    Box b = fv3_clip(nat, ws.w[w]);
    launch3(c, s, b, f);
  }
#else
  int ni = 0, nj = 0;
  for (int w = 0; w < ws.n; ++w) {
    const Box b = fv3_clip(nat, ws.w[w]);
    ni = std::max(ni, b.i1 - b.i0 + 1);
    nj = std::max(nj, b.j1 - b.j0 + 1);
  }
  const int nk = nat.k1 - nat.k0 + 1;
  if (ni <= 0 || nj <= 0 || nk <= 0) return;
  dim3 block(64, 4, 1), grid;
  const GridMap m = fv3_grid((ni + 63) / 64, (nj + 3) / 4, c->g.nsub * nk * ws.n, &grid);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(fv3_k3w<F>), grid, block, 0, s, nat, ws, nk, m, f);
#endif
}

// launch3 / launch2 in frame-first passes: pass 0 = the whole box; 1 = its frame, the FV3_FRAME_W cells next to the box boundary
// (what the halo updates of the neighbours are packed from: 3 cells + 1 for the staggered fields); 2 = the rest
#define FV3_FRAME_W 4
inline bool fv3_frame_boxes(const Box &b, Box (&w)[4], Box &inner) {
  const int F = FV3_FRAME_W;
  if (b.i1 - b.i0 + 1 <= 2 * F || b.j1 - b.j0 + 1 <= 2 * F) return false;  // (all frame)
  w[0] = Box{b.i0, b.i0 + F - 1, b.j0 + F, b.j1 - F, b.k0, b.k1};  // W
  w[1] = Box{b.i1 - F + 1, b.i1, b.j0 + F, b.j1 - F, b.k0, b.k1};  // E
  w[2] = Box{b.i0, b.i1, b.j0, b.j0 + F - 1, b.k0, b.k1};          // S
  w[3] = Box{b.i0, b.i1, b.j1 - F + 1, b.j1, b.k0, b.k1};          // N
  inner = Box{b.i0 + F, b.i1 - F, b.j0 + F, b.j1 - F, b.k0, b.k1};
  return true;
}
// Columns of the compute domain (1..nx, 1..ny) numbered FRAME FIRST for the column solvers' frame-first passes: the F rows along
// the S and N edges, the F columns along W and E between them, then the interior, each part row-major (F = 0, or a domain that
// is all frame: plain row-major).  at(n) -> (i, j).
struct ColumnOrder {
  int nx, ny, F;
  FV3_HD bool split() const { return F > 0 && nx > 2 * F && ny > 2 * F; }
  FV3_HD int n_frame() const { return split() ? 2 * F * nx + 2 * F * (ny - 2 * F) : nx * ny; }
  FV3_HD void at(int n, int &i, int &j) const {
    if (!split()) {
      j = n / nx;
      i = 1 + n - j * nx;
      j += 1;
      return;
    }
    const int band = F * nx, side = F * (ny - 2 * F);
    if (n < 2 * band) {  // S rows 1 .. F, then N rows ny-F+1 .. ny
      const int north = n >= band ? 1 : 0, m = n - north * band, r = m / nx;
      i = 1 + m - r * nx;
      j = (north ? ny - F + 1 : 1) + r;
    } else if (n < 2 * band + 2 * side) {  // W columns 1 .. F, then E columns nx-F+1 .. nx, rows F+1 .. ny-F
      const int q = n - 2 * band, east = q >= side ? 1 : 0, m = q - east * side, r = m / F;
      i = (east ? nx - F + 1 : 1) + m - r * F;
      j = F + 1 + r;
    } else {
      const int m = n - 2 * band - 2 * side, w = nx - 2 * F, r = m / w;
      i = F + 1 + m - r * w;
      j = F + 1 + r;
    }
  }
};

template <class F>
inline void launch3_pass(const fv3_ctx *c, fv3_stream_t s, Box b, int pass, F f) {
  Box w[4], inner;
  if (pass == 0 || !fv3_frame_boxes(b, w, inner)) {
    if (pass != 2) launch3(c, s, b, f);
    return;
  }
  if (pass == 1)
    launch_frame(c, s, Frame{{w[0], w[1], w[2], w[3]}}, f);
  else
    launch3(c, s, inner, f);
}

template <class F>
inline void launch2(const fv3_ctx *c, fv3_stream_t s, Box b, F f);
template <class F>
inline void launch2_pass(const fv3_ctx *c, fv3_stream_t s, Box b, int pass, F f) {
  Box w[4], inner;
  if (pass == 0 || !fv3_frame_boxes(b, w, inner)) {
    if (pass != 2) launch2(c, s, b, f);
    return;
  }
  if (pass == 1) {
    for (const Box &x : w) launch2(c, s, x, f);
  } else {
    launch2(c, s, inner, f);
  }
}

template <class F>
inline void launch2(const fv3_ctx *c, fv3_stream_t s, Box b, F f) {
  const int ni = b.i1 - b.i0 + 1, nj = b.j1 - b.j0 + 1;
  if (ni <= 0 || nj <= 0) return;
#ifdef FV3_HOST_EMU
  (void)s;
  const int nsub = c->g.nsub;
#pragma omp parallel for collapse(2) schedule(static)
  for (int t = 0; t < nsub; ++t)
    for (int j = b.j0; j <= b.j1; ++j)
      for (int i = b.i0; i <= b.i1; ++i) f(t, i, j);
#else
  dim3 block(64, 4, 1), grid;
  const GridMap m = fv3_grid((ni + 63) / 64, (nj + 3) / 4, c->g.nsub, &grid);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(fv3_k2<F>), grid, block, 0, s, b, m, f);
#endif
}

// ---------------------------------------------------------------------------------------------
// launch_chain: a CHAIN of dependent stages on up to four small windows in ONE launch.  One workgroup owns one
// (window, sub-domain, level) and runs the stages one after the other with a workgroup barrier in between -- the
// stages communicate through global fields, but only inside their own window and level, so nothing else has to
// be ordered.  Replaces 6-7 launches of a few dozen microseconds each (the cube-corner patches of the marching
// kernels) by one.  A stage = {natural box, f(t, k, i, j)}; it runs on natural box ∩ window.
// ---------------------------------------------------------------------------------------------
template <class F>
struct ChainStage {
  Box nat;
  F f;
};
template <class F>
inline ChainStage<F> chain_stage(Box nat, F f) {
  return ChainStage<F>{nat, f};
}
#ifndef FV3_HOST_EMU
template <class... St>
__global__ void __launch_bounds__(256) fv3_kchain(Wins ws, int k0, int nk, int nplanes, St... st) {
  const int kz = (int)blockIdx.x;
  if (kz >= nplanes) return;
  const int wi = kz % ws.n;
  const int tk = kz / ws.n;
  const int t = tk / nk;
  const int k = k0 + (tk - t * nk);
  const int tid = (int)threadIdx.x;
  auto run = [&](const auto &sg) {
    const Box b = fv3_clip(sg.nat, ws.w[wi]);
    const int ni = b.i1 - b.i0 + 1, nj = b.j1 - b.j0 + 1;
    if (ni > 0 && nj > 0 && k >= sg.nat.k0 && k <= sg.nat.k1)
      for (int idx = tid; idx < ni * nj; idx += 256) {
        const int jj = idx / ni;
        sg.f(t, k, b.i0 + (idx - jj * ni), b.j0 + jj);
      }
    __syncthreads();
  };
  (run(st), ...);
}
#endif
template <class... St>
inline void launch_chain(const fv3_ctx *c, fv3_stream_t s, const Wins &ws, int k0, int k1, St... st) {
  const int nk = k1 - k0 + 1;
  if (ws.n <= 0 || nk <= 0) return;
#ifdef FV3_HOST_EMU
  auto run = [&](const auto &sg) {
    Box nat = sg.nat;
    nat.k0 = std::max(nat.k0, k0);
    nat.k1 = std::min(nat.k1, k1);
    launch3w(c, s, nat, ws, sg.f);
  };
  (run(st), ...);
#else
  const int nplanes = c->g.nsub * nk * ws.n;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(fv3_kchain<St...>), dim3(nplanes, 1, 1), dim3(256, 1, 1), 0, s, ws, k0, nk, nplanes, st...);
#endif
}

// ---------------------------------------------------------------------------------------------
// block-level launch for LDS-tiled kernels: f(blk, smem) runs once per workgroup with
// blk.tid / blk.nthr / blk.bx,by,bz and a dynamic LDS buffer; phases are separated by blk.sync().
// Work inside a phase is distributed as `for (w = blk.tid; w < n; w += blk.nthr)`, so the host
// emulation (one "thread" per block, sync = no-op) executes exactly the same phases in order.
// ---------------------------------------------------------------------------------------------
struct Blk {
  int tid, nthr, bx, by, bz;
  FV3_HD inline void sync() const {
#if !defined(FV3_HOST_EMU) && defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
#endif
  }
  // ordering point for LDS exchanges inside ONE wavefront (wave kernels): a wave's LDS
  // instructions execute in issue order, so only the compiler has to be kept from moving the
  // reads above the writes -- no s_barrier, and outstanding global prefetches stay in flight
  // (a full __syncthreads() would drain vmcnt to 0).
  // workgroup barrier of the wave-group kernels: LDS traffic of this wave complete, then s_barrier -- WITHOUT draining the
  // vector-memory counter (a __syncthreads() would wait for every prefetched row)
  FV3_HD inline void group_sync() const {
#if !defined(FV3_HOST_EMU) && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
  }
  FV3_HD inline void wave_sync() const {
#if !defined(FV3_HOST_EMU) && defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
  }
};

#ifndef FV3_HOST_EMU
template <class F>
__global__ void __launch_bounds__(256, 2) fv3_kb(GridMap m, F f) {
  extern __shared__ __attribute__((aligned(16))) char fv3_smem[];
  int bx, by, bz;
  if (!fv3_tile(m, bx, by, bz)) return;
  Blk b{(int)threadIdx.x, (int)blockDim.x, bx, by, bz};
  f(b, fv3_smem);
}
#endif

template <class F>
inline void launch_blocks(const fv3_ctx *c, fv3_stream_t s, int gx, int gy, int gz, int nthr, size_t smem_bytes, F f) {
  if (gx <= 0 || gy <= 0 || gz <= 0) return;
#ifdef FV3_HOST_EMU
  (void)c;
  (void)s;
  (void)nthr;
#pragma omp parallel
  {
    std::vector<char> smem(smem_bytes + 16);
#pragma omp for collapse(2) schedule(static)
    for (int z = 0; z < gz; ++z)
      for (int y = 0; y < gy; ++y)
        for (int x = 0; x < gx; ++x) {
          Blk b{0, 1, x, y, z};
          f(b, smem.data());
        }
  }
#else
  (void)c;
  dim3 grid;
  const GridMap m = fv3_grid(gx, gy, gz, &grid);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(fv3_kb<F>), grid, dim3(nthr, 1, 1), smem_bytes, s, m, f);
#endif
}

// ---------------------------------------------------------------------------------------------
// Wave kernels: one 64-lane wavefront per workgroup; a lane owns one column and marches along j
// keeping its sliding windows in registers, neighbours in i are exchanged through a few LDS row
// lines private to the wave (sync = the wave's own LDS ordering, no cross-wave barrier).
// Per-lane state is declared as arrays [FV3_LPT] and every phase is written as
// FV3_LANES(blk, lane, l) { ... state[l] ... }: on the device FV3_LPT = 1 (state lives in
// registers, the loop body runs once for lane = threadIdx.x); the host emulation runs the 64 lanes
// of a phase one after the other with FV3_LPT = 64.
// ---------------------------------------------------------------------------------------------
#define FV3_WAVE 64
#ifdef FV3_HOST_EMU
#define FV3_LPT 64
#define FV3_LANES(blk, lane, l) for (int lane = 0, l = 0; lane < FV3_WAVE; ++lane, ++l)
#else
#define FV3_LPT 1
#define FV3_LANES(blk, lane, l) for (int lane = (blk).tid, l = 0; l < 1; ++l)
// WPE = waves per SIMD the register allocation is sized for (512 / WPE VGPRs per lane)
template <int WPE, class F>
__global__ void __launch_bounds__(FV3_WAVE) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) fv3_kw(GridMap m, F f) {
  extern __shared__ __attribute__((aligned(16))) char fv3_smem[];
  int bx, by, bz;
  if (!fv3_tile(m, bx, by, bz)) return;
  Blk b{(int)threadIdx.x, FV3_WAVE, bx, by, bz};
  f(b, fv3_smem);
}
#endif

// ---------------------------------------------------------------------------------------------
// Neighbour reads inside a wave: lane_shr<K>(x) = the value lane - K holds in x (0 below lane 0), lane_shl<K>(x) = lane + K's
// (0 above lane 63).  On the device these are DPP moves (wave_shr:1 / wave_shl:1 of the GFX9 DPP set, two per fp64 value
// and shift step): register to register, no LDS line, no ordering point, a few cycles of latency instead of the LDS
// round trip.  bound_ctrl: a lane without a source (lane 0 of a shift right) reads 0 -- with the `old` operand instead, the compiler
// zeroes the destination before every DPP move (one more VALU instruction per 32-bit half: 72 per row of the two-tracer march).
// The host emulation reads the neighbour's slot of the per-lane array (the phases of a step run lane by
// lane there, so the value is complete when it is read).
// ---------------------------------------------------------------------------------------------
#if !defined(FV3_HOST_EMU) && defined(__HIP_DEVICE_COMPILE__)
FV3_DEV inline int fv3_dpp_shr1_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }  // wave_shr:1
FV3_DEV inline int fv3_dpp_shl1_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }  // wave_shl:1
FV3_DEV inline double fv3_dpp_shr1(double v) {
  return __hiloint2double(fv3_dpp_shr1_i(__double2hiint(v)), fv3_dpp_shr1_i(__double2loint(v)));
}
FV3_DEV inline double fv3_dpp_shl1(double v) {
  return __hiloint2double(fv3_dpp_shl1_i(__double2hiint(v)), fv3_dpp_shl1_i(__double2loint(v)));
}
FV3_DEV inline float fv3_dpp_shr1(float v) { return __int_as_float(fv3_dpp_shr1_i(__float_as_int(v))); }
FV3_DEV inline float fv3_dpp_shl1(float v) { return __int_as_float(fv3_dpp_shl1_i(__float_as_int(v))); }
template <int K>
FV3_DEV inline Real lane_shr_dev(Real v) {
#pragma unroll
  for (int n = 0; n < K; ++n) v = fv3_dpp_shr1(v);
  return v;
}
template <int K>
FV3_DEV inline Real lane_shl_dev(Real v) {
#pragma unroll
  for (int n = 0; n < K; ++n) v = fv3_dpp_shl1(v);
  return v;
}
#define FV3_LANE_SHR(K, arr, l, lane) lane_shr_dev<K>((arr)[l])
#define FV3_LANE_SHL(K, arr, l, lane) lane_shl_dev<K>((arr)[l])
#else
#define FV3_LANE_SHR(K, arr, l, lane) ((lane) >= (K) ? (arr)[(l) - (K)] : (Real)0)
#define FV3_LANE_SHL(K, arr, l, lane) ((lane) + (K) < FV3_WAVE ? (arr)[(l) + (K)] : (Real)0)
#endif

// ---------------------------------------------------------------------------------------------
// Several adjacent columns per lane (CPL): a wave then covers FV3_WAVE * CPL "virtual lanes", virtual lane vl = lane * CPL + l.
// Why: the marching kernels issue one 8-byte load / store per column and field, and at fp64 a CU retires such a wave
// instruction only every ~30-45 cycles (measured: the interior c_sw march moved 2.8 TB/s with 76 % of its wave cycles
// waiting to issue).  With two columns per lane the compiler merges the pair into one 16-byte access: half the vector-memory
// instructions per point, half the shuffles (the neighbour of slot 1 is the lane's own slot 0).
// Kernels declare per-lane state as arrays [FV3_VLPT(CPL)] and write phases as
//   FV3_VLANES(CPL, blk, vl, l) ... state[l] ... FV3_VLANES_END
// (on the device a compile-time loop over the CPL slots, so that `l` is a constant expression; the host emulation runs the
// virtual lanes one after the other).  FV3_VSHR / FV3_VSHL(CPL, K, arr, l, vl) = the value virtual lane vl -/+ K holds in arr.
// Bodies must not `continue` / `break` / `return` (the device form is a lambda).
// ---------------------------------------------------------------------------------------------
#ifdef FV3_HOST_EMU
#define FV3_VLPT(CPL) (FV3_WAVE * (CPL))
#define FV3_VLANES(CPL, blk, vl, l) for (int vl = 0; vl < FV3_WAVE * (CPL); ++vl) { const int l = vl; (void)l;
#define FV3_VLANES_END }
#define FV3_VSHR(CPL, K, arr, l, vl) ((vl) >= (K) ? (arr)[(l) - (K)] : (Real)0)
#define FV3_VSHL(CPL, K, arr, l, vl) ((vl) + (K) < FV3_WAVE * (CPL) ? (arr)[(l) + (K)] : (Real)0)
#else
#define FV3_VLPT(CPL) (CPL)
template <int N0, int N1, class F>
FV3_HD inline void fv3_cpl_for(F &&f) {
  if constexpr (N0 < N1) {
    f(std::integral_constant<int, N0>{});
    fv3_cpl_for<N0 + 1, N1>(f);
  }
}
#define FV3_VLANES(CPL, blk, vl, l) fv3_cpl_for<0, CPL>([&](auto l##_c) { constexpr int l = decltype(l##_c)::value; const int vl = (blk).tid * (CPL) + l; (void)vl;
#define FV3_VLANES_END });
#if defined(__HIP_DEVICE_COMPILE__)
template <int CPL, int K, int L>
FV3_DEV inline Real fv3_vshr(const Real (&arr)[CPL]) {
  constexpr int d = L - K;                                 // source slot, counted from this lane's slot 0
  constexpr int sh = d >= 0 ? 0 : (-d + CPL - 1) / CPL;    // lanes to the left
  return lane_shr_dev<sh>(arr[d + sh * CPL]);
}
template <int CPL, int K, int L>
FV3_DEV inline Real fv3_vshl(const Real (&arr)[CPL]) {
  constexpr int d = L + K, sh = d / CPL;                   // lanes to the right
  return lane_shl_dev<sh>(arr[d - sh * CPL]);
}
#define FV3_VSHR(CPL, K, arr, l, vl) fv3_vshr<CPL, K, l>(arr)
#define FV3_VSHL(CPL, K, arr, l, vl) fv3_vshl<CPL, K, l>(arr)
#else
#define FV3_VSHR(CPL, K, arr, l, vl) ((arr)[0])  // (host pass of hipcc: never executed)
#define FV3_VSHL(CPL, K, arr, l, vl) ((arr)[0])
#endif
// the CPL adjacent values of a lane as ONE access (16 bytes for two fp64 columns; only element alignment is promised)
typedef Real fv3_real2 __attribute__((ext_vector_type(2)));
typedef fv3_real2 fv3_real2u __attribute__((aligned(sizeof(Real))));
template <int CPL>
FV3_HD inline void fv3_ld_cpl(Real (&dst)[CPL], const Real *p) {
  if constexpr (CPL == 2) {
    const fv3_real2 v = *(const fv3_real2u *)p;
    dst[0] = v.x;
    dst[1] = v.y;
  } else {
#pragma unroll
    for (int q = 0; q < CPL; ++q) dst[q] = p[q];
  }
}
template <int CPL>
FV3_HD inline void fv3_st_cpl(Real *p, const Real (&src)[CPL], const bool (&own)[CPL]) {
  if constexpr (CPL == 2) {
    if (own[0] && own[1]) {
      fv3_real2 v;
      v.x = src[0];
      v.y = src[1];
      *(fv3_real2u *)p = v;
      return;
    }
  }
#pragma unroll
  for (int q = 0; q < CPL; ++q)
    if (own[q]) p[q] = src[q];
}
#endif

// Rows a marching wave owns (FV3_SEG overrides for experiments).  A segment costs 6 warm-up steps, so longer
// is cheaper per row as long as the launch still fills the chip many times over: 96 rows when that leaves
// >= 8 waves per resident slot (C768 on one GPU: 1 % faster than 64, measured), 64 otherwise (the per-GPU loads
// of the 4- and 8-GPU runs, where the tail of the launch matters more).  16 / 32-row segments were 1-7 %
// slower than 64 at every size measured on MI355X, 128 equal to 64.
inline int fv3_pick_seg(long waves_at_64, int wpe) {
  const char *e = getenv("FV3_SEG");  // (read per launch: the production-shape parity tests force 96 on small level counts)
  if (e && atoi(e) > 0) return atoi(e);
  const long slots = 256L * 4 * wpe;
  return waves_at_64 * 2 / 3 >= 8 * slots ? 96 : 64;
}

template <int WPE = 3, class F>
inline void launch_waves(const fv3_ctx *c, fv3_stream_t s, int gx, int gy, int gz, size_t smem_bytes, F f) {
  if (gx <= 0 || gy <= 0 || gz <= 0) return;
#ifdef FV3_HOST_EMU
  launch_blocks(c, s, gx, gy, gz, FV3_WAVE, smem_bytes, f);
#else
  (void)c;
  dim3 grid;
  const GridMap m = fv3_grid(gx, gy, gz, &grid);
  if (smem_bytes > 64 * 1024) {  // opt in to the large-LDS allocation (up to 160 KB per workgroup on gfx950)
    static size_t granted = 0;   // per kernel instantiation
    if (smem_bytes > granted) {
      (void)hipFuncSetAttribute((const void *)fv3_kw<WPE, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes);
      granted = smem_bytes;
    }
  }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(fv3_kw<WPE, F>), grid, dim3(FV3_WAVE, 1, 1), smem_bytes, s, m, f);
#endif
}

// Wave kernels launched as workgroups of NW waves: blk.by = the wave's index in its group, blk.group_sync() = the workgroup
// barrier for the LDS hand-offs between them (c_sw's interior march: the waves walk the same rows of NW levels and share the
// 25 metric terms of a row through LDS, each wave fetching a quarter of them).
#ifndef FV3_HOST_EMU
template <int WPE, int NW, class F>
__global__ void __launch_bounds__(FV3_WAVE * NW) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) fv3_kwg(GridMap m, F f) {
  extern __shared__ __attribute__((aligned(16))) char fv3_smem[];
  int bx, by, bz;
  if (!fv3_tile(m, bx, by, bz)) return;
  Blk b{(int)(threadIdx.x & (FV3_WAVE - 1)), FV3_WAVE, bx, __builtin_amdgcn_readfirstlane((int)(threadIdx.x / FV3_WAVE)), bz};
  f(b, fv3_smem);
}
#endif
template <int WPE, int NW, class F>
inline void launch_wave_groups(const fv3_ctx *c, fv3_stream_t s, int gx, int gz, size_t smem_bytes, F f) {
  if (gx <= 0 || gz <= 0) return;
#ifdef FV3_HOST_EMU
  launch_blocks(c, s, gx, NW, gz, FV3_WAVE, smem_bytes, f);
#else
  (void)c;
  dim3 grid;
  const GridMap m = fv3_grid(gx, 1, gz, &grid);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(fv3_kwg<WPE, NW, F>), grid, dim3(FV3_WAVE * NW, 1, 1), smem_bytes, s, m, f);
#endif
}

// ... on a three-dimensional launch grid, the wave's index in its group handed to the body: f(blk, smem, wave).  Workgroups whose waves run DIFFERENT roles
// on the same tile and hand rows to each other through LDS (the coupled two-tracer marches of d_sw, fv3_tp4x.hip).  Device only: the host emulation runs its
// "waves" one after the other to completion, so a hand-over between concurrently running waves has no emulation -- callers keep an uncoupled form for it.
#ifndef FV3_HOST_EMU
template <int WPE, int NW, class F>
__global__ void __launch_bounds__(FV3_WAVE * NW) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) fv3_kwg3(GridMap m, F f) {
  extern __shared__ __attribute__((aligned(16))) char fv3_smem[];
  int bx, by, bz;
  if (!fv3_tile(m, bx, by, bz)) return;
  Blk b{(int)(threadIdx.x & (FV3_WAVE - 1)), FV3_WAVE, bx, by, bz};
  f(b, fv3_smem, __builtin_amdgcn_readfirstlane((int)(threadIdx.x / FV3_WAVE)));
}
template <int WPE, int NW, class F>
inline void launch_wave_groups3(const fv3_ctx *c, fv3_stream_t s, int gx, int gy, int gz, size_t smem_bytes, F f) {
  if (gx <= 0 || gy <= 0 || gz <= 0) return;
  (void)c;
  dim3 grid;
  const GridMap m = fv3_grid(gx, gy, gz, &grid);
  if (smem_bytes > 64 * 1024) {
    static size_t granted = 0;
    if (smem_bytes > granted) {
      (void)hipFuncSetAttribute((const void *)fv3_kwg3<WPE, NW, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes);
      granted = smem_bytes;
    }
  }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(fv3_kwg3<WPE, NW, F>), grid, dim3(FV3_WAVE * NW, 1, 1), smem_bytes, s, m, f);
}
#endif

// ---------------------------------------------------------------------------------------------
// cube-corner halo reads.  The reference fills the 3x3 corner block in place before every
// directional sweep (copy_corners / fill_4corners / fill_corners); the fills are pure
// functions of edge-halo cells, so here the *read* is redirected instead: no extra launch,
// no in-place race.  DIR 1 = x sweep, 2 = y sweep.  [SURVEY A.13]
// ---------------------------------------------------------------------------------------------
template <int DIR>
FV3_HD inline unsigned cc_index(const Geo &g, int fl, int i, int j) {
  // returns IX of the cell a copy_corners'ed read of cell (i, j) resolves to
  if ((i >= 1 && i <= g.nx) || (j >= 1 && j <= g.ny)) return IX(i, j);
  const int npx = g.npx, npy = g.npy;
  if (i < 1 && j < 1) {
    if ((fl & (FV3_W | FV3_S)) != (FV3_W | FV3_S)) return IX(i, j);
    return DIR == 1 ? IX(j, 1 - i) : IX(1 - j, i);
  }
  if (i > g.nx && j < 1) {
    if ((fl & (FV3_E | FV3_S)) != (FV3_E | FV3_S)) return IX(i, j);
    return DIR == 1 ? IX(npx - j, i - npx + 1) : IX(npx - 1 + j, npx - i);
  }
  if (i > g.nx && j > g.ny) {
    if ((fl & (FV3_E | FV3_N)) != (FV3_E | FV3_N)) return IX(i, j);
    return DIR == 1 ? IX(npx + (j - npy), npy - 1 - (i - npx)) : IX(npx - 1 - (j - npy), npy + (i - npx));
  }
  if ((fl & (FV3_W | FV3_N)) != (FV3_W | FV3_N)) return IX(i, j);
  return DIR == 1 ? IX(npy - j, npy - 1 + i) : IX(j - npy + 1, npy - i);
}

template <int DIR>
FV3_HD inline Real cc(const Real *q, const Geo &g, int fl, int i, int j) {
  return q[cc_index<DIR>(g, fl, i, j)];
}

// fill_4corners read (c_sw, update_dz_c): only the two corner cells next to the edge exist
template <int DIR>
FV3_HD inline unsigned f4_index(const Geo &g, int fl, int i, int j) {
  if ((i >= 1 && i <= g.nx) || (j >= 1 && j <= g.ny)) return IX(i, j);
  const int npx = g.npx, npy = g.npy;
  if (i < 1 && j < 1 && (fl & (FV3_W | FV3_S)) == (FV3_W | FV3_S)) {
    if (DIR == 1) {
      if (j == 0 && i == 0) return IX(0, 1);
      if (j == 0 && i == -1) return IX(0, 2);
    } else {
      if (i == 0 && j == 0) return IX(1, 0);
      if (i == 0 && j == -1) return IX(2, 0);
    }
  } else if (i > g.nx && j < 1 && (fl & (FV3_E | FV3_S)) == (FV3_E | FV3_S)) {
    if (DIR == 1) {
      if (j == 0 && i == npx) return IX(npx, 1);
      if (j == 0 && i == npx + 1) return IX(npx, 2);
    } else {
      if (i == npx && j == 0) return IX(npx - 1, 0);
      if (i == npx && j == -1) return IX(npx - 2, 0);
    }
  } else if (i > g.nx && j > g.ny && (fl & (FV3_E | FV3_N)) == (FV3_E | FV3_N)) {
    if (DIR == 1) {
      if (j == npy && i == npx) return IX(npx, npy - 1);
      if (j == npy && i == npx + 1) return IX(npx, npy - 2);
    } else {
      if (i == npx && j == npy) return IX(npx - 1, npy);
      if (i == npx && j == npy + 1) return IX(npx - 2, npy);
    }
  } else if (i < 1 && j > g.ny && (fl & (FV3_W | FV3_N)) == (FV3_W | FV3_N)) {
    if (DIR == 1) {
      if (j == npy && i == 0) return IX(0, npy - 1);
      if (j == npy && i == -1) return IX(0, npy - 2);
    } else {
      if (i == 0 && j == npy) return IX(1, npy);
      if (i == 0 && j == npy + 1) return IX(2, npy);
    }
  }
  return IX(i, j);
}

// fill_corners for a corner-staggered (B-grid) scalar: points (i, j) with both indices
// strictly outside [1, npx] x [1, npy]
template <int DIR>
FV3_HD inline unsigned bc_index(const Geo &g, int fl, int i, int j) {
  const int npx = g.npx, npy = g.npy;
  if ((i >= 1 && i <= npx) || (j >= 1 && j <= npy)) return IX(i, j);
  if (i < 1 && j < 1) {
    if ((fl & (FV3_W | FV3_S)) != (FV3_W | FV3_S)) return IX(i, j);
    // target (1-a, 1-b), a,b >= 1
    const int a = 1 - i, b = 1 - j;
    return DIR == 1 ? IX(1 - b, a + 1) : IX(b + 1, 1 - a);
  }
  if (i < 1 && j > npy) {
    if ((fl & (FV3_W | FV3_N)) != (FV3_W | FV3_N)) return IX(i, j);
    const int a = 1 - i, b = j - npy;
    return DIR == 1 ? IX(1 - b, npy - a) : IX(b + 1, npy + a);
  }
  if (i > npx && j < 1) {
    if ((fl & (FV3_E | FV3_S)) != (FV3_E | FV3_S)) return IX(i, j);
    const int a = i - npx, b = 1 - j;
    return DIR == 1 ? IX(npx + b, a + 1) : IX(npx - b, 1 - a);
  }
  if ((fl & (FV3_E | FV3_N)) != (FV3_E | FV3_N)) return IX(i, j);
  const int a = i - npx, b = j - npy;
  return DIR == 1 ? IX(npx + b, npy - a) : IX(npx - b, npy + a);
}

FV3_HD inline Real fv3_sign(Real a, Real b) { return b >= (Real)0 ? fabs(a) : -fabs(a); }
FV3_HD inline Real fv3_max(Real a, Real b) { return a > b ? a : b; }
FV3_HD inline Real fv3_min(Real a, Real b) { return a < b ? a : b; }
