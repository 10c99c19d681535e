// fv3_halo.hip -- halo updaters behind the C ABI: fv3_halo_plan_create / start / wait, the RCCL
// point-to-point transport, and the registration that lets fv3_acoustic_step run without a host callback.
//
// Mirror of NDSL's HaloUpdater.start() / wait() [REF docs/util/communication.rst:100-109,169-176]: start() packs the
// send regions (gather kernels), posts every message of the update in ONE ncclGroupStart / ncclGroupEnd and runs the
// copies between co-resident sub-domains; wait() unpacks.  With a communicator the whole exchange runs on the
// context's communication stream, ordered against the caller's stream by one event per call, so every operator the
// sequencer issues between start() and wait() overlaps with it -- and nothing of it is Python.
//
// RCCL is bound at run time (dlopen / dlsym of librccl.so: the copy already in the process -- PyTorch's -- or the
// ROCm one), so the library has no link-time dependency on it and a single-process run never touches it.  The
// communicator is created by the host: fv3_comm_unique_id() on rank 0, the 128 bytes travel by whatever launcher
// channel exists (torchrun's store, a file), then fv3_ctx_comm_init() on every rank.
// Tests on one GPU / on CPU use the host-driven transport instead (fv3_ctx_set_xfer): the plan packs, calls the host
// with (plan, phase), the host moves the message buffers (gloo), the plan unpacks -- same plans, same buffers.
#include "fv3_ops.h"

#ifndef FV3_HOST_EMU
#include <dlfcn.h>
#endif

struct fv3_halo_plan {
  std::vector<fv3_halo_op> ops;
  std::vector<fv3_halo_peer> peers;
  std::vector<void *> send_buf, recv_buf;
  std::vector<char> own_send, own_recv;  // buffers allocated (and freed) by the plan
  int started = 0;
};

namespace {

// ---- RCCL, bound lazily -------------------------------------------------------------------------------------------
struct Rccl {
  void *h = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, fv3_nccl_id, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, void *) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, void *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string err;
  bool ok = false;
};

Rccl &rccl() {
  static Rccl r;
#ifndef FV3_HOST_EMU
  static bool tried = false;
  if (tried) return r;
  tried = true;
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char *n : names) {
    r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (r.h) break;
  }
  if (!r.h) {
    r.err = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "");
    return r;
  }
  auto sym = [&](const char *n) {
    void *p = dlsym(r.h, n);
    if (!p) r.err += std::string(" missing ") + n;
    return p;
  };
  r.GetUniqueId = (int (*)(void *))sym("ncclGetUniqueId");
  r.CommInitRank = (int (*)(void **, int, fv3_nccl_id, int))sym("ncclCommInitRank");
  r.CommDestroy = (int (*)(void *))sym("ncclCommDestroy");
  r.GroupStart = (int (*)())sym("ncclGroupStart");
  r.GroupEnd = (int (*)())sym("ncclGroupEnd");
  r.Send = (int (*)(const void *, size_t, int, int, void *, void *))sym("ncclSend");
  r.Recv = (int (*)(void *, size_t, int, int, void *, void *))sym("ncclRecv");
  r.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
  r.ok = r.err.empty();
#else
  r.err = "the host-emulation build has no RCCL transport";
#endif
  return r;
}

int run_ops(fv3_ctx *c, fv3_halo_plan *p, int kind, void *stream) {
  std::vector<fv3_gather_job> jobs;  // (the gathers of one kind of an update are independent of one another: one batch)
  jobs.reserve(p->ops.size());
  for (const fv3_halo_op &o : p->ops) {
    if (o.kind != kind) continue;
    void *dst = o.dst;
    const void *src = o.src;
    int64_t dks = o.dst_kstride, sks = o.src_kstride;
    for (int n = 0; n < c->pp_n; ++n) {  // the state fields that currently live in their alternate buffers (fv3_step.hip)
      if (dst == c->pp_from[n]) dst = c->pp_to[n];
      if (src == c->pp_from[n]) src = c->pp_to[n];
    }
    if (kind == FV3_HALO_PACK) {
      dst = (char *)p->send_buf[o.peer] + (size_t)o.buf_off * sizeof(Real);
      dks = o.buf_kstride;
    } else if (kind == FV3_HALO_UNPACK) {
      src = (const char *)p->recv_buf[o.peer] + (size_t)o.buf_off * sizeof(Real);
      sks = o.buf_kstride;
    }
    if (!dst || !src) return FV3_ERR_ARG;
    jobs.push_back(fv3_gather_job{o.plan, dst, src, dks, sks, o.nk});
  }
  return fv3_gather_run_jobs(c, jobs.data(), (int)jobs.size(), stream);
}

}  // namespace

extern "C" {

int fv3_halo_plan_create(fv3_ctx *c, fv3_halo_plan **out, int n_ops, const fv3_halo_op *ops, int n_peers, const fv3_halo_peer *peers) {
  if (!c || !out || n_ops < 0 || n_peers < 0 || (n_ops && !ops) || (n_peers && !peers)) return FV3_ERR_ARG;
  fv3_halo_plan *p = new fv3_halo_plan();
  p->ops.assign(ops, ops + n_ops);
  p->peers.assign(peers, peers + n_peers);
  for (const fv3_halo_op &o : p->ops) {
    const bool msg = o.kind == FV3_HALO_PACK || o.kind == FV3_HALO_UNPACK;
    if (!o.plan || (msg && (o.peer < 0 || o.peer >= n_peers)) || (o.kind == FV3_HALO_LOCAL && (!o.dst || !o.src)) || o.kind < 0 || o.kind > FV3_HALO_UNPACK) {
      delete p;
      return fv3_fail(c, FV3_ERR_ARG, "halo plan: bad operation (null gather plan / field pointer, or peer index out of range)");
    }
  }
  for (int i = 0; i < n_peers; ++i) {
    void *sb = peers[i].send_buf, *rb = peers[i].recv_buf;
    char os = 0, orc = 0;
    if (!sb && peers[i].send_elems > 0) {
      sb = fv3_dev_alloc(c, (size_t)peers[i].send_elems * sizeof(Real));
      os = 1;
    }
    if (!rb && peers[i].recv_elems > 0) {
      rb = fv3_dev_alloc(c, (size_t)peers[i].recv_elems * sizeof(Real));
      orc = 1;
    }
    if ((peers[i].send_elems > 0 && !sb) || (peers[i].recv_elems > 0 && !rb)) {
      delete p;
      return fv3_fail(c, FV3_ERR_NOMEM, "halo plan: message buffer allocation failed");
    }
    p->send_buf.push_back(sb);
    p->recv_buf.push_back(rb);
    p->own_send.push_back(os);
    p->own_recv.push_back(orc);
  }
  *out = p;
  return FV3_OK;
}

int fv3_halo_plan_destroy(fv3_halo_plan *p) {
  delete p;  // (plan-owned message buffers belong to the context's allocation list and go with it)
  return FV3_OK;
}

int fv3_halo_plan_buffer(fv3_halo_plan *p, int peer, int recv, void **ptr, int64_t *elems) {
  if (!p || peer < 0 || peer >= (int)p->peers.size() || !ptr || !elems) return FV3_ERR_ARG;
  *ptr = recv ? p->recv_buf[peer] : p->send_buf[peer];
  *elems = recv ? p->peers[peer].recv_elems : p->peers[peer].send_elems;
  return FV3_OK;
}

int fv3_halo_plan_start(fv3_ctx *c, fv3_halo_plan *p, void *stream) {
  if (!c || !p) return FV3_ERR_ARG;
  if (p->started) return fv3_fail(c, FV3_ERR_ARG, "halo plan: start() called twice without wait()");
  void *x = stream;
#ifndef FV3_HOST_EMU
  if (c->comm_stream_on && c->comm_stream) {
    // the exchange may start once everything enqueued so far on the caller's stream is done
    (void)hipEventRecord((hipEvent_t)c->comm_ev[0], (hipStream_t)stream);
    (void)hipStreamWaitEvent((hipStream_t)c->comm_stream, (hipEvent_t)c->comm_ev[0], 0);
    x = c->comm_stream;
  }
#endif
  int st = run_ops(c, p, FV3_HALO_PACK, x);
  if (st != FV3_OK) return st;
  if (!p->peers.empty()) {
    if (c->nccl_comm) {
#ifndef FV3_HOST_EMU
      Rccl &r = rccl();
      const int dt = sizeof(Real) == 8 ? 8 : 7;  // ncclFloat64 / ncclFloat32
      int rc = r.GroupStart();
      for (size_t i = 0; i < p->peers.size() && rc == 0; ++i) {
        const fv3_halo_peer &q = p->peers[i];
        if (q.recv_elems > 0) rc = r.Recv(p->recv_buf[i], (size_t)q.recv_elems, dt, q.rank, c->nccl_comm, x);
        if (rc == 0 && q.send_elems > 0) rc = r.Send(p->send_buf[i], (size_t)q.send_elems, dt, q.rank, c->nccl_comm, x);
      }
      const int rc2 = r.GroupEnd();
      if (rc != 0 || rc2 != 0) return fv3_fail(c, FV3_ERR_HIP, std::string("halo plan: RCCL send / recv failed: ") + (r.GetErrorString ? r.GetErrorString(rc ? rc : rc2) : "?"));
#endif
    } else if (c->xfer) {
      if (c->xfer(c->xfer_user, p, 0) != 0) return fv3_fail(c, FV3_ERR_ARG, "halo plan: the host transport reported an error (start)");
    } else {
      return fv3_fail(c, FV3_ERR_ARG, "halo plan: this update has messages for other processes but the context has no transport (fv3_ctx_comm_init / fv3_ctx_set_xfer)");
    }
  }
  st = run_ops(c, p, FV3_HALO_LOCAL, x);
  if (st != FV3_OK) return st;
  p->started = 1;
  return FV3_OK;
}

int fv3_halo_plan_wait(fv3_ctx *c, fv3_halo_plan *p, void *stream) {
  if (!c || !p) return FV3_ERR_ARG;
  if (!p->started) return FV3_OK;
  void *x = stream;
#ifndef FV3_HOST_EMU
  const bool cs = c->comm_stream_on && c->comm_stream;
  if (cs) x = c->comm_stream;
#endif
  if (!p->peers.empty() && !c->nccl_comm && c->xfer) {
    if (c->xfer(c->xfer_user, p, 1) != 0) return fv3_fail(c, FV3_ERR_ARG, "halo plan: the host transport reported an error (wait)");
  }
  const int st = run_ops(c, p, FV3_HALO_UNPACK, x);
  if (st != FV3_OK) return st;
#ifndef FV3_HOST_EMU
  if (cs) {
    // what follows on the caller's stream sees the filled halos
    (void)hipEventRecord((hipEvent_t)c->comm_ev[1], (hipStream_t)c->comm_stream);
    (void)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)c->comm_ev[1], 0);
  }
#endif
  p->started = 0;
  return FV3_OK;
}

int fv3_rccl_available(void) { return rccl().ok ? 1 : 0; }

int fv3_comm_unique_id(fv3_nccl_id *id) {
  if (!id) return FV3_ERR_ARG;
  Rccl &r = rccl();
  if (!r.ok) return fv3_fail(nullptr, FV3_ERR_UNSUPPORTED, "RCCL is not available: " + r.err);
  return r.GetUniqueId(id) == 0 ? FV3_OK : fv3_fail(nullptr, FV3_ERR_HIP, "ncclGetUniqueId failed");
}

int fv3_ctx_comm_init(fv3_ctx *c, const fv3_nccl_id *id, int world, int rank) {
  if (!c || !id || world < 1 || rank < 0 || rank >= world) return FV3_ERR_ARG;
  if (c->nccl_comm) return fv3_fail(c, FV3_ERR_ARG, "fv3_ctx_comm_init: this context already has a communicator (fv3_ctx_comm_destroy first)");
  Rccl &r = rccl();
  if (!r.ok) return fv3_fail(c, FV3_ERR_UNSUPPORTED, "RCCL is not available: " + r.err);
  void *comm = nullptr;
  const int rc = r.CommInitRank(&comm, world, *id, rank);
  if (rc != 0) return fv3_fail(c, FV3_ERR_HIP, std::string("ncclCommInitRank failed: ") + (r.GetErrorString ? r.GetErrorString(rc) : "?"));
  c->nccl_comm = comm;
  c->comm_world = world;
  c->comm_rank = rank;
  c->comm_stream_on = 1;  // there is a transfer to hide: run the exchanges on the communication stream
  return FV3_OK;
}

int fv3_ctx_comm_destroy(fv3_ctx *c) {
  if (!c) return FV3_ERR_ARG;
  if (c->nccl_comm) {
    Rccl &r = rccl();
    if (r.ok) (void)r.CommDestroy(c->nccl_comm);
    c->nccl_comm = nullptr;
  }
  return FV3_OK;
}

int fv3_ctx_set_xfer(fv3_ctx *c, fv3_xfer_fn fn, void *user) {
  if (!c) return FV3_ERR_ARG;
  c->xfer = fn;
  c->xfer_user = user;
  return FV3_OK;
}

int fv3_ctx_set_comm_stream(fv3_ctx *c, int on) {
  if (!c) return FV3_ERR_ARG;
  c->comm_stream_on = on ? 1 : 0;
  return FV3_OK;
}

void *fv3_ctx_get_comm_stream(fv3_ctx *c) { return c && c->comm_stream_on ? c->comm_stream : nullptr; }

int fv3_ctx_set_halo_plans(fv3_ctx *c, fv3_halo_plan *const *plans, int n) {
  if (!c || (n && !plans) || n < 0 || n > FV3_HALO_COUNT) return FV3_ERR_ARG;
  for (int i = 0; i < FV3_HALO_COUNT; ++i) c->halo_plans[i] = i < n ? plans[i] : nullptr;
  return FV3_OK;
}

}  // extern "C"

// the sequencer's halo step when no host callback is given
int fv3_halo_step(fv3_ctx *c, int update, int phase, void *stream) {
  if (update < 0 || update >= FV3_HALO_COUNT || !c->halo_plans[update])
    return fv3_fail(c, FV3_ERR_ARG, "acoustic_step: no halo plan registered for this update (fv3_ctx_set_halo_plans) and no callback given");
  return phase == 0 ? fv3_halo_plan_start(c, c->halo_plans[update], stream) : fv3_halo_plan_wait(c, c->halo_plans[update], stream);
}
