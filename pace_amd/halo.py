"""Halo exchange for the acoustic path: device-local gathers between co-resident sub-domains
and packed point-to-point messages between processes (one process per GPU, RCCL over xGMI
through ``torch.distributed``; ``gloo`` on CPU for the tests).

Mirror of NDSL's ``Communicator.get_scalar_halo_updater / get_vector_halo_updater`` ->
``HaloUpdater.start() / wait() / update()`` and ``synchronize_vector_interfaces``
[REF docs/util/communication.rst:43-109,152-196].  The reference builds per-neighbour slices
and rotates blocks with ``n_clockwise_rotations``; here one *gather list* per phase is derived
from the partitioner geometry (``topology.build_halo_map``), which the same HIP kernel uses to
pack, to unpack and to copy between co-resident sub-domains -- the shape NDSL's own CUDA
pack/unpack kernels have (flat index arrays).

Mapping of the reference's ranks to processes: contiguous blocks,
``process = rank // (total_ranks / world_size)``.  RCCL allows one communicator
rank per device, so the 24 logical ranks of C768 2x2 on 8 GPUs are 3 sub-domains per process;
neighbours on the same device never touch RCCL.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import weakref
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .quantity import Quantity
from .topology import STAGGER, CubedSpherePartitioner, GatherMap, build_halo_map, build_interface_sync_map

_KIND = {
    "cell": [STAGGER["cell"]],
    "corner": [STAGGER["corner"]],
    "dgrid": [STAGGER["dgrid_u"], STAGGER["dgrid_v"]],
    "cgrid": [STAGGER["cgrid_u"], STAGGER["cgrid_v"]],
}


class Layout:
    """Which global ranks live on which process."""

    def __init__(self, part: CubedSpherePartitioner, world_size: int = 1, proc: int = 0):
        if part.total_ranks % world_size:
            raise ValueError(f"{part.total_ranks} ranks cannot be split evenly over {world_size} processes")
        self.part = part
        self.world_size = world_size
        self.proc = proc
        self.per_proc = part.total_ranks // world_size
        self.local_ranks = list(range(proc * self.per_proc, (proc + 1) * self.per_proc))

    def owner(self, rank: int) -> int:
        return rank // self.per_proc

    def sub_index(self, rank: int) -> int:
        return rank - self.owner(rank) * self.per_proc


class _Plan:
    """A device gather plan (``fv3_gather_plan``)."""

    def __init__(self, sf, dst_off, src_off, sign):
        self.sf = sf
        self._lib = sf.lib  # (the plan may outlive the factory by a moment when both are dropped together)
        self.n = int(len(dst_off))
        self.h = C.c_void_p()
        d = np.ascontiguousarray(dst_off, dtype=np.int64)
        s = np.ascontiguousarray(src_off, dtype=np.int64)
        g = np.ascontiguousarray(sign, dtype=np.int8)
        self.offsets = (d, s, g)  # (kept for the loop-back transport, which runs an unpack plan backwards once)
        st = sf.lib.fv3_gather_plan_create(
            sf.ctx,
            C.byref(self.h),
            self.n,
            d.ctypes.data_as(C.POINTER(C.c_int64)),
            s.ctypes.data_as(C.POINTER(C.c_int64)),
            g.ctypes.data_as(C.POINTER(C.c_int8)),
        )
        if st != 0:
            raise RuntimeError("fv3_gather_plan_create failed")

    def run(self, dst_ptr, dks, src_ptr, sks, nk, stream):
        if self.n == 0:
            return
        st = self.sf.lib.fv3_gather_run(self.sf.ctx, self.h, dst_ptr, dks, src_ptr, sks, nk, stream)
        if st != 0:
            raise RuntimeError("fv3_gather_run failed: " + self.sf.lib.fv3_last_error(self.sf.ctx).decode())

    def __del__(self):
        try:
            if self.h:
                self._lib.fv3_gather_plan_destroy(self.h)
        except Exception:
            pass


@dataclass
class _Phase:
    """Everything precomputed for one (kind, n_halo) exchange pattern."""

    local: Dict[Tuple[int, int], _Plan] = field(default_factory=dict)  # (dst_comp, src_comp) -> plan
    # per peer process: pack plans per source component, unpack plans per destination component
    send: Dict[int, Dict[int, _Plan]] = field(default_factory=dict)
    send_count: Dict[int, int] = field(default_factory=dict)
    recv: Dict[int, Dict[int, _Plan]] = field(default_factory=dict)
    recv_count: Dict[int, int] = field(default_factory=dict)


class HaloExchanger:
    """Builds and runs halo updates for the sub-domains of one process.

    ONE exchanger per context: it owns the context's transport (the RCCL communicator or the host-transport callback), so every
    operator that exchanges halos on a StencilFactory goes through :meth:`shared` -- a second exchanger would initialise the
    communicator again / replace the callback under the first one's plans."""

    @classmethod
    def shared(cls, sf, layout: "Layout", group=None):
        ex = getattr(sf, "_halo_exchanger", None)
        if ex is not None:
            if ex.layout.world_size != layout.world_size or ex.layout.proc != layout.proc or ex.part.total_ranks != layout.part.total_ranks:
                raise ValueError("this StencilFactory already exchanges halos with a different process layout")
            if group is not None and ex.group is not group:
                raise ValueError("this StencilFactory already exchanges halos over another process group (one transport per context: build a second "
                                 "StencilFactory for a second group)")
            return ex
        ex = cls(sf, layout, group=group)
        sf._halo_exchanger = ex
        return ex

    @property
    def sf(self):
        """The StencilFactory this exchanger belongs to (held weakly, see __init__); a clear error instead of a ReferenceError deep
        inside update() when a caller kept an updater but dropped the factory."""
        sf = self._sf_ref()
        if sf is None:
            raise RuntimeError("the StencilFactory of this HaloExchanger has been released: keep the factory (or the harness / AcousticDynamics that "
                               "owns it) alive as long as its halo updaters are used")
        return sf

    def __init__(self, sf, layout: Layout, group=None, comm_stream=None):
        # (the factory keeps the exchanger -- `shared` -- so the way back is weak: a strong reference would close a cycle and the
        #  context's device memory would wait for the cycle collector instead of going when the last user drops the factory)
        self._sf_ref = weakref.ref(sf)
        self._sf_proxy = weakref.proxy(sf)  # what the exchanger's own plan objects hold (they live and die with the exchanger)
        self.layout = layout
        self.part = layout.part
        self.group = group
        s = sf.sizer
        self.ni, self.nj, self.nk = s.storage_shape
        self.sk = self.ni * self.nj
        self.st = self.sk * self.nk
        self._phases: Dict[Tuple[str, int], _Phase] = {}
        self._buffers: Dict[Tuple, torch.Tensor] = {}
        self._next_uid = 0  # message buffers are keyed by a per-exchanger serial number (id() of a collected updater can be recycled)
        # Second HIP stream for the exchange: pack / device-local copies / RCCL point-to-point / unpack
        # run there, ordered against the compute stream by two events (start: comm waits for compute;
        # wait: compute waits for comm), so everything the sequencer issues between start() and wait()
        # overlaps with the exchange.  Used when there is a network transfer to hide (world_size > 1);
        # with all sub-domains on one GPU the "exchange" is a few small copy kernels and the extra stream
        # measured 1 % slower (MI355X, C768), so it stays on the compute stream there.
        # FV3_HALO_STREAM=1 / 0 forces it on / off.
        want = os.environ.get("FV3_HALO_STREAM", "1" if layout.world_size > 1 else "0") != "0"
        if comm_stream is None and torch.device(sf.device).type == "cuda" and want:
            comm_stream = torch.cuda.Stream(device=sf.device)
        self.comm_stream = comm_stream
        # gloo cannot move device memory: with a gloo group and device-resident fields the packed
        # messages are staged through pinned host buffers (used to exercise the multi-process
        # path on a single-GPU box; RCCL takes the device buffers directly)
        self.host_staged = False
        if layout.world_size > 1 and torch.device(sf.device).type == "cuda" and not getattr(layout, "loopback", False):
            import torch.distributed as dist

            self.host_staged = dist.get_backend(group) == "gloo"
        # Native updaters (default): every HaloUpdater is a fv3_halo_plan -- pack / RCCL send+recv / local copies / unpack are
        # issued by the library (fv3_halo_plan_start / wait), fv3_acoustic_step needs no callback and no Python runs between
        # its operators.  FV3_HALO_NATIVE=0 keeps the torch.distributed path below (A/B reference).
        self.native = os.environ.get("FV3_HALO_NATIVE", "1") != "0"
        self.transport = None
        self.fallback_reason = None
        self._by_plan: Dict[int, "HaloUpdater"] = {}
        self._xfer_cb = None
        if self.native:
            lib = sf.lib
            if not sf.hostemu:
                lib.fv3_ctx_set_comm_stream(sf.ctx, 1 if want else 0)
            if layout.world_size > 1:
                if getattr(layout, "loopback", False):
                    self._init_loopback()
                else:
                    self._init_transport(group)

    def _init_transport(self, group):
        """RCCL through the library when the process group is nccl (device buffers go straight to ncclSend / ncclRecv on the
        context's communication stream); otherwise the host-driven transport: the library packs, calls back with
        (plan, phase), this process moves the message buffers with torch.distributed (gloo), the library unpacks.

        The ranks AGREE on the transport before anything collective happens: (1) every rank reports whether librccl could be
        bound (all-reduce MIN); (2) rank 0 always broadcasts a payload -- the id or an error marker; (3) after
        ncclCommInitRank every rank reports its status (all-reduce MIN).  A rank that cannot use RCCL therefore never leaves
        the others waiting in a collective; all of them fall back together to the torch.distributed path (loudly: the bench
        refuses to report a number over a transport nobody asked for, see ``transport`` / ``FV3_HALO_NATIVE=0``)."""
        import torch.distributed as dist

        from . import lib as _lib

        sf, lay = self.sf, self.layout
        backend = dist.get_backend(group)

        def agree(ok: bool) -> bool:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=sf.device if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return int(flag.item()) == 1

        def fall_back(why):
            if lay.proc == 0:
                print(f"[pace_amd.halo] native RCCL transport unavailable ({why}); every rank uses the torch.distributed path", flush=True)
            self.native, self.transport = False, None
            self.fallback_reason = str(why)

        if backend == "nccl" and not sf.hostemu and os.environ.get("FV3_HALO_TRANSPORT", "rccl") == "rccl":
            if not agree(bool(sf.lib.fv3_rccl_available())):
                return fall_back("librccl.so could not be bound on some rank")
            ident = _lib.fv3_nccl_id()
            payload = [None]
            if lay.proc == 0:
                st = sf.lib.fv3_comm_unique_id(C.byref(ident))
                payload = [bytes(ident.internal) if st == 0 else ("error", sf.lib.fv3_last_error(None).decode())]
            dist.broadcast_object_list(payload, src=0, group=group)
            if isinstance(payload[0], tuple):
                return fall_back("fv3_comm_unique_id failed on rank 0: " + payload[0][1])
            C.memmove(C.byref(ident), payload[0], 128)
            st = sf.lib.fv3_ctx_comm_init(sf.ctx, C.byref(ident), lay.world_size, lay.proc)
            why = "" if st == 0 else sf.lib.fv3_last_error(sf.ctx).decode()
            if not agree(st == 0):
                if st == 0:
                    sf.lib.fv3_ctx_comm_destroy(sf.ctx)
                return fall_back("fv3_ctx_comm_init failed" + (": " + why if why else " on another rank"))
            self.transport = "rccl"
            return

        def xfer(_user, plan, phase):
            try:
                up = self._by_plan[int(plan)]
                up._host_transfer(int(phase))
                return 0
            except Exception as e:  # never let an exception cross the C frame
                self._xfer_error = e
                return 1

        self._xfer_cb = _lib.fv3_xfer_fn(xfer)
        sf.lib.fv3_ctx_set_xfer(sf.ctx, self._xfer_cb, None)
        self.transport = "host"

    def _init_loopback(self):
        """One process playing rank ``proc`` of ``world_size`` ALONE (bench.py --emulate-share): the plans, pack / unpack kernels,
        message buffers and start / wait protocol of the multi-process run, with every message looped back -- the bytes a peer
        would have sent are this process's own send buffer for that peer (a device copy stands in for the xGMI transfer).  The
        halo VALUES are therefore not the neighbours' (timing runs only; never a parity path)."""
        from . import lib as _lib

        if os.environ.get("FV3_LOOPBACK_TRANSPORT", "copy") == "rccl" and torch.device(self.sf.device).type == "cuda" and not self.sf.hostemu:
            # The looped-back messages through RCCL itself: a communicator of ONE rank, every message an ncclSend to / ncclRecv from
            # rank 0 posted by the library exactly as in a multi-GPU run (fv3_halo_plan_start: one group per update on the
            # communication stream).  What a single-GPU box can execute of the RCCL path: the binding (dlopen, ncclUniqueId by value,
            # datatype codes), communicator set-up and the group / stream protocol; not the inter-GPU transfer.
            sf = self.sf
            if not sf.lib.fv3_rccl_available():
                raise RuntimeError("FV3_LOOPBACK_TRANSPORT=rccl: librccl.so could not be bound")
            ident = _lib.fv3_nccl_id()
            if sf.lib.fv3_comm_unique_id(C.byref(ident)) != 0:
                raise RuntimeError("fv3_comm_unique_id failed: " + sf.lib.fv3_last_error(None).decode())
            if sf.lib.fv3_ctx_comm_init(sf.ctx, C.byref(ident), 1, 0) != 0:
                raise RuntimeError("fv3_ctx_comm_init failed: " + sf.lib.fv3_last_error(sf.ctx).decode())
            self.transport = "loopback-rccl"
            return

        # FV3_LOOPBACK_DELAY_US: a device-side stall of that many microseconds per update on the stream the exchange runs on -- the
        # transfer time of a real interconnect, to measure how much of it the sequencer hides (tools/overlap_experiment.py)
        delay_us = float(os.environ.get("FV3_LOOPBACK_DELAY_US", "0"))
        on_gpu = torch.device(self.sf.device).type == "cuda"
        spin = 0
        if delay_us > 0 and on_gpu:
            # cycles of torch.cuda._sleep per microsecond, measured (the device properties of this build carry no clock rate)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(self.sf.device)
            torch.cuda._sleep(1000)
            e0.record()
            torch.cuda._sleep(20_000_000)
            e1.record()
            torch.cuda.synchronize(self.sf.device)
            spin = int(delay_us * 20_000_000 / (e0.elapsed_time(e1) * 1.0e3))

        def xfer(_user, plan, phase):
            try:
                up = self._by_plan[int(plan)]
                if int(phase) == 0:
                    h = self.sf.lib.fv3_ctx_get_comm_stream(self.sf.ctx) if on_gpu else None
                    ctx = torch.cuda.stream(torch.cuda.ExternalStream(int(h), device=self.sf.device)) if h else contextlib.nullcontext()
                    with ctx:  # (the library's communication stream: ordered with the plan's pack / unpack kernels)
                        for p in up._peers:
                            rb, sb = up._recv_bufs[p], up._send_bufs[p]
                            if rb is not None and sb is not None and rb.numel() == sb.numel():
                                rb.copy_(sb, non_blocking=True)
                        if spin:
                            torch.cuda._sleep(spin)
                return 0
            except Exception as e:  # never let an exception cross the C frame
                self._xfer_error = e
                return 1

        self._xfer_cb = _lib.fv3_xfer_fn(xfer)
        self.sf.lib.fv3_ctx_set_xfer(self.sf.ctx, self._xfer_cb, None)
        self.transport = "loopback"

    @property
    def transport_name(self) -> str:
        """What moves the messages between processes: 'local' (one process), 'rccl-native' (ncclSend / ncclRecv issued by the
        library), 'gloo-host' (the library's plans, messages moved by torch.distributed on the host) or 'torch' (the Python
        torch.distributed path: FV3_HALO_NATIVE=0, or the fallback)."""
        if self.layout.world_size == 1:
            return "local"
        if self.native:
            return {"rccl": "rccl-native", "host": "gloo-host", "loopback": "loopback (one process alone, messages looped back)",
                    "loopback-rccl": "loopback-rccl (one process alone, messages sent to itself through RCCL)"}.get(self.transport, "torch")
        return "torch"

    # ------------------------------------------------------------------------------------------
    def _maps(self, key, rank) -> GatherMap:
        kind, n_halo = key
        if kind == "sync_dgrid":
            return build_interface_sync_map(self.part, rank, _KIND["dgrid"], self.sf.sizer.n_halo, self.ni)
        return build_halo_map(self.part, rank, _KIND[kind], n_halo, self.sf.sizer.n_halo, self.ni)

    def _phase(self, key, two_d: bool = False) -> _Phase:
        pkey = key + (two_d,)
        if pkey in self._phases:
            return self._phases[pkey]
        lay = self.layout
        sub_stride = self.sk if two_d else self.st
        ph = _Phase()
        ncomp = 2 if key[0] in ("dgrid", "cgrid", "sync_dgrid") else 1
        loc = {(a, b): ([], [], []) for a in range(ncomp) for b in range(ncomp)}
        recv = {}  # peer -> list of (dst_comp, dst_off) in message order
        # my halo points
        for r in lay.local_ranks:
            m = self._maps(key, r)
            dsub = lay.sub_index(r)
            owners = m.src_rank // lay.per_proc
            for e in range(len(m)):
                o = int(owners[e])
                doff = dsub * sub_stride + int(m.dst_flat[e])
                if o == lay.proc:
                    soff = lay.sub_index(int(m.src_rank[e])) * sub_stride + int(m.src_flat[e])
                    d, s_, g = loc[(int(m.dst_comp[e]), int(m.src_comp[e]))]
                    d.append(doff)
                    s_.append(soff)
                    g.append(int(m.sign[e]))
                else:
                    recv.setdefault(o, []).append((int(m.dst_comp[e]), doff))
        for k, (d, s_, g) in loc.items():
            if d:
                ph.local[k] = _Plan(self._sf_proxy, d, s_, g)
        for peer, items in recv.items():
            ph.recv_count[peer] = len(items)
            ph.recv[peer] = {}
            for comp in range(ncomp):
                idx = [i for i, (cc, _) in enumerate(items) if cc == comp]
                if idx:
                    ph.recv[peer][comp] = _Plan(self._sf_proxy, [items[i][1] for i in idx], idx, [1] * len(idx))
        # what the peers need from me: walk their maps in the same (rank, entry) order
        if lay.world_size > 1:
            send = {}
            for peer in range(lay.world_size):
                if peer == lay.proc:
                    continue
                for r in range(peer * lay.per_proc, (peer + 1) * lay.per_proc):
                    m = self._maps(key, r)
                    owners = m.src_rank // lay.per_proc
                    for e in np.nonzero(owners == lay.proc)[0]:
                        soff = lay.sub_index(int(m.src_rank[e])) * sub_stride + int(m.src_flat[e])
                        send.setdefault(peer, []).append((int(m.src_comp[e]), soff, int(m.sign[e])))
            for peer, items in send.items():
                ph.send_count[peer] = len(items)
                ph.send[peer] = {}
                for comp in range(ncomp):
                    idx = [i for i, (cc, _, _) in enumerate(items) if cc == comp]
                    if idx:
                        ph.send[peer][comp] = _Plan(self._sf_proxy, idx, [items[i][1] for i in idx], [items[i][2] for i in idx])
        self._phases[pkey] = ph
        return ph

    def _buffer(self, tag, n, host=False):
        k = (tag, n, host)
        if k not in self._buffers:
            if host:
                self._buffers[k] = torch.empty(n, dtype=self.sf.dtype, pin_memory=True)
            else:
                self._buffers[k] = torch.empty(n, dtype=self.sf.dtype, device=self.sf.device)
        return self._buffers[k]

    # ------------------------------------------------------------------------------------------
    def updater(self, kind: str, groups: Sequence[Sequence[Quantity]], n_halo: int = 3) -> "HaloUpdater":
        """``groups``: for scalars a list of 1-tuples ``[(q,), ...]`` (several fields share one message),
        for vectors a list of pairs ``[(x, y), ...]``."""
        return HaloUpdater(self, (kind, n_halo), groups)

    def synchronize_vector_interfaces(self, u: Quantity, v: Quantity):
        HaloUpdater(self, ("sync_dgrid", 0), [(u, v)]).update()


class HaloUpdater:
    """start() / wait() / update() for a fixed set of quantities (NDSL HaloUpdater)."""

    def __init__(self, ex: HaloExchanger, key, groups):
        self.ex = ex
        self.key = key
        self.groups = [tuple(gp) for gp in groups]
        two_d = self.groups[0][0].is_2d
        if any(q.is_2d != two_d for gp in self.groups for q in gp):
            raise ValueError("a halo updater cannot mix 2-D and 3-D quantities")
        self.ph = ex._phase(key, two_d)
        self.uid = ex._next_uid
        ex._next_uid += 1
        self._inflight = None
        self.nk = 1 if two_d else ex.nk
        self._plan = None  # fv3_halo_plan handle (native updaters)
        self._peers: List[int] = []

    # ------------------------------------------------------------------------------------------
    # native form: the whole update as one fv3_halo_plan
    def _native_plan(self):
        if self._plan is not None:
            return self._plan
        from . import lib as _lib

        ex, ph, sf = self.ex, self.ph, self.ex.sf
        ng, nk = len(self.groups), self.nk
        self._prime = []
        peers = sorted(set(ph.send_count) | set(ph.recv_count))
        pidx = {p: i for i, p in enumerate(peers)}
        ops = []

        def op(kind, plan, dst=None, src=None, dks=0, sks=0, boff=0, bks=0, peer=-1):
            ops.append(_lib.fv3_halo_op(plan.h, dst, src, dks, sks, boff, bks, peer, kind, nk, 0))

        for gi, gp in enumerate(self.groups):
            for (dc, sc), plan in ph.local.items():
                if plan.n:
                    op(_lib.HALO_LOCAL, plan, gp[dc].storage.data_ptr(), gp[sc].storage.data_ptr(), ex.sk, ex.sk)
            for peer, cnt in ph.send_count.items():
                for comp, plan in ph.send[peer].items():
                    op(_lib.HALO_PACK, plan, None, gp[comp].storage.data_ptr(), 0, ex.sk, gi * cnt * nk, cnt, pidx[peer])
            for peer, cnt in ph.recv_count.items():
                for comp, plan in ph.recv[peer].items():
                    op(_lib.HALO_UNPACK, plan, gp[comp].storage.data_ptr(), None, ex.sk, 0, gi * cnt * nk, cnt, pidx[peer])
                    if getattr(ex.layout, "loopback", False) and not ph.send_count.get(peer, 0) and plan.n:
                        # loop-back run, a peer this process only receives from: its message is primed ONCE with what the halo
                        # cells hold before the first update (the unpack plan run backwards), so the update keeps writing finite,
                        # consistent values instead of zeros
                        d, s_, g = plan.offsets
                        self._prime.append((_Plan(sf, s_, d, g), peer, gp[comp].storage.data_ptr(), gi * cnt * nk, cnt))
        plist = []
        self._send_bufs, self._recv_bufs = {}, {}
        self_rccl = ex.transport == "loopback-rccl"
        for p in peers:
            ns, nr = ph.send_count.get(p, 0) * nk * ng, ph.recv_count.get(p, 0) * nk * ng
            sb = ex._buffer(("s", self.uid, p), ns) if ns else None
            rb = ex._buffer(("r", self.uid, p), nr) if nr else None
            self._send_bufs[p], self._recv_bufs[p] = sb, rb
            if self_rccl:
                # every peer is this process (rank 0 of a one-rank communicator); a send to oneself needs its receive of the same
                # size in the same group, so the one-directional messages (primed receive buffers, see above) are not posted
                n = ns if ns == nr else 0
                plist.append(_lib.fv3_halo_peer(0, 0, n, n, sb.data_ptr() if sb is not None else None, rb.data_ptr() if rb is not None else None))
            else:
                plist.append(_lib.fv3_halo_peer(p, 0, ns, nr, sb.data_ptr() if sb is not None else None, rb.data_ptr() if rb is not None else None))
        self._peers = peers
        h = C.c_void_p()
        ops_a = (_lib.fv3_halo_op * max(len(ops), 1))(*ops)
        peers_a = (_lib.fv3_halo_peer * max(len(plist), 1))(*plist)
        st = sf.lib.fv3_halo_plan_create(sf.ctx, C.byref(h), len(ops), ops_a, len(plist), peers_a)
        if st != 0:
            raise RuntimeError("fv3_halo_plan_create failed: " + sf.lib.fv3_last_error(sf.ctx).decode())
        self._plan = h
        ex._by_plan[int(h.value)] = self
        if self._prime:  # (loop-back runs: the one-directional receive buffers are primed when the plan is built)
            for rev, p, fptr, boff, bks in self._prime:
                rb = self._recv_bufs[p]
                rev.run(rb.data_ptr() + boff * rb.element_size(), bks, fptr, ex.sk, self.nk, self._stream())
            self._prime = []
            if torch.device(sf.device).type == "cuda":
                torch.cuda.synchronize(sf.device)
        return h

    def _host_transfer(self, phase: int):
        """Host-driven transport (gloo): move the plan's message buffers between the processes."""
        import torch.distributed as dist

        ex = self.ex
        dev = torch.device(ex.sf.device).type == "cuda"
        if phase == 0:
            if dev:
                torch.cuda.synchronize(ex.sf.device)  # the pack kernels (on the library's stream)
            ops = []
            self._wires = {}
            for p in self._peers:
                rb, sb = self._recv_bufs[p], self._send_bufs[p]
                if rb is not None:
                    wire = ex._buffer(("r", self.uid, p), rb.numel(), host=True) if dev else rb
                    self._wires[p] = wire
                    ops.append(dist.P2POp(dist.irecv, wire, p, group=ex.group))
                if sb is not None:
                    wire = sb
                    if dev:
                        wire = ex._buffer(("s", self.uid, p), sb.numel(), host=True)
                        wire.copy_(sb)
                    ops.append(dist.P2POp(dist.isend, wire, p, group=ex.group))
            self._reqs = dist.batch_isend_irecv(ops) if ops else []
        else:
            for r in self._reqs:
                r.wait()
            if dev:
                for p, wire in self._wires.items():
                    self._recv_bufs[p].copy_(wire)
                torch.cuda.synchronize(ex.sf.device)  # the unpack kernels run on the library's stream
            self._reqs = []

    def _native_call(self, fn, stream_handle):
        ex = self.ex
        sf = ex.sf
        h = stream_handle if stream_handle else self._stream()
        st = fn(sf.ctx, self._native_plan(), h)
        if st != 0:
            err = getattr(ex, "_xfer_error", None)
            if err is not None:
                ex._xfer_error = None
                raise err
            raise RuntimeError("halo plan failed: " + sf.lib.fv3_last_error(sf.ctx).decode())

    def _stream(self):
        return self.ex.sf.stream_handle

    def _on_comm_stream(self):
        """Context manager + raw handle of the stream the exchange runs on."""
        ex = self.ex
        if ex.comm_stream is None:
            import contextlib

            return contextlib.nullcontext(), self._stream()
        return torch.cuda.stream(ex.comm_stream), ex.comm_stream.cuda_stream

    def _compute_stream(self, handle=None):
        """The stream the operators run on, as a torch stream: the handle the C sequencer passes to the halo callback
        (or the factory's stream_handle) -- NOT torch's current stream, which is a different one when the factory was
        given a raw hipStream_t."""
        ex = self.ex
        h = handle if handle else self._stream()
        cur = torch.cuda.current_stream(ex.sf.device)
        if h is None or int(h) == int(cur.cuda_stream):
            return cur
        return torch.cuda.ExternalStream(int(h), device=ex.sf.device)

    def start(self, stream_handle=None):
        ex, ph = self.ex, self.ph
        if ex.native:
            self._native_call(ex.sf.lib.fv3_halo_plan_start, stream_handle)
            self._inflight = True
            return
        if ex.comm_stream is not None:
            # the exchange may start once everything enqueued so far on the compute stream is done
            ex.comm_stream.wait_stream(self._compute_stream(stream_handle))
            ctx, stream = self._on_comm_stream()
        else:
            import contextlib

            ctx, stream = contextlib.nullcontext(), (stream_handle if stream_handle else self._stream())
        with ctx:
            self._start(stream)

    def _start(self, stream):
        ex, ph = self.ex, self.ph
        reqs = []
        if ph.send or ph.recv:
            import torch.distributed as dist

            ops = []
            self._recv_bufs = {}
            for peer, cnt in ph.recv_count.items():
                buf = ex._buffer(("r", self.uid, peer), cnt * self.nk * len(self.groups))
                self._recv_bufs[peer] = buf
                wire = ex._buffer(("r", self.uid, peer), buf.numel(), host=True) if ex.host_staged else buf
                ops.append(dist.P2POp(dist.irecv, wire, peer, group=ex.group))
            for peer, cnt in ph.send_count.items():
                buf = ex._buffer(("s", self.uid, peer), cnt * self.nk * len(self.groups))
                for gi, gp in enumerate(self.groups):
                    base = buf.data_ptr() + gi * cnt * self.nk * buf.element_size()
                    for comp, plan in ph.send[peer].items():
                        plan.run(base, cnt, gp[comp].storage.data_ptr(), ex.sk, self.nk, stream)
                wire = buf
                if ex.host_staged:
                    wire = ex._buffer(("s", self.uid, peer), buf.numel(), host=True)
                    wire.copy_(buf)  # synchronous device -> host copy, ordered after the pack kernels
                ops.append(dist.P2POp(dist.isend, wire, peer, group=ex.group))
            if ops:
                # NCCL/RCCL p2p ops order themselves after the work already enqueued on the
                # current stream (the pack kernels above) and wait() orders the unpack after them.
                reqs = dist.batch_isend_irecv(ops)
        # device-local part (co-resident sub-domains)
        for gp in self.groups:
            for (dc, sc), plan in ph.local.items():
                plan.run(gp[dc].storage.data_ptr(), ex.sk, gp[sc].storage.data_ptr(), ex.sk, self.nk, stream)
        self._inflight = reqs

    def wait(self, stream_handle=None):
        ex = self.ex
        if self._inflight is None:
            return
        if ex.native:
            self._native_call(ex.sf.lib.fv3_halo_plan_wait, stream_handle)
            self._inflight = None
            return
        if ex.comm_stream is not None:
            ctx, stream = self._on_comm_stream()
        else:
            import contextlib

            ctx, stream = contextlib.nullcontext(), (stream_handle if stream_handle else self._stream())
        with ctx:
            self._wait(stream)
        if ex.comm_stream is not None:
            # what follows on the compute stream sees the filled halos
            self._compute_stream(stream_handle).wait_stream(ex.comm_stream)

    def _wait(self, stream):
        ex, ph = self.ex, self.ph
        for r in self._inflight:
            r.wait()
        for peer, cnt in ph.recv_count.items():
            buf = self._recv_bufs[peer]
            if ex.host_staged:
                buf.copy_(ex._buffer(("r", self.uid, peer), buf.numel(), host=True))
            for gi, gp in enumerate(self.groups):
                base = buf.data_ptr() + gi * cnt * self.nk * buf.element_size()
                for comp, plan in ph.recv[peer].items():
                    plan.run(gp[comp].storage.data_ptr(), ex.sk, base, cnt, self.nk, stream)
        self._inflight = None

    def update(self):
        self.start()
        self.wait()
