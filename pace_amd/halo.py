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

import ctypes as C
import os
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
        self.n = int(len(dst_off))
        self.h = C.c_void_p()
        d = np.ascontiguousarray(dst_off, dtype=np.int64)
        s = np.ascontiguousarray(src_off, dtype=np.int64)
        g = np.ascontiguousarray(sign, dtype=np.int8)
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
                self.sf.lib.fv3_gather_plan_destroy(self.h)
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
    """Builds and runs halo updates for the sub-domains of one process."""

    def __init__(self, sf, layout: Layout, group=None, comm_stream=None):
        self.sf = sf
        self.layout = layout
        self.part = layout.part
        self.group = group
        s = sf.sizer
        self.ni, self.nj, self.nk = s.storage_shape
        self.sk = self.ni * self.nj
        self.st = self.sk * self.nk
        self._phases: Dict[Tuple[str, int], _Phase] = {}
        self._buffers: Dict[Tuple, torch.Tensor] = {}
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
        if layout.world_size > 1 and torch.device(sf.device).type == "cuda":
            import torch.distributed as dist

            self.host_staged = dist.get_backend(group) == "gloo"

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
                ph.local[k] = _Plan(self.sf, d, s_, g)
        for peer, items in recv.items():
            ph.recv_count[peer] = len(items)
            ph.recv[peer] = {}
            for comp in range(ncomp):
                idx = [i for i, (cc, _) in enumerate(items) if cc == comp]
                if idx:
                    ph.recv[peer][comp] = _Plan(self.sf, [items[i][1] for i in idx], idx, [1] * len(idx))
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
                        ph.send[peer][comp] = _Plan(self.sf, idx, [items[i][1] for i in idx], [items[i][2] for i in idx])
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
        self._inflight = None
        self.nk = 1 if two_d else ex.nk

    def _stream(self):
        return self.ex.sf.stream_handle

    def _on_comm_stream(self):
        """Context manager + raw handle of the stream the exchange runs on."""
        ex = self.ex
        if ex.comm_stream is None:
            import contextlib

            return contextlib.nullcontext(), self._stream()
        return torch.cuda.stream(ex.comm_stream), ex.comm_stream.cuda_stream

    def start(self):
        ex, ph = self.ex, self.ph
        if ex.comm_stream is not None:
            # the exchange may start once everything enqueued so far on the compute stream is done
            ex.comm_stream.wait_stream(torch.cuda.current_stream(ex.sf.device))
        ctx, stream = self._on_comm_stream()
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
                buf = ex._buffer(("r", id(self), peer), cnt * self.nk * len(self.groups))
                self._recv_bufs[peer] = buf
                wire = ex._buffer(("r", id(self), peer), buf.numel(), host=True) if ex.host_staged else buf
                ops.append(dist.P2POp(dist.irecv, wire, peer, group=ex.group))
            for peer, cnt in ph.send_count.items():
                buf = ex._buffer(("s", id(self), peer), cnt * self.nk * len(self.groups))
                for gi, gp in enumerate(self.groups):
                    base = buf.data_ptr() + gi * cnt * self.nk * buf.element_size()
                    for comp, plan in ph.send[peer].items():
                        plan.run(base, cnt, gp[comp].storage.data_ptr(), ex.sk, self.nk, stream)
                wire = buf
                if ex.host_staged:
                    wire = ex._buffer(("s", id(self), peer), buf.numel(), host=True)
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

    def wait(self):
        ex = self.ex
        if self._inflight is None:
            return
        ctx, stream = self._on_comm_stream()
        with ctx:
            self._wait(stream)
        if ex.comm_stream is not None:
            # what follows on the compute stream sees the filled halos
            torch.cuda.current_stream(ex.sf.device).wait_stream(ex.comm_stream)

    def _wait(self, stream):
        ex, ph = self.ex, self.ph
        for r in self._inflight:
            r.wait()
        for peer, cnt in ph.recv_count.items():
            buf = self._recv_bufs[peer]
            if ex.host_staged:
                buf.copy_(ex._buffer(("r", id(self), peer), buf.numel(), host=True))
            for gi, gp in enumerate(self.groups):
                base = buf.data_ptr() + gi * cnt * self.nk * buf.element_size()
                for comp, plan in ph.recv[peer].items():
                    plan.run(gp[comp].storage.data_ptr(), ex.sk, base, cnt, self.nk, stream)
        self._inflight = None

    def update(self):
        self.start()
        self.wait()
