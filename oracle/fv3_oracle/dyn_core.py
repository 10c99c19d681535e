"""ORACLE (test infrastructure) -- the acoustic sub-step sequencer ``AcousticDynamics``
over all ranks of a cubed sphere held in one process, with halo updates done by
index gather  [SURVEY §3.3; FV3 dyn_core.F90; pyFV3 ``dyn_core.AcousticDynamics.__call__``;
called from REF driver/pace/driver/driver.py:641 via DynamicalCore.step_dynamics].
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

from pace_amd.topology import STAGGER, CubedSpherePartitioner, build_halo_map, build_interface_sync_map

from . import c_sw as _c_sw
from . import d_sw as _d_sw
from . import nh as _nh
from .util import Dom, alt

STATE_3D = (
    "u v w ua va uc vc delp delz pt pe pk peln pkz q_con omga cappa mfxd mfyd cxd cyd diss_estd".split()
)


class OracleExchange:
    """In-process halo update for every rank of the cube (numpy gather)."""

    def __init__(self, part: CubedSpherePartitioner, n_halo: int = 3):
        self.part = part
        self.nh = n_halo
        self.ni = part.nx + 2 * n_halo + 1
        self._maps = {}

    def _map(self, key, rank):
        k = (key, rank)
        if k not in self._maps:
            if key == "sync_dgrid":
                m = build_interface_sync_map(self.part, rank, [STAGGER["dgrid_u"], STAGGER["dgrid_v"]], self.nh, self.ni)
            else:
                kind, n_halo = key
                st = {
                    "cell": [STAGGER["cell"]],
                    "corner": [STAGGER["corner"]],
                    "dgrid": [STAGGER["dgrid_u"], STAGGER["dgrid_v"]],
                    "cgrid": [STAGGER["cgrid_u"], STAGGER["cgrid_v"]],
                }[kind]
                m = build_halo_map(self.part, rank, st, n_halo, self.nh, self.ni)
            self._maps[k] = m
        return self._maps[k]

    def _apply(self, key, comps: Sequence[List[np.ndarray]]):
        """comps[c][rank] -> array [i, j, k]; gather from pre-update copies of the sources."""
        nranks = self.part.total_ranks
        ni = self.ni
        updates = []
        for r in range(nranks):
            m = self._map(key, r)
            di, dj = m.dst_flat % ni, m.dst_flat // ni
            si, sj = m.src_flat % ni, m.src_flat // ni
            for c in range(len(comps)):
                sel = m.dst_comp == c
                if not sel.any():
                    continue
                vals = np.empty((int(sel.sum()),) + comps[c][r].shape[2:])
                idx = np.nonzero(sel)[0]
                for sc in range(len(comps)):
                    for sr in np.unique(m.src_rank[idx]):
                        pick = idx[(m.src_comp[idx] == sc) & (m.src_rank[idx] == sr)]
                        if len(pick) == 0:
                            continue
                        pos = np.searchsorted(idx, pick)
                        vals[pos] = comps[sc][sr][si[pick], sj[pick]] * m.sign[pick].astype(np.float64).reshape((-1,) + (1,) * (vals.ndim - 1))
                updates.append((c, r, di[idx], dj[idx], vals))
        for c, r, di, dj, vals in updates:
            comps[c][r][di, dj] = vals

    def scalar(self, fields: List[np.ndarray], kind="cell", n_halo=3):
        self._apply((kind, n_halo), [fields])

    def vector(self, xs: List[np.ndarray], ys: List[np.ndarray], kind="dgrid", n_halo=3):
        self._apply((kind, n_halo), [xs, ys])

    def synchronize_vector_interfaces(self, us, vs):
        self._apply("sync_dgrid", [us, vs])


class OracleAcousticDynamics:
    """All ranks of the cube in one process; ``states[r]`` is a dict of [i, j, k] arrays."""

    def __init__(self, part: CubedSpherePartitioner, grids, cfg, consts, phis: List[np.ndarray]):
        self.part = part
        self.cfg = cfg
        self.c = consts
        self.doms = [Dom(g, consts) for g in grids]
        self.nranks = part.total_ranks
        g0 = grids[0]
        self.nz = g0.nz
        self.ex = OracleExchange(part, g0.n_halo)
        self.col = _d_sw.get_column_namelist(cfg, self.nz)
        self.dp_ref = g0.dp_ref
        self.pfull = g0.pfull
        self.ptop = g0.ptop
        self.akap = consts.KAPPA
        shp = (g0.nx + 2 * g0.n_halo + 1, g0.ny + 2 * g0.n_halo + 1, self.nz + 1)
        self.shape = shp
        names = "gz zh pkc pk3 crx cry xfx yfx divgd ut vt heat_source delpc ptc vt_scratch".split()
        self.tmp = [{n: np.zeros(shp) for n in names} for _ in range(self.nranks)]
        for t in self.tmp:
            t["ws3"] = np.zeros(shp[:2] + (1,))
            t["wsd"] = np.zeros(shp[:2] + (1,))
        self.phis = phis
        self.zs = [p * consts.RGRAV for p in phis]
        self.ex.scalar(self.zs)

    def _each(self, fn):
        for r in range(self.nranks):
            fn(r, self.doms[r], self.tmp[r])

    def __call__(self, states: List[Dict[str, np.ndarray]], timestep: float, n_map: int = 1):
        cfg, c, ex, nz = self.cfg, self.c, self.ex, self.nz
        n_split = cfg.n_split
        dt = timestep / n_split
        dt2 = 0.5 * dt
        end_step = n_map == cfg.k_split
        F = lambda name: [s[name] for s in states]
        T = lambda name: [t[name] for t in self.tmp]
        ex.scalar(F("q_con"))
        ex.scalar(F("cappa"))
        ex.scalar(F("delp"))
        ex.scalar(F("pt"))
        ex.vector(F("u"), F("v"), "dgrid")
        for r in range(self.nranks):
            # every call: the accumulators cover one call, which the tracer advection after it consumes (dyn_core.F90: "Empty the
            # flux capacitors"); n_map only matters for what the caller does with end_step
            for n in ("mfxd", "mfyd", "cxd", "cyd"):
                states[r][n][...] = 0.0
            if not alt("heat_zero_first_call") or n_map == 1:  # (FV3_ALT: DESIGN §2, uncertain restatement 5)
                self.tmp[r]["heat_source"][...] = 0.0
            states[r]["diss_estd"][...] = 0.0
        for it in range(n_split):
            remap_step = it == n_split - 1
            ex.scalar(F("w"))
            if it == 0:
                for r in range(self.nranks):
                    gz = self.tmp[r]["gz"]
                    gz[:, :, nz] = self.zs[r][:, :, 0]
                    for k in range(nz - 1, -1, -1):
                        gz[:, :, k] = gz[:, :, k + 1] - states[r]["delz"][:, :, k]
                ex.scalar(T("gz"))
            for r in range(self.nranks):
                s, t, D = states[r], self.tmp[r], self.doms[r]
                sl = slice(0, nz)
                V = lambda a: a[:, :, sl]
                delpc, ptc = _c_sw.c_sw(D, V(s["delp"]), V(s["pt"]), V(s["u"]), V(s["v"]), V(s["w"]), V(s["uc"]), V(s["vc"]), V(s["ua"]), V(s["va"]), V(t["ut"]), V(t["vt"]), V(t["divgd"]), V(s["omga"]), dt2, nord=cfg.nord)
                t["delpc"][:, :, sl] = delpc
                t["ptc"][:, :, sl] = ptc
            if cfg.nord > 0:
                ex.scalar(T("divgd"), "corner")
            for r in range(self.nranks):
                s, t, D = states[r], self.tmp[r], self.doms[r]
                if it == 0:
                    t["zh"][...] = t["gz"]
                else:
                    t["gz"][...] = t["zh"]
                _nh.update_dz_c(D, self.dp_ref, self.zs[r], t["ut"], t["vt"], t["gz"], t["ws3"], dt2)
                _nh.riem_solver_c(D, dt2, s["cappa"], self.ptop, self.phis[r], t["ws3"], t["ptc"], s["q_con"], t["delpc"], t["gz"], t["pkc"], s["omga"], cfg.p_fac)
                _nh.p_grad_c(D, D.m.rdxc, D.m.rdyc, s["uc"], s["vc"], t["delpc"], t["pkc"], t["gz"], dt2)
            ex.vector(F("uc"), F("vc"), "cgrid")
            for r in range(self.nranks):
                s, t, D = states[r], self.tmp[r], self.doms[r]
                sl = slice(0, nz)
                V = lambda a: a[:, :, sl]
                _d_sw.d_sw(D, cfg, self.col, V(t["vt_scratch"]), V(s["delp"]), V(s["pt"]), V(s["u"]), V(s["v"]), V(s["w"]), V(s["uc"]), V(s["vc"]), V(s["ua"]), V(s["va"]), V(t["divgd"]), V(s["mfxd"]), V(s["mfyd"]), V(s["cxd"]), V(s["cyd"]), V(t["crx"]), V(t["cry"]), V(t["xfx"]), V(t["yfx"]), V(s["q_con"]), t["zh"], V(t["heat_source"]), V(s["diss_estd"]), dt)
            ex.scalar(F("delp"))
            ex.scalar(F("pt"))
            ex.scalar(F("q_con"))
            for r in range(self.nranks):
                s, t, D = states[r], self.tmp[r], self.doms[r]
                _nh.update_dz_d(D, cfg, self.col, self.dp_ref, self.zs[r], t["zh"], t["crx"], t["cry"], t["xfx"], t["yfx"], t["wsd"], dt)
                _nh.riem_solver3(D, remap_step, dt, s["cappa"], self.ptop, self.zs[r], t["wsd"], s["delz"], s["q_con"], s["delp"], s["pt"], t["zh"], s["pe"], t["pkc"], t["pk3"], s["pk"], s["peln"], s["w"], cfg.p_fac)
            ex.scalar(T("zh"))
            ex.scalar(T("pkc"))
            for r in range(self.nranks):
                s, t, D = states[r], self.tmp[r], self.doms[r]
                if remap_step:
                    _nh.pe_halo(D, s["pe"], s["delp"], self.ptop)
                _nh.pk3_halo(D, t["pk3"], s["delp"], self.ptop, self.akap)
                t["gz"][...] = t["zh"] * c.GRAV
                _nh.nh_p_grad(D, s["u"], s["v"], t["pkc"], t["gz"], t["pk3"], s["delp"], dt, self.ptop, self.akap)
                if cfg.rf_fast:
                    _nh.ray_fast(D, cfg, s["u"], s["v"], s["w"], self.dp_ref, self.pfull, dt, self.ptop)
            if it != n_split - 1:
                ex.vector(F("u"), F("v"), "dgrid")
            else:
                ex.synchronize_vector_interfaces(F("u"), F("v"))
        if cfg.d_con > 1.0e-5:
            ex.scalar(T("heat_source"))
            cd = c.CNST_0P20 * self.doms[0].grid.da_min
            for r in range(self.nranks):
                s, t, D = states[r], self.tmp[r], self.doms[r]
                hs = t["heat_source"][:, :, :nz]
                _nh.del2_cubed(D, hs, cd, nmax=min(3, cfg.nord + 1))


                _nh.apply_diffusive_heating(D, s["delp"], s["delz"], s["cappa"], t["heat_source"], s["pt"], abs((timestep if alt("heat_dt_full") else dt) * cfg.delt_max))
