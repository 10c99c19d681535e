"""ORACLE (test infrastructure, never the shipped path) -- ctypes wrapper of oracle/remap_oracle.c, the plain-C restatement of
the Lagrangian-to-Eulerian vertical remapping (pyFV3 ``LagrangianToEulerian``; see the header of the C file for the reference
evidence and what is restated).  ``build_lib()`` compiles it with gcc into oracle/_build/ (git-ignored; it travels to the GPU
box like the other built libraries)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(os.path.dirname(_HERE), "remap_oracle.c")
_OUT = os.path.join(os.path.dirname(_HERE), "_build", "libremap_oracle.so")
_lib = None


class _Geom(C.Structure):
    _fields_ = [(n, C.c_int) for n in "ni nj nz nh nx ny".split()] + [(n, C.c_double) for n in "ptop akap rrg t_min".split()]


def build_lib(force: bool = False) -> str:
    if force or not os.path.exists(_OUT) or os.path.getmtime(_OUT) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-std=c99", "-ffp-contract=off", "-shared", "-fPIC", "-o", _OUT, _SRC, "-lm"], check=True)
    return _OUT


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_lib())
        _lib.remap_one.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_double, C.c_int, C.c_double]
        _lib.remap_rank.argtypes = [C.POINTER(_Geom)] + [C.c_void_p] * 15 + [C.c_int, C.c_void_p]
    return _lib


def _p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], "oracle arrays are contiguous float64 [i, j, k]"
    return a.ctypes.data_as(C.c_void_p)


def remap_column(pe1, q1, pe2, iv=1, qs=0.0, qmin: Optional[float] = None):
    """map1_ppm / map_scalar of one column: layer means q1 on interfaces pe1 -> interfaces pe2 (kord 9)."""
    pe1, q1, pe2 = (np.ascontiguousarray(x, dtype=np.float64) for x in (pe1, q1, pe2))
    q2 = np.zeros_like(q1)
    _load().remap_one(len(q1), _p(pe1), _p(q1), _p(pe2), _p(q2), int(iv), float(qs), int(qmin is not None), float(qmin or 0.0))
    return q2


def lagrangian_to_eulerian(D, consts, state: Dict[str, np.ndarray], wsd: np.ndarray, tracers: Optional[List[np.ndarray]] = None, t_min: float = 184.0):
    """One rank, in place: delp, pt, delz, w, u, v, tracers remapped to the Eulerian levels ak + bk * ps; pe, peln, pk, pkz rebuilt;
    returns ps [i, j].  ``state`` holds the oracle's [i, j, k] arrays (cappa read-only); ``wsd`` [i, j, 1] or [i, j]."""
    g = D.grid
    geo = _Geom(state["delp"].shape[0], state["delp"].shape[1], g.nz, g.n_halo, g.nx, g.ny, float(g.ptop), float(consts.KAPPA), float(-consts.RDGAS / consts.GRAV), float(t_min))
    ak, bk = np.ascontiguousarray(g.ak, dtype=np.float64), np.ascontiguousarray(g.bk, dtype=np.float64)
    ps = np.zeros(state["delp"].shape[:2])
    ws = np.ascontiguousarray(np.asarray(wsd, dtype=np.float64).reshape(ps.shape))
    tr = tracers or []
    arr = (C.c_void_p * max(len(tr), 1))(*[t.ctypes.data for t in tr])
    _load().remap_rank(C.byref(geo), _p(ak), _p(bk), *[_p(state[n]) for n in ("delp", "pt", "delz", "w", "u", "v", "cappa", "pe", "peln", "pk", "pkz")], _p(ps), _p(ws),
                       len(tr), arr)
    return ps
