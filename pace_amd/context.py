"""Device context: GridData / config / constants uploaded once, scratch owned by the C side.

Plays the role of the reference's ``StencilFactory`` for this backend: every operator is
constructed from it, all kernels are built ahead of time and nothing is allocated or compiled
at call time [REF driver/pace/driver/driver.py:761-765; tests/main/fv3core/test_dycore_call.py:193-211].
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import lib as _lib
from .config import AcousticDynamicsConfig
from .constants import ConstantSet, get_constants
from .grid import GridData
from .quantity import GridSizer, Quantity, QuantityFactory


class StencilFactory:
    """Owns the ``fv3_ctx`` for a set of co-resident sub-domains."""

    def __init__(
        self,
        grids: Sequence[GridData],
        config: AcousticDynamicsConfig,
        constants: Optional[ConstantSet] = None,
        backend: str = "hip:gfx950",
        device: Optional[str] = None,
        dtype=torch.float64,
        stream: Optional[int] = None,
        _testing_token=None,
    ):
        config.validate()
        from . import _testing

        if backend == "hostemu":
            # (the host emulation is test infrastructure: pace_amd._testing.hostemu_factory)
            if _testing_token is not _testing._TOKEN:
                raise ValueError("backend 'hostemu' is not part of the product surface: this package runs on 'hip:gfx950' only (tests: pace_amd._testing)")
        elif backend not in ("hip:gfx950", "hip"):
            raise ValueError(f"backend {backend!r}: this build provides 'hip:gfx950'")
        self.backend = backend
        self.hostemu = backend == "hostemu"
        self.config = config
        self.constants = constants or get_constants()
        self.grids = list(grids)
        g0 = self.grids[0]
        self.sizer = GridSizer(g0.nx, g0.ny, g0.nz, g0.n_halo, len(self.grids))
        self.dtype = dtype
        precision = 64 if dtype == torch.float64 else 32
        self.lib = _lib.load(precision, hostemu=self.hostemu)
        if self.hostemu:
            self.device = torch.device("cpu")
        else:
            if not torch.cuda.is_available():
                raise _lib.Fv3Error("backend 'hip:gfx950' needs a GPU; there is no CPU fallback on the product path")
            self.device = torch.device(device or "cuda:0")
        self.quantity_factory = QuantityFactory(self.sizer, self.device, dtype)
        self._keep = []
        self.grid_fields = {}
        self._ctx = C.c_void_p()
        self._create()
        self.stream = stream

    # ------------------------------------------------------------------------------------------
    def _upload2d(self, name):
        ni, nj, _ = self.sizer.storage_shape
        host = np.stack([np.ascontiguousarray(g.fields[name].T) for g in self.grids])  # [n_sub, nj, ni]
        t = torch.from_numpy(host).to(self.dtype).to(self.device).contiguous()
        self.grid_fields[name] = Quantity(t, ("x", "y"), n_halo=self.sizer.n_halo)
        return t

    def _upload1d(self, arrs):
        t = torch.from_numpy(np.stack(arrs)).to(self.dtype).to(self.device).contiguous()
        self._keep.append(t)
        return t

    def _create(self):
        s = self.sizer
        spec = _lib.fv3_gridspec()
        spec.nx, spec.ny, spec.nz, spec.n_halo, spec.n_sub = s.nx, s.ny, s.nz, s.n_halo, s.n_sub
        for t, g in enumerate(self.grids):
            spec.edge_flags[t] = (1 if g.west_edge else 0) | (2 if g.east_edge else 0) | (4 if g.south_edge else 0) | (8 if g.north_edge else 0)
        gd = _lib.fv3_griddata()
        for name in _lib.GRID_PTR_FIELDS:
            if name.startswith("edge_"):
                t = self._upload1d([getattr(g, name) for g in self.grids])
            else:
                t = self._upload2d(name)
            setattr(gd, name, t.data_ptr())
        if "sin_sg5" in self.grids[0].fields:
            gd.sin_sg5 = self._upload2d("sin_sg5").data_ptr()
        ce = np.ascontiguousarray(np.stack([g.corner_extrap for g in self.grids]), dtype=np.float64)
        ak = np.ascontiguousarray(self.grids[0].ak, dtype=np.float64)
        bk = np.ascontiguousarray(self.grids[0].bk, dtype=np.float64)
        self._keep += [ce, ak, bk]
        gd.corner_extrap = ce.ctypes.data_as(C.POINTER(C.c_double))
        gd.ak = ak.ctypes.data_as(C.POINTER(C.c_double))
        gd.bk = bk.ctypes.data_as(C.POINTER(C.c_double))
        gd.da_min = self.grids[0].da_min
        gd.da_min_c = self.grids[0].da_min_c
        cfg = _lib.fv3_acoustic_config()
        for n in _lib.CFG_INT:
            setattr(cfg, n, int(getattr(self.config, n)))
        for n in _lib.CFG_DBL:
            setattr(cfg, n, float(getattr(self.config, n)))
        c = self.constants
        cst = _lib.fv3_constants(c.RADIUS, c.OMEGA, c.GRAV, c.RDGAS, c.RVGAS, c.CP_AIR, c.DZ_MIN, c.PI, c.SECONDS_PER_DAY)
        dev_index = 0 if self.hostemu else (self.device.index or 0)
        dtype_code = _lib.FV3_F64 if self.dtype == torch.float64 else _lib.FV3_F32
        st = self.lib.fv3_ctx_create(C.byref(self._ctx), C.byref(spec), C.byref(gd), C.byref(cfg), C.byref(cst), dev_index, dtype_code)
        if st != 0:
            raise _lib.Fv3Error(f"fv3_ctx_create failed ({st}): {self.lib.fv3_last_error(None).decode()}")

    # ------------------------------------------------------------------------------------------
    @property
    def ctx(self):
        return self._ctx

    @property
    def stream_handle(self):
        if self.hostemu:
            return None
        if self.stream is not None:
            return self.stream
        return torch.cuda.current_stream(self.device).cuda_stream

    def call(self, name, *args):
        """Invoke ``fv3_<name>(ctx, *args, stream)`` and raise on a non-zero status."""
        fn = getattr(self.lib, "fv3_" + name)
        st = fn(self._ctx, *args, self.stream_handle)
        if st != 0:
            raise _lib.Fv3Error(f"fv3_{name} failed ({st}): {self.lib.fv3_last_error(self._ctx).decode()}")

    def set_profiling(self, on):
        """HIP-event pairs around every operator of ``fv3_acoustic_step`` (read with ``profile()``); ``on == 2``: around d_sw only."""
        self.lib.fv3_ctx_set_profiling(self._ctx, int(on))

    def profile(self, reset: bool = True):
        """{operator: (total ms, calls)} accumulated since the last reset; waits for the recorded events."""
        n = len(_lib.OP_NAMES)
        ms = (C.c_double * n)()
        calls = (C.c_int64 * n)()
        st = self.lib.fv3_profile_read(self._ctx, ms, calls, int(reset))
        if st != 0:
            raise _lib.Fv3Error(f"fv3_profile_read failed ({st})")
        return {name: (float(ms[i]), int(calls[i])) for i, name in enumerate(_lib.OP_NAMES) if calls[i]}

    def set_device_sync(self, on: bool):
        self.lib.fv3_ctx_set_device_sync(self._ctx, int(on))

    @property
    def scratch_bytes(self) -> int:
        return int(self.lib.fv3_ctx_scratch_bytes(self._ctx))

    def close(self):
        if self._ctx:
            self.lib.fv3_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
