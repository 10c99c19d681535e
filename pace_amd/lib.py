"""ctypes binding of ``include/fv3_mi355x.h``.

The product path loads ``pace_amd/csrc/libfv3_mi355x_f{64,32}.so`` (hipcc, gfx950) and fails
loudly when it is missing -- there is no CPU fallback.  ``load(hostemu=True)`` is for tests
only: it loads the host-emulation build of the same kernel sources (``tests/_hostemu``) so
kernel logic can be compared with the oracle where no GPU exists.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Tuple

from . import build as _build

FV3_MAX_SUB = 32
FV3_F64, FV3_F32 = 0, 1

c_f = C.c_void_p


class fv3_field(C.Structure):
    _fields_ = [
        ("ptr", C.c_void_p),
        ("shape", C.c_int64 * 3),
        ("stride", C.c_int64 * 3),
        ("sub_stride", C.c_int64),
        ("n_sub", C.c_int32),
        ("dtype", C.c_int32),
    ]


class fv3_gridspec(C.Structure):
    _fields_ = [
        ("nx", C.c_int32),
        ("ny", C.c_int32),
        ("nz", C.c_int32),
        ("n_halo", C.c_int32),
        ("n_sub", C.c_int32),
        ("edge_flags", C.c_int32 * FV3_MAX_SUB),
    ]


GRID_PTR_FIELDS = (
    "dx dy dxa dya dxc dyc rdx rdy rdxa rdya rdxc rdyc area rarea area_c rarea_c "
    "cosa sina rsina cosa_u cosa_v cosa_s sina_u sina_v rsin_u rsin_v rsin2 "
    "sin_sg1 sin_sg2 sin_sg3 sin_sg4 cos_sg1 cos_sg2 cos_sg3 cos_sg4 fC f0 "
    "del6_u del6_v divg_u divg_v edge_w edge_e edge_s edge_n"
).split()


class fv3_griddata(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in GRID_PTR_FIELDS] + [
        ("corner_extrap", C.POINTER(C.c_double)),
        ("ak", C.POINTER(C.c_double)),
        ("bk", C.POINTER(C.c_double)),
        ("da_min", C.c_double),
        ("da_min_c", C.c_double),
        ("sin_sg5", C.c_void_p),
    ]


CFG_INT = "n_split k_split hord_dp hord_mt hord_tm hord_vt nord n_sponge do_vort_damp rf_fast hydrostatic use_logp grid_type".split()
CFG_DBL = "a_imp beta p_fac d2_bg d2_bg_k1 d2_bg_k2 d4_bg dddmp d_con d_ext delt_max ke_bg vtdm4 rf_cutoff tau".split()


class fv3_acoustic_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in CFG_INT] + [(n, C.c_double) for n in CFG_DBL]


class fv3_constants(C.Structure):
    _fields_ = [(n, C.c_double) for n in "radius omega grav rdgas rvgas cp_air dz_min pi seconds_per_day".split()]


STATE_FIELDS = "u v w ua va uc vc delp delz pt pe pk peln pkz q_con omga cappa mfxd mfyd cxd cyd diss_estd phis".split()
WORK_FIELDS = "gz zh pkc pk3 crx cry xfx yfx divgd ut vt delpc ptc dsw_delpc heat_source ws3 wsd zs".split()
HALO_UPDATES = "q_con__cappa delp__pt u__v w gz divgd uc__vc delp__pt__q_con zh pkc heat_source interface_u__v".split()
OP_NAMES = "c_sw update_dz_c riem_solver_c p_grad_c d_sw update_dz_d riem_solver3 pk3_halo_edge_pe nh_p_grad ray_fast diffusive_heating glue halo".split()


class fv3_state(C.Structure):
    _fields_ = [(n, fv3_field) for n in STATE_FIELDS]


class fv3_workspace(C.Structure):
    _fields_ = [(n, fv3_field) for n in WORK_FIELDS]


class fv3_halo_op(C.Structure):
    _fields_ = [("plan", C.c_void_p), ("dst", C.c_void_p), ("src", C.c_void_p), ("dst_kstride", C.c_int64), ("src_kstride", C.c_int64), ("buf_off", C.c_int64),
                ("buf_kstride", C.c_int64), ("peer", C.c_int32), ("kind", C.c_int32), ("nk", C.c_int32), ("reserved", C.c_int32)]


class fv3_halo_peer(C.Structure):
    _fields_ = [("rank", C.c_int32), ("reserved", C.c_int32), ("send_elems", C.c_int64), ("recv_elems", C.c_int64), ("send_buf", C.c_void_p), ("recv_buf", C.c_void_p)]


class fv3_nccl_id(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


HALO_LOCAL, HALO_PACK, HALO_UNPACK = 0, 1, 2
# int (*fv3_xfer_fn)(void *user, fv3_halo_plan *plan, int phase)
fv3_xfer_fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)

# int (*fv3_halo_fn)(void *user, int update, int phase, void *stream)
fv3_halo_fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p)

P = C.POINTER
F = P(fv3_field)
_D = C.c_double
_I = C.c_int
_S = C.c_void_p  # stream

_PROTOS = {
    "fv3_version": (C.c_int, []),
    "fv3_backend": (C.c_char_p, []),
    "fv3_build_id": (C.c_char_p, []),
    "fv3_last_error": (C.c_char_p, [C.c_void_p]),
    "fv3_ctx_create": (C.c_int, [P(C.c_void_p), P(fv3_gridspec), P(fv3_griddata), P(fv3_acoustic_config), P(fv3_constants), _I, _I]),
    "fv3_ctx_destroy": (C.c_int, [C.c_void_p]),
    "fv3_ctx_scratch_bytes": (C.c_int64, [C.c_void_p]),
    "fv3_ctx_set_device_sync": (C.c_int, [C.c_void_p, _I]),
    "fv3_c_sw": (C.c_int, [C.c_void_p] + [F] * 15 + [_D, _S]),
    "fv3_update_dz_c": (C.c_int, [C.c_void_p] + [F] * 5 + [_D, _S]),
    "fv3_riem_solver_c": (C.c_int, [C.c_void_p, _D, F, _D] + [F] * 8 + [_S]),
    "fv3_p_grad_c": (C.c_int, [C.c_void_p] + [F] * 5 + [_D, _S]),
    "fv3_fxadv": (C.c_int, [C.c_void_p] + [F] * 8 + [_D, _S]),
    "fv3_fv_tp_2d": (C.c_int, [C.c_void_p] + [F] * 10 + [_I, _I, _D, _S]),
    "fv3_a2b_ord4": (C.c_int, [C.c_void_p, F, F, _I, _I, _I, _S]),
    "fv3_d_sw": (C.c_int, [C.c_void_p] + [F] * 23 + [_D, _S]),
    "fv3_update_dz_d": (C.c_int, [C.c_void_p] + [F] * 7 + [_D, _S]),
    "fv3_riem_solver3": (C.c_int, [C.c_void_p, _I, _D, F, _D] + [F] * 13 + [_S]),
    "fv3_pk3_halo": (C.c_int, [C.c_void_p, F, F, _D, _D, _S]),
    "fv3_edge_pe": (C.c_int, [C.c_void_p, F, F, _D, _S]),
    "fv3_nh_p_grad": (C.c_int, [C.c_void_p] + [F] * 6 + [_D, _D, _D, _S]),
    "fv3_ray_fast": (C.c_int, [C.c_void_p, F, F, F, _D, _D, _S]),
    "fv3_del2_cubed": (C.c_int, [C.c_void_p, F, _D, _I, _S]),
    "fv3_apply_diffusive_heating": (C.c_int, [C.c_void_p] + [F] * 5 + [_D, _S]),
    "fv3_set_gz": (C.c_int, [C.c_void_p, F, F, F, _S]),
    "fv3_copy": (C.c_int, [C.c_void_p, F, F, _S]),
    "fv3_zero": (C.c_int, [C.c_void_p, F, _S]),
    "fv3_compute_geopotential": (C.c_int, [C.c_void_p, F, F, _S]),
    "fv3_acoustic_step": (C.c_int, [C.c_void_p, P(fv3_state), P(fv3_workspace), _D, _I, fv3_halo_fn, C.c_void_p, _S]),
    "fv3_ctx_set_profiling": (C.c_int, [C.c_void_p, _I]),
    "fv3_op_name": (C.c_char_p, [_I]),
    "fv3_profile_read": (C.c_int, [C.c_void_p, P(C.c_double), P(C.c_int64), _I]),
    "fv3_selftest_math": (C.c_int, [C.c_void_p, _I, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, _S]),
    "fv3_gather_plan_create": (C.c_int, [C.c_void_p, P(C.c_void_p), C.c_int64, P(C.c_int64), P(C.c_int64), P(C.c_int8)]),
    "fv3_gather_plan_destroy": (C.c_int, [C.c_void_p]),
    "fv3_gather_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, _I, _S]),
    "fv3_halo_plan_create": (C.c_int, [C.c_void_p, P(C.c_void_p), _I, P(fv3_halo_op), _I, P(fv3_halo_peer)]),
    "fv3_halo_plan_start": (C.c_int, [C.c_void_p, C.c_void_p, _S]),
    "fv3_halo_plan_wait": (C.c_int, [C.c_void_p, C.c_void_p, _S]),
    "fv3_halo_plan_destroy": (C.c_int, [C.c_void_p]),
    "fv3_halo_plan_buffer": (C.c_int, [C.c_void_p, _I, _I, P(C.c_void_p), P(C.c_int64)]),
    "fv3_rccl_available": (C.c_int, []),
    "fv3_ctx_get_comm_stream": (C.c_void_p, [C.c_void_p]),
    "fv3_comm_unique_id": (C.c_int, [P(fv3_nccl_id)]),
    "fv3_ctx_comm_init": (C.c_int, [C.c_void_p, P(fv3_nccl_id), _I, _I]),
    "fv3_ctx_comm_destroy": (C.c_int, [C.c_void_p]),
    "fv3_ctx_set_xfer": (C.c_int, [C.c_void_p, fv3_xfer_fn, C.c_void_p]),
    "fv3_ctx_set_comm_stream": (C.c_int, [C.c_void_p, _I]),
    "fv3_ctx_set_halo_plans": (C.c_int, [C.c_void_p, P(C.c_void_p), _I]),
    "fv3_tracer_2d_1l_cmax": (C.c_int, [C.c_void_p, F, F, P(C.c_double), _S]),
    "fv3_tracer_2d_1l": (C.c_int, [C.c_void_p, _I, P(F), F, F, F, F, F, _I, _I, C.c_void_p, _S]),
    "fv3_remap": (C.c_int, [C.c_void_p, _I, P(F)] + [F] * 13 + [_S]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)

_cache: Dict[Tuple[int, bool], C.CDLL] = {}


class Fv3Error(RuntimeError):
    pass


def load(precision: int = 64, hostemu: bool = False) -> C.CDLL:
    key = (precision, hostemu)
    if key in _cache:
        return _cache[key]
    path = _build.lib_path(precision, hostemu)
    if not os.path.exists(path):
        what = "tests/_hostemu (python -m pace_amd.build --hostemu)" if hostemu else "python -m pace_amd.build (or __graft_entry__.build())"
        raise Fv3Error(f"{path} is missing; build it with {what}. The MI355X path has no CPU fallback.")
    if not hostemu:
        # The device buffers are torch tensors, so this process will hold torch's HIP runtime (the wheel bundles its
        # own libamdhip64).  Load it FIRST: if this library pulled in /opt/rocm's copy before torch, the process would
        # carry two runtimes and fv3_ctx_create fails with "hipSetDevice failed" (seen on the MI355X box when build()
        # and smoke() ran in one process).
        import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in _PROTOS.items():
        try:
            fn = getattr(lib, name)  # AttributeError = missing export
        except AttributeError:
            if os.environ.get("FV3_DEV_PARTIAL") == "1":  # developer builds of a subset of sources
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    backend = lib.fv3_backend().decode()
    if hostemu != (backend == "hostemu"):
        raise Fv3Error(f"{path} reports backend {backend!r}; refusing (product loads hip:gfx950 only, tests ask for hostemu explicitly)")
    _cache[key] = lib
    return lib


def check(lib, ctx, status: int, what: str = ""):
    if status != 0:
        msg = lib.fv3_last_error(ctx)
        raise Fv3Error(f"{what} failed with status {status}: {msg.decode() if msg else ''}")
