"""The C-ABI library loads, exports every symbol include/fv3_mi355x.h declares, and reports errors
through status codes (no compute calls: no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from pace_amd import build, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "fv3_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fv3_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert set(_declared()) == set(lib.EXPORTED_SYMBOLS)


@pytest.mark.parametrize("precision", [64, 32])
def test_hip_library_exports_every_symbol(precision):
    path = build.lib_path(precision)
    if not os.path.exists(path):
        build.build(precision)
    so = C.CDLL(path)
    for name in _declared():
        assert hasattr(so, name), name
    so.fv3_backend.restype = C.c_char_p
    assert so.fv3_backend() == b"hip:gfx950"
    so.fv3_version.restype = C.c_int
    assert so.fv3_version() == 2


def test_product_loader_refuses_hostemu(hostemu, monkeypatch):
    monkeypatch.setattr(build, "lib_path", lambda precision=64, hostemu=False: build.os.path.join(build.HOSTEMU_DIR, f"libfv3_hostemu_f{precision}.so"))
    lib._cache.clear()
    with pytest.raises(lib.Fv3Error):
        lib.load(64, hostemu=False)
    lib._cache.clear()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(build, "lib_path", lambda precision=64, hostemu=False: str(tmp_path / "nope.so"))
    lib._cache.clear()
    with pytest.raises(lib.Fv3Error, match="no CPU fallback"):
        lib.load(64)
    lib._cache.clear()


def test_ctx_create_rejects_unsupported_config(hostemu):
    so = lib.load(64, hostemu=True)
    ctx = C.c_void_p()
    spec = lib.fv3_gridspec(8, 8, 8, 3, 1)
    gd = lib.fv3_griddata()
    cfg = lib.fv3_acoustic_config()
    cfg.hord_dp = cfg.hord_mt = cfg.hord_tm = cfg.hord_vt = 6
    cfg.a_imp = 0.2  # not SIM1
    cst = lib.fv3_constants()
    st = so.fv3_ctx_create(C.byref(ctx), C.byref(spec), C.byref(gd), C.byref(cfg), C.byref(cst), 0, lib.FV3_F64)
    assert st == -3 and b"SIM1" in so.fv3_last_error(None)
    cfg.a_imp = 1.0
    st = so.fv3_ctx_create(C.byref(ctx), C.byref(spec), C.byref(gd), C.byref(cfg), C.byref(cst), 0, lib.FV3_F32)
    assert st == -1  # dtype does not match the f64 build
    st = so.fv3_ctx_create(C.byref(ctx), C.byref(spec), C.byref(gd), C.byref(cfg), C.byref(cst), 0, lib.FV3_F64)
    assert st == -1 and b"null" in so.fv3_last_error(None)  # griddata pointers missing


def test_fp32_build_refuses_overflowing_damping_tables():
    """(da_min_c * d4_bg)^(nord + 1) ~ (6e9)^4 leaves the float range on C48 and coarser grids: the fp32 build must refuse the
    context instead of stepping into NaN."""
    import torch

    from helpers import Case

    build.build(32, hostemu=True, verbose=False)
    with pytest.raises(lib.Fv3Error, match="overflows"):
        Case(12, (1, 1), (0,), nz=4, dtype=torch.float32)
    Case(12, (1, 1), (0,), nz=4, dtype=torch.float32, cfg_kw=dict(nord=1))  # (6e9)^2 fits


def test_python_config_validation():
    from pace_amd.config import AcousticDynamicsConfig

    with pytest.raises(NotImplementedError):
        AcousticDynamicsConfig(hydrostatic=True).validate()
    with pytest.raises(NotImplementedError):
        AcousticDynamicsConfig(hord_tm=8).validate()
    AcousticDynamicsConfig().validate()


def test_field_validation(hostemu):
    import torch

    from helpers import Case

    cs = Case(8, (1, 1), (0,), nz=4)
    good = cs.q()
    bad = cs.qf.zeros(("x", "y"))  # 2-D where 3-D expected
    with pytest.raises(lib.Fv3Error, match="field"):
        cs.sf.call("copy", good.fref, bad.fref)
    other = torch.zeros((1, 5, 15, 16), dtype=torch.float64)
    from pace_amd.quantity import Quantity

    with pytest.raises(lib.Fv3Error, match="shape"):
        cs.sf.call("copy", good.fref, Quantity(other, ("x", "y", "z")).fref)


def test_header_is_plain_c(tmp_path):
    """include/fv3_mi355x.h is the boundary a non-Python host binds (cgo / JNI / Fortran ISO_C_BINDING ...): it must
    compile as plain C99, and a C translation unit that references every declared entry point must link against the
    built library (no C++ or torch types in the signatures)."""
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "fv3_mi355x.h")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
    text = open(hdr).read()
    names = sorted(set(re.findall(r"\b(fv3_[a-z0-9_]+)\s*\(", text)) - {"fv3_halo_fn"})
    names = [n for n in names if re.search(r"^\s*(?:const\s+)?[a-z_0-9 ]+\**\s*\b%s\s*\(" % n, text, re.M)]
    assert len(names) >= 20
    src = tmp_path / "use_all.c"
    body = "\n".join(f"  p[{i}] = (void *)&{n};" for i, n in enumerate(names))
    src.write_text(f'#include "fv3_mi355x.h"\n#include <stdio.h>\nint main(void) {{\n  void *p[{len(names)}];\n{body}\n  printf("%d %p\\n", {len(names)}, p[0]);\n  return 0;\n}}\n')
    obj = tmp_path / "use_all.o"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-c", str(src), "-o", str(obj)], check=True)


def test_context_memory_goes_with_the_last_reference(hostemu):
    """The factory owns the context's device memory (scratch, tables, the second halves of the scalar pairs: tens of GB at C768).
    Dropping the harness must free it at once -- without waiting for the cycle collector (the shared halo exchanger refers back
    to its factory weakly for this reason; a full-size GPU test ran out of memory behind its predecessors before)."""
    import gc
    import weakref

    from pace_amd._testing import hostemu_harness

    gc.collect()
    gc.disable()
    try:
        h = hostemu_harness(12, nz=8, layout=(1, 1), dt_atmos=225.0, k_split=1, n_split=2)
        h.dyn(h.state, 225.0, n_map=1)
        r = weakref.ref(h.sf)
        del h
        assert r() is None
    finally:
        gc.enable()
