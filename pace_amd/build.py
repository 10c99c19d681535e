"""In-tree build of the HIP library (``libfv3_mi355x_{f64,f32}.so``, gfx950) and of the
test-only host-emulation library (same sources, g++ ``-DFV3_HOST_EMU``).

``python -m pace_amd.build`` builds the product libraries; ``--hostemu`` builds the test
library under ``tests/_hostemu`` (never loaded by the product loader).
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
HOSTEMU_DIR = os.path.join(ROOT, "tests", "_hostemu")

SOURCES = ["fv3_ctx.hip", "fv3_tp2d.hip", "fv3_tp2x.hip", "fv3_tp4.hip", "fv3_tp4x.hip", "fv3_a2b.hip", "fv3_csw.hip", "fv3_dsw.hip", "fv3_wind.hip", "fv3_nh.hip", "fv3_del2x.hip", "fv3_pgf.hip", "fv3_step.hip", "fv3_halo.hip", "fv3_tracer.hip", "fv3_remap.hip"]
HEADERS = ["fv3_common.h", "fv3_ops.h", "fv3_ppm.h", "fv3_a2b.h", "fv3_math.h", "fv3_agpr.h", "fv3_march.h", os.path.join("..", "..", "include", "fv3_mi355x.h")]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]
HOST_FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-ffp-contract=off", "-fopenmp", "-fPIC", "-DFV3_HOST_EMU"]


def _tag() -> str:
    """FV3_LIB_TAG=<name>: a separately named build variant (A/B experiments: built here with FV3_EXTRA_FLAGS / FV3_FLAGS_<stem>,
    selected by the same variable at load time on the GPU box).  Empty = the product library."""
    t = os.environ.get("FV3_LIB_TAG", "")
    return ("." + t) if t else ""


def lib_path(precision: int = 64, hostemu: bool = False) -> str:
    if hostemu:
        return os.path.join(HOSTEMU_DIR, f"libfv3_hostemu_f{precision}.so")
    return os.path.join(CSRC, f"libfv3_mi355x_f{precision}{_tag()}.so")


def _digest(paths, extra):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(repr(extra).encode())
    return h.hexdigest()


def src_hash() -> str:
    """sha256 (first 16 hex digits) of the kernel sources and headers: what a library was built from (embedded as fv3_build_id()) and what a
    counter file was measured on (tools/pmc_traffic.py, tools/pmc_sq.py write it; bench.py compares the two)."""
    paths = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return _digest(paths, None)[:16]


def variant_suffix() -> str:
    """"" for the product build; "+<8 hex>[.<tag>]" when anything changes what the SAME sources compile to: FV3_EXTRA_FLAGS, a per-file FV3_FLAGS_<stem>
    override, or a variant tag (FV3_LIB_TAG).  It is appended to the embedded build id, so a diagnostic library (ablation builds, always-hit metric reads --
    wrong values by design) never matches the counter files bench.py quotes for the product tree, and the bench line shows what it ran on."""
    extra = os.environ.get("FV3_EXTRA_FLAGS", "").split()
    per = sorted((k, v) for k, v in os.environ.items() if k.startswith("FV3_FLAGS_") and v.strip())
    tag = os.environ.get("FV3_LIB_TAG", "")
    if not extra and not per and not tag:
        return ""
    return "+" + hashlib.sha256(repr((extra, per)).encode()).hexdigest()[:8] + (("." + tag) if tag else "")


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)


def build(precision: int = 64, hostemu: bool = False, force: bool = False, verbose: bool = True) -> str:
    out = lib_path(precision, hostemu)
    objdir = os.path.join(HOSTEMU_DIR if hostemu else CSRC, f"_obj_f{precision}{'' if hostemu else _tag()}")
    os.makedirs(objdir, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    real = [] if precision == 64 else ["-DFV3_REAL=float"]
    flags = (HOST_FLAGS if hostemu else HIP_FLAGS) + real + os.environ.get("FV3_EXTRA_FLAGS", "").split()
    cc = os.environ.get("CXX", "g++") if hostemu else HIPCC
    stamp = os.path.join(objdir, "stamp")
    dig = _digest(srcs + hdrs, (flags, sorted((k, v) for k, v in os.environ.items() if k.startswith("FV3_FLAGS_"))))
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == dig:
        return out
    objs = [os.path.join(objdir, os.path.basename(s) + ".o") for s in srcs]

    per_file = {}
    if not hostemu:
        # per-file flag overrides: FV3_FLAGS_<stem>="..." (experiments), e.g. FV3_FLAGS_fv3_nh="-ffp-contract=fast"
        for sname in SOURCES:
            extra = os.environ.get("FV3_FLAGS_" + os.path.splitext(sname)[0])
            if extra:
                per_file[sname] = extra.split()

    idflag = {"fv3_ctx.hip": [f'-DFV3_SRC_HASH="{src_hash()}{"" if hostemu else variant_suffix()}"']}

    def one(i):
        _run([cc] + flags + per_file.get(SOURCES[i], []) + idflag.get(SOURCES[i], []) + ["-c", srcs[i], "-o", objs[i]])

    with cf.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        list(ex.map(one, range(len(srcs))))
    link = [cc, "-shared", "-o", out] + objs
    if hostemu:
        link += ["-fopenmp"]
    else:
        link = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    _run(link)
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print(f"built {out}")
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--hostemu", action="store_true")
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--precision", type=int, nargs="*", default=[64, 32])
    a = ap.parse_args(argv)
    for p in a.precision:
        build(p, hostemu=a.hostemu, force=a.force)


if __name__ == "__main__":
    main()
