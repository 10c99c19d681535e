"""Accuracy pin of the range-specific fp64 log / exp the SIM1 Riemann solvers use (pace_amd/csrc/fv3_math.h) against an 80-bit
reference over the solvers' argument range: pressures in Pa (1e-2 .. 2e5), pressure ratios near 1, powers of pressures."""
import ctypes as C

import numpy as np
import pytest

from pace_amd import build


def _hostemu():
    lib = C.CDLL(build.build(64, hostemu=True, verbose=False))
    lib.fv3_hostemu_log.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    lib.fv3_hostemu_log.restype = None
    lib.fv3_hostemu_exp.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    lib.fv3_hostemu_exp.restype = None
    return lib


def _log(lib, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib.fv3_hostemu_log(x.ctypes.data, y.ctypes.data, x.size)
    return y


def _exp(lib, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib.fv3_hostemu_exp(x.ctypes.data, y.ctypes.data, x.size)
    return y


def _ulp_err(y, x):
    """|y - log(x)| in units of the last place of the result, against numpy's long double (80-bit on x86-64)."""
    if np.finfo(np.longdouble).nmant < 63:
        pytest.skip("no extended-precision long double on this platform")
    ref = np.log(x.astype(np.longdouble))
    ulp = np.spacing(np.abs(ref.astype(np.float64)))
    return np.abs((y.astype(np.longdouble) - ref) / ulp.astype(np.longdouble)).astype(np.float64)


def test_fast_log_is_within_one_ulp_over_the_solvers_range():
    lib = _hostemu()
    rng = np.random.default_rng(20261003)
    xs = [
        np.exp(rng.uniform(np.log(1.0e-2), np.log(2.0e5), 400000)),  # pressures: model top (ptop ~ 3 Pa and below) to the surface
        1.0 + rng.uniform(-0.5, 1.0, 400000),                         # ratios of adjacent interface pressures
        1.0 + rng.uniform(-1.0e-3, 1.0e-3, 200000),                   # ... of thin layers (cancellation-prone: f tiny)
        np.exp(rng.uniform(-700.0, 700.0, 200000)),                   # the whole normal range
        np.array([1.0, 2.0, 0.5, np.sqrt(0.5), np.sqrt(2.0), np.nextafter(np.sqrt(0.5), 0), np.nextafter(1.0, 0), np.nextafter(1.0, 2), 2.2250738585072014e-308,
                  1.7976931348623157e308]),
    ]
    worst = 0.0
    for x in xs:
        e = _ulp_err(_log(lib, x), x)
        worst = max(worst, float(e.max()))
        assert e.max() <= 1.0, (float(e.max()), float(x[np.argmax(e)]))
    # and it agrees with the platform libm to the last bit almost everywhere (both are < 1 ulp functions)
    x = xs[0]
    same = np.mean(_log(lib, x) == np.log(x))
    assert same > 0.90, same
    assert _log(lib, np.array([1.0]))[0] == 0.0
    print(f"fast log: worst error {worst:.3f} ulp; bitwise equal to libm on {100 * same:.1f} % of the pressure sample")


def test_fast_log_special_values_are_libms():
    lib = _hostemu()
    with np.errstate(all="ignore"):
        x = np.array([0.0, -1.0, np.inf, np.nan, 5e-324, 1e-310])
        y = _log(lib, x)
        ref = np.log(x)
    assert np.array_equal(np.isnan(y), np.isnan(ref))
    m = ~np.isnan(ref)
    assert np.array_equal(y[m], ref[m])


def test_fast_exp_is_within_one_ulp_over_the_solvers_range():
    if np.finfo(np.longdouble).nmant < 63:
        pytest.skip("no extended-precision long double on this platform")
    lib = _hostemu()
    rng = np.random.default_rng(20261004)
    worst = 0.0
    # gamma * log(p) <= 1.41 * 12.3, (kappa - 1) * log(p) >= -0.72 * 12.3, kappa * log(p) <= 3.6; then the whole supported range
    for lo, hi, n in ((-20.0, 20.0, 600000), (-1.0, 1.0, 200000), (-1.0e-5, 1.0e-5, 100000), (-700.0, 700.0, 300000)):
        x = rng.uniform(lo, hi, n)
        y = _exp(lib, x)
        ref = np.exp(x.astype(np.longdouble))
        ulp = np.spacing(ref.astype(np.float64)).astype(np.longdouble)
        e = np.abs((y.astype(np.longdouble) - ref) / ulp).astype(np.float64)
        worst = max(worst, float(e.max()))
        assert e.max() <= 1.0, (float(e.max()), float(x[np.argmax(e)]))
    assert _exp(lib, np.array([0.0]))[0] == 1.0
    with np.errstate(all="ignore"):
        x = np.array([710.0, -750.0, np.inf, -np.inf, np.nan])  # outside the polynomial's range: libm's answers
        y, ref = _exp(lib, x), np.exp(x)
    assert np.array_equal(np.isnan(y), np.isnan(ref)) and np.array_equal(y[~np.isnan(ref)], ref[~np.isnan(ref)])
    print(f"fast exp: worst error {worst:.3f} ulp")
