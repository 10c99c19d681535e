"""The hand-written fp64 arithmetic and register tables of the wave Riemann solvers, checked directly (C ABI: `fv3_selftest_math`):

* `fv3_div` (the compiler's division sequence without operand scaling / special-value fix-up, csrc/fv3_math.h) against IEEE division,
  bit for bit, over the solvers' operand ranges and over 600 binades;
* the device `log` / `exp` against the host emulation of the same source, bit for bit (the host emulation is pinned against an
  80-bit reference by tests/test_fast_math.py);
* the accumulation-register column (csrc/fv3_agpr.h): every one of the 80 slots of a wave returns what was put there, through
  the one-level and the four-level tables.

The host-emulation run of the same entry point keeps the plumbing covered without a GPU."""
import ctypes as C

import numpy as np
import torch

from helpers import Case


def _factory(backend):
    return Case(nx_tile=12, nz=4, backend=backend).sf


def _run(sf, backend, which, x, y=None):
    n = x.size
    if backend == "hostemu":
        xs, ys, out = np.ascontiguousarray(x), (np.ascontiguousarray(y) if y is not None else None), np.empty_like(x)
        st = sf.lib.fv3_selftest_math(sf.ctx, which, xs.ctypes.data, ys.ctypes.data if ys is not None else None, out.ctypes.data, n, sf.stream_handle)
        assert st == 0, sf.lib.fv3_last_error(sf.ctx).decode()
        return out
    xd = torch.as_tensor(x, device="cuda:0")
    yd = torch.as_tensor(y, device="cuda:0") if y is not None else None
    od = torch.empty_like(xd)
    torch.cuda.synchronize()
    st = sf.lib.fv3_selftest_math(sf.ctx, which, xd.data_ptr(), yd.data_ptr() if yd is not None else None, od.data_ptr(), n, sf.stream_handle)
    assert st == 0, sf.lib.fv3_last_error(sf.ctx).decode()
    torch.cuda.synchronize()
    return od.cpu().numpy()


def _operands(rng, n):
    """(numerator, denominator) pairs: what the solvers divide, plus random signs and 600 binades of both."""
    xs, ys = [], []
    # air masses / heights / pivots of the tridiagonal systems
    xs.append(rng.uniform(1e-2, 2e3, n)), ys.append(-rng.uniform(1.0, 3e3, n))          # -dm / dz
    xs.append(rng.uniform(1e-2, 2e3, n)), ys.append(rng.uniform(1e-2, 2e3, n))          # dm(k-1) / dm(k)
    xs.append(rng.uniform(-1e5, 1e5, n)), ys.append(rng.uniform(2.0, 4.5, n))           # (dd - pp) / bet
    xs.append(np.ones(n)), ys.append(1.0 - rng.uniform(0.2, 0.35, n))                   # 1 / (1 - cappa)
    xs.append(rng.uniform(1.0, 1.1e5, n)), ys.append(np.log(1.0 + rng.uniform(1e-6, 1.0, n)))  # dp / log(p2 / p1)
    # anything normal whose quotient is normal
    ex, ey = rng.integers(-300, 300, n), rng.integers(-300, 300, n)
    xs.append(np.ldexp(rng.uniform(1.0, 2.0, n), ex) * rng.choice([-1.0, 1.0], n)), ys.append(np.ldexp(rng.uniform(1.0, 2.0, n), ey) * rng.choice([-1.0, 1.0], n))
    # exact and nearly exact quotients (ties of the final rounding)
    a = rng.integers(1, 1 << 26, n).astype(np.float64)
    b = rng.integers(1, 1 << 26, n).astype(np.float64)
    xs.append(a * b), ys.append(b)
    xs.append(a), ys.append(b)
    return np.concatenate(xs), np.concatenate(ys)


def test_division_sequence_is_ieee_division(backend):
    sf = _factory(backend)
    rng = np.random.default_rng(20261004)
    x, y = _operands(rng, 200000 if backend != "hostemu" else 2000)
    got = _run(sf, backend, 0, x, y)
    want = x / y
    bad = np.flatnonzero(got.view(np.int64) != want.view(np.int64))
    assert bad.size == 0, (bad.size, x[bad[:3]], y[bad[:3]], got[bad[:3]], want[bad[:3]])


def test_device_log_exp_are_the_host_emulations(backend, hostemu):
    from pace_amd import build

    emu = C.CDLL(build.lib_path(64, hostemu=True))
    for fn in (emu.fv3_hostemu_log, emu.fv3_hostemu_exp):
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        fn.restype = None
    sf = _factory(backend)
    rng = np.random.default_rng(7)
    n = 300000 if backend != "hostemu" else 3000
    xl = np.concatenate([np.exp(rng.uniform(np.log(1e-2), np.log(2e5), n)), 1.0 + rng.uniform(-0.5, 1.0, n), np.exp(rng.uniform(-700.0, 700.0, n))])
    xe = np.concatenate([rng.uniform(-20.0, 20.0, n), rng.uniform(-700.0, 700.0, n)])
    for which, x, ref in ((1, xl, emu.fv3_hostemu_log), (2, xe, emu.fv3_hostemu_exp)):
        x = np.ascontiguousarray(x)
        want = np.empty_like(x)
        ref(x.ctypes.data, want.ctypes.data, x.size)
        got = _run(sf, backend, which, x)
        assert np.array_equal(got.view(np.int64), want.view(np.int64)), which


def test_every_accumulation_register_slot_round_trips(backend):
    sf = _factory(backend)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(80 * 64) * np.exp(rng.uniform(-200.0, 200.0, 80 * 64))  # (every bit pattern matters: low and high words)
    got = _run(sf, backend, 3, x)
    assert np.array_equal(got.view(np.int64), x.view(np.int64))
    # the entry point refuses any other size
    if backend == "hostemu":
        out = np.empty(64)
        assert sf.lib.fv3_selftest_math(sf.ctx, 3, x.ctypes.data, None, out.ctypes.data, 64, sf.stream_handle) != 0
