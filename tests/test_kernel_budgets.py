"""Register budgets of the product kernels, read from the code-object metadata of the built HIP library (no GPU needed;
tools/kernel_budget.py).  The marching kernels are built around an occupancy -- two waves per SIMD = at most 256 VGPRs, four =
128 -- and none of the hot ones may spill: a change of compiler, flags or source that crosses a limit fails here instead of
surfacing as a slower bench three weeks later."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# (substrings that identify the kernel, VGPR limit, spilled-VGPR limit, scratch-byte limit)
BUDGETS = [
    (("dsw_scalars_tILi1ELi1ELb0ELb1E",), 256, 0, 0),  # delp + w march, interior strips, del-n chain inside: two waves / SIMD
    (("dsw_scalars_tILi2ELi1ELb0ELb1E",), 256, 0, 0),  # q_con + pt march, likewise
    (("dsw_scalars_tILi1ELi1ELb0ELb0E",), 256, 0, 0),  # the same marches without the chain (sponge layers)
    (("dsw_scalars_tILi2ELi1ELb0ELb0E",), 256, 0, 0),
    (("dsw_scalars_tILi2ELi1ELb1ELb0E",), 256, 0, 0),  # tracer pairs, hord 8
    (("tp2d_stream_tILj280E",), 256, 0, 0),            # interface-height transport (area form + del-n chain)
    (("tp2d_stream_tILj288E",), 256, 8, 32),           # vorticity transport + wind epilogue + del-n chain: AT the limit (documented: 4 - 8 spilled)
    (("csw_fused_stream", "fv3_kwgILi2ELi4E"), 256, 0, 0),  # c_sw interior march
    (("nh_pgf_fused", "fv3_kwILi2E"), 256, 0, 0),      # fused nh_p_grad march
    (("ke_stream", "fv3_kwILi4E"), 128, 0, 0),         # corner kinetic energy: four waves / SIMD
    (("fv3_riem_solver_c", "fv3_kwILi1E"), 256, 0, 0),  # wave Riemann solvers: the LDS line, not the registers, sets their occupancy
    (("fv3_riem_solver3", "fv3_kwILi1E", "Lb0E"), 256, 0, 0),
]


@pytest.fixture(scope="module")
def kernel_table():
    from pace_amd import build

    import kernel_budget

    lib = build.lib_path(64)
    if not os.path.exists(lib):
        build.build(64)
    ks = kernel_budget.kernels(lib)
    assert len(ks) > 100, "code-object metadata not found in the library"
    return ks


@pytest.mark.parametrize("keys, vgpr, spill, scratch", BUDGETS, ids=[b[0][0] + ("/" + b[0][-1] if len(b[0]) > 2 else "") for b in BUDGETS])
def test_kernel_stays_inside_its_register_budget(kernel_table, keys, vgpr, spill, scratch):
    hits = {n: k for n, k in kernel_table.items() if all(s in n for s in keys)}
    assert hits, f"no kernel matches {keys}"
    for n, k in hits.items():
        assert k["vgpr"] <= vgpr and k["spill"] <= spill and k["scratch"] <= scratch, f"{n[:100]}: {k} (budget: {vgpr} VGPRs, {spill} spilled, {scratch} B scratch)"
