"""Register budgets of the product kernels, read from the code-object metadata of the built HIP library (no GPU needed;
tools/kernel_budget.py).  The marching kernels are built around an occupancy -- two waves per SIMD = at most 256 VGPRs, four =
128 -- and none of the hot ones may spill: a change of compiler, flags or source that crosses a limit fails here instead of
surfacing as a slower bench three weeks later."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# The template arguments in the mangled names are spelled from the enums of the sources (fv3_tp4.hip: Q4_AIR = 1, Q4_TRC = 2,
# Q4_INTERIOR = 1; fv3_tp2d.hip: the TF_* feature bits), so a re-numbered enum changes the key here instead of silently matching
# another kernel; every entry also states how many kernels it must match.
Q4_AIR, Q4_TRC, Q4_INTERIOR = 1, 2, 1
TF_EPI, TF_AREA, TF_WIND, TF_FD = 8, 16, 32, 256


def _q4(role, part, m8, fd, hc=0):
    return f"dsw_scalars_tILi{role}ELi{part}ELb{int(m8)}ELb{int(fd)}ELi{hc}E"


def _tp(feat, hc=0, fa=False):
    return f"tp2d_stream_tILj{feat}ELi{hc}ELb{int(fa)}E"


# (substrings that identify the kernel, kernels expected to match, VGPR limit, spilled-VGPR limit, scratch-byte limit)
BUDGETS = [
    ((_q4(Q4_AIR, Q4_INTERIOR, False, True),), 1, 256, 0, 0),   # delp + w march, interior strips, del-n chain inside: two waves / SIMD
    ((_q4(Q4_TRC, Q4_INTERIOR, False, True),), 1, 256, 0, 0),   # q_con + pt march, likewise
    ((_q4(Q4_AIR, Q4_INTERIOR, False, True, 6),), 1, 256, 32, 128),  # ... with the PPM order as a constant (opt-in, FV3_HORD_CONST=1): spills, measured neutral
    ((_q4(Q4_TRC, Q4_INTERIOR, False, True, 6),), 1, 256, 32, 128),
    ((_q4(Q4_AIR, Q4_INTERIOR, False, False),), 1, 256, 0, 0),  # the same marches without the chain (sponge layers)
    ((_q4(Q4_TRC, Q4_INTERIOR, False, False),), 1, 256, 0, 0),
    ((_q4(Q4_TRC, Q4_INTERIOR, True, False),), 1, 256, 0, 0),   # tracer pairs, hord 8
    ((_tp(TF_EPI | TF_AREA | TF_FD, 6, True),), 1, 256, 0, 0),  # interface-height transport, product form (order constant, chain always on)
    ((_tp(TF_WIND | TF_FD, 6, True),), 1, 256, 0, 0),           # vorticity transport + wind epilogue, product form (rolled: no spill)
    ((_tp(TF_EPI | TF_AREA | TF_FD),), 1, 256, 0, 0),           # ... the general forms (run-time order / flags)
    ((_tp(TF_WIND | TF_FD),), 1, 256, 8, 32),                   # AT the limit (documented: 4 - 8 spilled)
    (("csw_fused_stream", "fv3_kwgILi2ELi4E"), 1, 256, 0, 0),   # c_sw interior march
    (("nh_pgf_fused", "fv3_kwILi2E"), 1, 256, 0, 0),            # fused nh_p_grad march
    (("ke_stream_tILb0E", "fv3_kwILi4E"), 1, 128, 0, 0),        # corner kinetic energy: four waves / SIMD
    (("ke_stream_tILb1E", "fv3_kwILi4E"), 1, 128, 0, 0),        # ... forming the cell-mean vorticity too (FV3_DSW_VORT_IN_KE=1, off by default)
    (("fv3_riem_solver_c", "fv3_kwILi1E", "IbLb0E"), 1, 256, 0, 0),       # wave Riemann solvers, gam through the scratch field (FV3_RIEM_REGS=0): the LDS line sets their occupancy
    # (fv3_riem_solver3's kernels are instantiated for (last sub-step, gam in registers, update_dz_d's scan as the pre-sweep); the mangled tags below)
    (("fv3_riem_solver3", "fv3_kwILi1E", "IbLb0EES6_S6_EE"), 1, 256, 0, 0),      # (no, no, no)
    (("fv3_riem_solver3", "fv3_kwILi1E", "IbLb1EES5_IbLb0EES7_EE"), 1, 256, 0, 0),  # (last, no, no)
    # ... the product form: gam in 160 accumulation registers that fv3_agpr.h addresses by hand.  The compiler must not use the
    # accumulation file itself there (it would overwrite the column): exactly 160, architectural <= 256, nothing spilled.
    # (architectural registers strictly BELOW the limit: at 256 the allocator's next register would be a0, i.e. the column)
    (("fv3_riem_solver_c", "fv3_kwILi1E", "IbLb1E"), 1, 240, 0, 0, 160),
    (("fv3_riem_solver3", "fv3_kwILi1E", "IbLb0EES5_IbLb1EES6_EE"), 1, 240, 0, 0, 160),  # (no, registers, no)
    (("fv3_riem_solver3", "fv3_kwILi1E", "IbLb1EES6_S5_IbLb0EEEE"), 1, 240, 0, 0, 160),  # (last, registers, no)
    (("fv3_riem_solver3", "fv3_kwILi1E", "IbLb0EES5_IbLb1EES7_EE"), 1, 240, 0, 0, 160),  # (no, registers, pre-sweep): the product form inside the sequencer
    (("fv3_riem_solver3", "fv3_kwILi1E", "IbLb1EES6_S6_EE"), 1, 240, 0, 0, 160),         # (last, registers, pre-sweep)
    # round-5 marches (fv3_tp4x.hip / fv3_tp2x.hip): two waves per SIMD, nothing spilled in the product forms
    (("pair_march_tILi1E",), 1, 256, 8, 40),     # delp + w (a few spills in the general steps are tolerated: they run 9 of 102 rows)
    (("pair_march_tILi2E",), 1, 256, 8, 40),     # q_con + pt
    # round 6: d_sw's wind stage as one march (fv3_wind.hip): no LDS, nothing spilled, room to spare at two waves per SIMD
    (("wind_stage_march_tILi6E",), 1, 192, 0, 0),   # PPM order 6 as a constant (the product form)
    (("wind_stage_march_tILi0E",), 1, 192, 0, 0),   # run-time order
    # ... the two roles of the pair march as coupled wave pairs (FV3_DSW_MARCH=coupled: measured, NOT the default -- DESIGN §7; the merged kernel spills in the
    #     second role's loop, which is one of the three reasons it lost): pinned so that a change shows
    (("pair_march_tILi3E", "fv3_kwg3ILi2ELi2E"), 1, 256, 160, 320),
    (("single_march_tILi1ELb0E",), 1, 224, 0, 0),   # vorticity transport + winds
    (("single_march_tILi1ELb1E",), 1, 256, 0, 128), # ... with the damping-heat epilogue
    (("single_march_tILi2ELb0E",), 1, 208, 0, 0),   # interface heights
    # the rest of what the reference configurations launch per sub-step (round 6: every kernel of the default path has a row)
    (("edge_profile_wave1ILi79E", "fv3_kwILi2E"), 1, 256, 0, 0),   # update_dz_d's interface interpolation, L79: the column in registers at two waves per SIMD
    ((_q4(Q4_AIR, 2, False, False),), 1, 256, 0, 0, -1),           # sponge levels: the transposed tile-edge marches of round 4 (one wave per SIMD on the auxiliary
    ((_q4(Q4_TRC, 2, False, False),), 1, 256, 0, 0, -1),           # stream: the compiler parks ~80 values in accumulation registers -- any number of them, no memory spill)
    ((_tp(24),), 1, 256, 0, 0),                                    # sponge levels: interface heights (TF_EPI | TF_AREA) ...
    ((_tp(32),), 1, 256, 0, 0),                                    # ... and the vorticity transport (TF_WIND) without the chain
    (("del6_stream", "fv3_kwILi3E"), 1, 168, 0, 0),                # del-n fluxes of the sponge levels: three waves per SIMD
    (("divdamp_stream", "fv3_kwILi4E"), 1, 128, 0, 0),             # damping chain of the levels under the fused wind stage: four waves per SIMD
    (("a2b_ord4_tILi8E", "fv3_kwILi8E"), 2, 64, 0, 0),             # corner interpolation marches (plain store / d_sw's damping epilogue): eight waves per SIMD
    (("fv3_update_dz_c_from", "fv3_k2I", "EUliiiE_E"), 1, 96, 0, 0),  # update_dz_c, one-kernel zh -> gz form: five waves per SIMD
    # the LDS-tile smoothing of the damping-heat tail (fv3_del2x.hip): three workgroups of four waves per CU need <= 170 registers
    (("d2_launchILb0ELb0E",), 1, 168, 0, 0),        # plain tiles, heating as its own launch (the product form)
    (("d2_launchILb1ELb0E",), 1, 256, 0, 0),        # tiles with a cube corner (six LDS offsets per cell; 4 workgroups per sub-domain and level block)
]


@pytest.fixture(scope="module")
def kernel_table():
    import shutil

    from pace_amd import build

    import kernel_budget

    lib = build.lib_path(64)
    if not os.path.exists(kernel_budget.READELF):
        pytest.skip(f"{kernel_budget.READELF} not found (no ROCm LLVM tools on this machine)")
    if not os.path.exists(lib):
        if not (os.path.exists(build.HIPCC) or shutil.which(build.HIPCC)):
            pytest.skip("the HIP library is not built and hipcc is not available")
        build.build(64)
    ks = kernel_budget.kernels(lib)
    if not ks and b"CCOB" in open(lib, "rb").read(1 << 22):
        pytest.skip("compressed offload bundle (--offload-compress): the metadata reader does not unpack it")
    assert len(ks) > 100, "code-object metadata not found in the library"
    return ks


@pytest.mark.parametrize("budget", BUDGETS, ids=[b[0][0] + ("/" + b[0][-1] if len(b[0]) > 2 else "") for b in BUDGETS])
def test_kernel_stays_inside_its_register_budget(kernel_table, budget):
    keys, n_expected, vgpr, spill, scratch = budget[:5]
    agpr = budget[5] if len(budget) > 5 else 0  # accumulation registers: none, except where the source claims them by hand
    hits = {n: k for n, k in kernel_table.items() if all(s in n for s in keys)}
    assert len(hits) == n_expected, f"{keys} matches {len(hits)} kernels, expected {n_expected}: {[n[:90] for n in hits]}"
    for n, k in hits.items():
        arch = k["vgpr"] - k["agpr"]
        assert arch <= vgpr and (agpr < 0 or k["agpr"] == agpr) and k["spill"] <= spill and k["scratch"] <= scratch, (
            f"{n[:100]}: {k} (budget: {vgpr} architectural VGPRs, {agpr} accumulation registers, {spill} spilled, {scratch} B scratch)")


def test_accumulation_registers_are_named_only_inside_the_hand_written_tables():
    """fv3_agpr.h keeps a column of `gam` in a0 .. a159 behind the compiler's back.  In the kernels that do, the disassembly must not name an
    accumulation register anywhere but in the jump tables (the runs of v_accvgpr moves + s_branch right behind an `s_setpc_b64`): an
    allocator that ran out of architectural registers would show up here as a stray `v_accvgpr_*` / `a[..]` operand."""
    import re

    from pace_amd import build

    import loop_mix

    lib = build.lib_path(64)
    if not os.path.exists(loop_mix.OBJDUMP) or not os.path.exists(lib):
        pytest.skip("no ROCm LLVM tools / library on this machine")
    areg = re.compile(r"\ba(\d+|\[\d+:\d+\])")
    checked = 0
    for key in (("fv3_riem_solver_c", "fv3_kwILi1E", "IbLb1E"), ("fv3_riem_solver3", "fv3_kwILi1E", "IbLb0EES5_IbLb1EES6_EE"), ("fv3_riem_solver3", "fv3_kwILi1E", "IbLb1EES6_S5_IbLb0EEEE"),
                ("fv3_riem_solver3", "fv3_kwILi1E", "IbLb0EES5_IbLb1EES7_EE"), ("fv3_riem_solver3", "fv3_kwILi1E", "IbLb1EES6_S6_EE")):
        for name, lines in loop_mix.kernel_asm(lib, key):
            ins, _ = loop_mix.main_loop(lines)
            in_table, n_table, stray = False, 0, []
            for _, op, args in ins:
                if op == "s_setpc_b64":
                    in_table = True
                    continue
                if in_table and op in ("v_accvgpr_write_b32", "v_accvgpr_read_b32", "s_branch"):
                    n_table += op != "s_branch"
                    continue
                in_table = False
                if "accvgpr" in op or areg.search(args):
                    stray.append(f"{op} {args}")
            assert not stray, f"{name[:80]}: accumulation registers named outside the tables: {stray[:5]}"
            assert n_table >= 160, f"{name[:80]}: the tables were not found ({n_table} table moves)"
            checked += 1
    assert checked == 5
