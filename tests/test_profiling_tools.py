"""The counter post-processing behind the bench line's `roofline.traffic`, `roofline_operators[*].traffic_GB` and `frac_of_measured_copy` (tools/pmc_traffic.py) and
the kernel-trace summaries of profiles/ (tools/summarize_rocprof.py), on synthetic rocprofv3 CSVs: calibration on fv3_copy, the operator windows by launch order with
the sequencer's glue left out, the self-dating hash, the flag on kernels that ran beside others.  No GPU, no profiler."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

K = "void fv3_k3<1, {fn}(fv3_ctx*)::{{lambda(int, int, int, int)#{n}}}>(Box, int, GridMap, {fn}(fv3_ctx*)::{{lambda(int, int, int, int)#{n}}})"
KW = "void fv3_kw<2, (anonymous namespace)::{fn}(fv3_ctx*)::{{lambda(Blk const&, char*)#1}}>(GridMap, (anonymous namespace)::{fn}(fv3_ctx*)::{{lambda(Blk const&, char*)#1}})"

# one sub-step in launch order: (kernel name, FETCH_SIZE KB raw, WRITE_SIZE KB, ns)
STEP = [
    (K.format(fn="fv3_copy", n=1), 1000.0, 2000.0, 700000.0),            # calibration: 2 048 000 B each way -> read scale 2, write scale 1
    (K.format(fn="fv3_c_sw", n=1), 100.0, 50.0, 1000.0),
    (KW.format(fn="csw_fused_stream"), 5000.0, 9000.0, 9000000.0),
    ("fv3_gather_kernel(long, long const*)", 77.0, 77.0, 90000.0),        # halo gather: glue, in nobody's window
    (K.format(fn="fv3_update_dz_c_from", n=1), 3000.0, 2000.0, 2600000.0),
    (K.format(fn="fv3_riem_solver_c", n=1), 8000.0, 7000.0, 8000000.0),
    (K.format(fn="fv3_p_grad_c", n=1), 3000.0, 4000.0, 3400000.0),
    (K.format(fn="fxadv", n=1), 6000.0, 14000.0, 6000000.0),
    (KW.format(fn="wind_stage_march_t<6>"), 8000.0, 7000.0, 5000000.0),
    (K.format(fn="fv3_d_sw_out", n=3), 10.0, 5.0, 50000.0),
    ("fv3_gather_kernel(long, long const*)", 77.0, 77.0, 90000.0),
    (KW.format(fn="edge_profile_wave1<79>"), 1100.0, 2400.0, 1000000.0),
    (K.format(fn="fv3_update_dz_d", n=3), 1100.0, 2300.0, 950000.0),
    (K.format(fn="fv3_riem_solver3", n=1), 17000.0, 21000.0, 9900000.0),
    (KW.format(fn="nh_pgf_fused"), 1900.0, 1000.0, 1200000.0),
    (K.format(fn="fv3_zero", n=1), 0.0, 2300.0, 390000.0),                # glue
]


def _write(path, counter, col):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
        t = 0.0
        for i, row in enumerate(STEP):
            w.writerow([i + 1, row[0], counter, row[col], t, t + row[3]])
            t += row[3] + 10.0


def test_operator_windows_calibration_and_self_dating(tmp_path, capsys):
    import pmc_traffic

    fetch, write, out = str(tmp_path / "f.csv"), str(tmp_path / "w.csv"), str(tmp_path / "traffic_d_sw.json")
    _write(fetch, "FETCH_SIZE", 1)
    _write(write, "WRITE_SIZE", 2)
    pmc_traffic.main(fetch, write, 40, 2048000.0, "fxadv", "fv3_d_sw_out#", out)
    rec = json.load(open(out))
    ops = json.load(open(str(tmp_path / "traffic_operators.json")))
    from pace_amd import build

    assert rec["csrc_hash"] == ops["csrc_hash"] == build.src_hash()
    assert abs(rec["read_scale"] - 2.0) < 1e-12 and abs(rec["write_scale"] - 1.0) < 1e-12
    # d_sw's window: fxadv .. the last fv3_d_sw_out launch (FETCH x 2 x 1024, WRITE x 1024)
    assert rec["launches"] == 3 and abs(rec["read_bytes"] - (6000 + 8000 + 10) * 1024 * 2) < 1 and abs(rec["write_bytes"] - (14000 + 7000 + 5) * 1024) < 1
    assert abs(rec["copy_GBps"] - 2 * 2048000.0 / 0.7e-3 / 1e9) < 1e-9 and ops["copy_GBps"] == rec["copy_GBps"]
    o = ops["operators"]
    assert set(o) == {"c_sw", "update_dz_c", "riem_solver_c", "p_grad_c", "d_sw", "update_dz_d", "riem_solver3", "nh_p_grad"}
    assert o["c_sw"]["launches"] == 2 and o["d_sw"]["launches"] == 3 and o["update_dz_d"]["launches"] == 2  # (the gathers and the zero launch belong to nobody)
    assert abs(o["d_sw"]["traffic_GB"] - rec["bytes"] / 1e9) < 1e-12
    assert abs(o["update_dz_d"]["read_GB"] - 2200 * 1024 * 2 / 1e9) < 1e-12 and abs(o["nh_p_grad"]["write_GB"] - 1000 * 1024 / 1e9) < 1e-12
    assert "operator window fxadv" in capsys.readouterr().out


def test_kernel_summary_flags_the_launches_that_ran_beside_others(tmp_path, capsys):
    import summarize_rocprof

    path = str(tmp_path / "s_kernel_stats.csv")
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs"])
        w.writerow([KW.format(fn="pair_march_t<1>"), 36, 36 * 7.8e6, 7.8e6])
        w.writerow([KW.format(fn="dsw_scalars_t<1, 2, false, false, 0>"), 36, 36 * 6.4e6, 6.4e6])
        w.writerow(["void at::native::vectorized_elementwise_kernel<4>(int)", 100, 1e6, 1e4])
    summarize_rocprof.main(path, 10)
    out = capsys.readouterr().out
    assert "| dsw_scalars_t<1, 2, false, false, 0>#1 † |" in out and "| pair_march_t<1>#1 |" in out  # the sponge-level march on the auxiliary stream is marked
    assert "not additive" in out and "at::native" not in out.split("| kernel")[1]
