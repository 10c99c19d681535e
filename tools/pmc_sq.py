#!/usr/bin/env python3
"""Per-kernel means of SQ counters from a rocprofv3 --pmc counter_collection.csv.

    python tools/pmc_sq.py <counter_collection.csv> [filter-substring] [top]

Prints, per kernel (averaged over its dispatches): duration, waves, VALU / SALU / LDS / VMEM
instructions per wave, and the wave-cycle split (active / wait_any / wait_inst_any) when the
counters were collected.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in quad-cycles
(MI355X_MICROARCH.md, PMC slots).
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"fv3_k(?:wg|3n|fr|[23bw])<(?:\d+, )*(.*?)::\{lambda.*?#(\d+)\}", name)
    if m:
        fn = re.sub(r"\(.*\)", "", m.group(1).replace("(anonymous namespace)::", ""))
        return f"{fn}#{m.group(2)}"
    m = re.search(r"fv3_k(?:wg|3n|fr|[23bw])<(?:\d+, )*Z*L?\d*([A-Za-z_0-9]+)\(", name)
    if m:
        return m.group(1)
    return re.sub(r"\(.*", "", name)[:50]


def main(path, flt="", top=40):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    dur = defaultdict(float)
    for r in csv.DictReader(open(path)):
        n = short(r["Kernel_Name"])
        if flt and flt not in n:
            continue
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[n][r["Counter_Name"]] += 1
        dur[n] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    rows = []
    for n, a in acc.items():
        k = max(cnt[n].values())
        ncounters = len(a)
        m = {c: v / cnt[n][c] for c, v in a.items()}
        ms = dur[n] / (k * ncounters) / 1e6 * 1.0
        rows.append((dur[n] / ncounters, n, k, ms, m))
    rows.sort(reverse=True)
    for _, n, k, ms, m in rows[:top]:
        w = m.get("SQ_WAVES", 0) or 1
        out = [f"{n}: calls {k}, {ms:.3f} ms, waves {w:.0f}"]
        for c, lab in (("SQ_INSTS_VALU", "valu"), ("SQ_INSTS_SALU", "salu"), ("SQ_INSTS_LDS", "lds"), ("SQ_INSTS_VMEM_RD", "vmem_rd"), ("SQ_INSTS_VMEM_WR", "vmem_wr"),
                       ("SQ_INSTS_SMEM", "smem"), ("SQ_INSTS_FLAT", "flat")):
            if c in m:
                out.append(f"{lab}/wave {m[c] / w:.0f}")
        if "SQ_WAVE_CYCLES" in m:
            wc = m["SQ_WAVE_CYCLES"]
            out.append(f"wave_cycles/wave {4 * wc / w:.0f}")
            for c, lab in (("SQ_ACTIVE_INST_ANY", "active"), ("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst"), ("SQ_ACTIVE_INST_VALU", "act_valu"),
                           ("SQ_ACTIVE_INST_LDS", "act_lds"), ("SQ_ACTIVE_INST_VMEM", "act_vmem"), ("SQ_WAIT_INST_LDS", "wait_lds")):
                if c in m:
                    out.append(f"{lab} {100 * m[c] / wc:.0f}%")
        for c in sorted(m):  # instruction-class counters (tools/prof_mix.sh): per wave
            if c.startswith("SQ_INSTS_VALU_") or c in ("SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_SENDMSG", "SQ_INSTS_VSKIPPED"):
                out.append(f"{c[9:].lower()}/wave {m[c] / w:.0f}")
        if "SQ_BUSY_CYCLES" in m:
            out.append(f"busy_cycles {m['SQ_BUSY_CYCLES']:.3g}")
        print(", ".join(out))


def _src_hash():
    import os

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pace_amd import build

    return build.src_hash()


def measured_clock(path, first, last):
    """Shader clock while the operator ran: GRBM_GUI_ACTIVE (busy cycles per dispatch, reported SUMMED over the 8 XCDs of the MI355X -- a
    bandwidth-bound stage kernel reads 19 cycles / ns = 8 x 2.37 GHz, profiles/r05_ablation.md) over the dispatch durations of the same window."""
    rows = [(int(r["Dispatch_Id"]), short(r["Kernel_Name"]), r["Counter_Name"], float(r["Counter_Value"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            for r in csv.DictReader(open(path))]
    rows.sort()
    a = next((d for d, n, _, _, _ in rows if n.startswith(first)), None)
    z = max((d for d, n, _, _, _ in rows if n.startswith(last)), default=None)
    if a is None or z is None:
        return None
    cyc = sum(v for d, n, c, v, _ in rows if a <= d <= z and c == "GRBM_GUI_ACTIVE")
    ns = sum(t for d, n, c, v, t in rows if a <= d <= z and c == "GRBM_GUI_ACTIVE")
    return cyc / ns * 1e9 / 8.0 if ns > 0 and cyc > 0 else None


def valu_window(path, first, last, json_out=None, simds=1024, clock_hz=2.4e9, clock_csv=None):
    """Sum of SQ_INSTS_VALU over the dispatches of one operator call (first launch of kernel `first` .. last launch of kernel `last`)
    = the operator's VALU-issue floor: every VALU wave-instruction occupies its SIMD for 4 cycles, so the call cannot take less
    than  sum(VALU wave-instructions) * 4 / SIMDs / clock  however well memory is overlapped.  SQ_INSTS_VALU is reported summed
    over the waves of a dispatch.  Written to a json the bench line quotes beside the HBM roofline."""
    import json

    rows = [(int(r["Dispatch_Id"]), short(r["Kernel_Name"]), r["Counter_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(path))]
    rows.sort()
    a = next((d for d, n, _, _ in rows if n.startswith(first)), None)
    z = max((d for d, n, _, _ in rows if n.startswith(last)), default=None)
    if a is None or z is None:
        raise SystemExit(f"kernels {first!r} / {last!r} not found in {path}")
    valu = sum(v for d, n, c, v in rows if a <= d <= z and c == "SQ_INSTS_VALU")
    launches = len({d for d, n, c, v in rows if a <= d <= z})
    floor_ms = valu * 4.0 / simds / clock_hz * 1e3
    clk = measured_clock(clock_csv, first, last) if clock_csv else None
    out = {"csrc_hash": _src_hash(), "clock_hz_measured": clk, "valu_floor_ms_measured_clock": (valu * 4.0 / simds / clk * 1e3) if clk else None,
           "window": [first, last], "launches": launches, "valu_wave_instructions": valu, "simds": simds, "clock_hz": clock_hz, "valu_floor_ms": floor_ms,
           "source": "rocprofv3 --pmc SQ_INSTS_VALU (kernel trace only), summed over the launches of one operator call; floor = instructions x 4 cycles / 1024 SIMDs / 2.4 GHz"}
    print(f"operator window {first} .. {last}: {launches} launches, {valu:.4g} VALU wave-instructions -> VALU-issue floor {floor_ms:.2f} ms at {clock_hz / 1e9:.1f} GHz")
    if json_out:
        json.dump(out, open(json_out, "w"))
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--valu-window":
        valu_window(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None, clock_csv=sys.argv[6] if len(sys.argv) > 6 else None)
    else:
        main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", int(sys.argv[3]) if len(sys.argv) > 3 else 40)
