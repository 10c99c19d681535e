#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [top]

Counter units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section): both counters
are reported in KiB... no -- in units of 1 KB?  rocprofv3 documents FETCH_SIZE / WRITE_SIZE in
kilobytes; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950.  8-B-per-lane f64
accesses are "uncalibrated" there, so the table is calibrated on this library's own fv3_copy
kernel (reads N*8 B, writes N*8 B): the script prints the raw per-launch values of fv3_copy next
to its known byte count and applies that ratio (read_scale / write_scale) to every kernel.
"""
import csv
import os
import re
import sys
from collections import defaultdict


def _src_hash():
    """hash of the kernel sources this run measured (the snapshot the script runs in): bench.py quotes the file only for a library built from them"""
    import os

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pace_amd import build

    return build.src_hash()


def short(name):
    m = re.search(r"fv3_k(?:wg|3n|fr|[23bw])<(?:\d+, )*(.*?)::\{lambda.*?#(\d+)\}", name)
    if m:
        fn = re.sub(r"\(.*\)", "", m.group(1).replace("(anonymous namespace)::", ""))
        return f"{fn}#{m.group(2)}"
    m = re.search(r"fv3_k(?:wg|3n|fr|[23bw])<(?:\d+, )*Z*L?\d*([A-Za-z_0-9]+)\(", name)
    if m:
        return m.group(1)
    return re.sub(r"\(.*", "", name)[:60]


def load(path, counter):
    per = defaultdict(lambda: [0, 0.0, 0.0])  # calls, sum value, sum ns
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        e = per[short(r["Kernel_Name"])]
        e[0] += 1
        e[1] += float(r["Counter_Value"])
        e[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return per


def window(path, counter, first, last):
    """Sum of the counter over the dispatches from the first launch of kernel `first` to the last
    launch of kernel `last` (one operator call, e.g. fxadv .. fv3_d_sw#14 = one fv3_d_sw)."""
    rows = [(int(r["Dispatch_Id"]), short(r["Kernel_Name"]), float(r["Counter_Value"])) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort()
    a = next((d for d, n, _ in rows if n.startswith(first)), None)
    z = max((d for d, n, _ in rows if n.startswith(last)), default=None)
    if a is None or z is None:
        return None, 0
    sel = [v for d, n, v in rows if a <= d <= z]
    return sum(sel), len(sel)


# One acoustic sub-step in launch order (the counter passes profile exactly one: --k-split 1 --n-split 1): an operator's window opens at the first launch whose
# name starts with one of its markers and closes where the next operator's opens; launches of the sequencer's glue (halo gathers, copies, zero launches) and
# of PyTorch's set-up are left out of every window.
OPERATORS = [
    ("c_sw", ("fv3_c_sw", "csw_")),
    ("update_dz_c", ("fv3_update_dz_c",)),
    ("riem_solver_c", ("fv3_riem_solver_c",)),
    ("p_grad_c", ("fv3_p_grad_c",)),
    ("d_sw", ("fxadv",)),
    ("update_dz_d", ("edge_profile", "fv3_update_dz_d")),
    ("riem_solver3", ("fv3_riem_solver3",)),
    ("pk3_halo_edge_pe", ("fv3_edge_pe", "fv3_pk3_halo")),
    ("nh_p_grad", ("nh_pgf", "fv3_nh_p_grad")),
    ("ray_fast", ("fv3_ray_fast",)),
    ("diffusive_heating", ("d2_launch", "fv3_del2", "del2_", "fv3_apply_diffusive")),
]
GLUE = ("fv3_gather_kernel", "copy_frames", "copy_part", "fv3_zero", "zero_unwritten", "fv3_set_gz", "fv3_copy", "void at::", "__amd_rocclr", "void  ", "void rocblas")


def operator_windows(path, counter):
    """{operator: [counter sum, launches, ns]} over the launches of one sub-step, attributed as described above"""
    rows = [(int(r["Dispatch_Id"]), short(r["Kernel_Name"]), float(r["Counter_Value"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort()
    out = {name: [0.0, 0, 0.0] for name, _ in OPERATORS}
    cur, nxt = None, 0
    for _, n, v, ns in rows:
        for q in range(nxt, len(OPERATORS)):
            if n.startswith(OPERATORS[q][1]):
                cur, nxt = OPERATORS[q][0], q + 1
                break
        if cur is None or n.startswith(GLUE):
            continue
        e = out[cur]
        e[0] += v
        e[1] += 1
        e[2] += ns
    return out


def main(fetch_csv, write_csv, top=40, copy_bytes=None, first=None, last=None, json_out=None):
    rd = load(fetch_csv, "FETCH_SIZE")
    wr = load(write_csv, "WRITE_SIZE")
    names = sorted(set(rd) | set(wr), key=lambda n: -(rd.get(n, [0, 0, 0])[2]))
    kb = 1024.0
    rs = ws = 1.0
    copy_gbps = None
    cp = [n for n in names if n.startswith("fv3_copy")]
    if cp and copy_bytes:
        c = cp[0]
        raw_r = rd[c][1] / rd[c][0] * kb
        raw_w = wr[c][1] / wr[c][0] * kb
        rs, ws = copy_bytes / raw_r, copy_bytes / raw_w
        copy_ms = rd[c][2] / rd[c][0] / 1e6
        copy_gbps = 2.0 * copy_bytes / (copy_ms * 1e-3) / 1e9
        print(f"calibration on {c}: known {copy_bytes / 1e6:.1f} MB each way per launch; raw FETCH_SIZE {raw_r / 1e6:.1f} MB (scale {rs:.3f}), raw WRITE_SIZE {raw_w / 1e6:.1f} MB (scale {ws:.3f})\n")
    elif copy_bytes:
        # (the sequencer's default order has no fv3_copy launch any more: the counter passes run with FV3_GZ_FIRST=copy to have one -- tools/collect_profiles.sh)
        print("NO CALIBRATION KERNEL (fv3_copy) in the counter files: raw counter units below, no traffic file written\n")
        json_out = None
    if first and last:
        r, n1 = window(fetch_csv, "FETCH_SIZE", first, last)
        w, n2 = window(write_csv, "WRITE_SIZE", first, last)
        if r is not None and w is not None:
            print(f"operator window {first} .. {last}: {n1} launches, read {r * kb * rs / 1e9:.2f} GB, write {w * kb * ws / 1e9:.2f} GB, "
                  f"total {(r * kb * rs + w * kb * ws) / 1e9:.2f} GB (corrected)\n")
            if json_out:
                import json

                json.dump({"csrc_hash": _src_hash(), "window": [first, last], "launches": n1, "read_bytes": r * kb * rs, "write_bytes": w * kb * ws, "bytes": r * kb * rs + w * kb * ws,
                           "read_scale": rs, "write_scale": ws, "copy_GBps": copy_gbps, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), calibrated on fv3_copy"
                           + ("; " + os.environ["FV3_PMC_NOTE"] if os.environ.get("FV3_PMC_NOTE") else "")},
                          open(json_out, "w"))
    if json_out and copy_gbps:
        # every operator of the sub-step from the same two passes (bench.py: roofline_operators[*].traffic_GB); next to traffic_d_sw.json
        import json

        ro, wo = operator_windows(fetch_csv, "FETCH_SIZE"), operator_windows(write_csv, "WRITE_SIZE")
        ops = {k: {"read_GB": ro[k][0] * kb * rs / 1e9, "write_GB": wo[k][0] * kb * ws / 1e9, "traffic_GB": (ro[k][0] * kb * rs + wo[k][0] * kb * ws) / 1e9,
                   "launches": ro[k][1], "ms_serialized_pmc_pass": ro[k][2] / 1e6} for k in ro if ro[k][1]}
        json.dump({"csrc_hash": _src_hash(), "copy_GBps": copy_gbps, "operators": ops, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of ONE acoustic "
                   "sub-step, calibrated on fv3_copy; windows by launch order (tools/pmc_traffic.py OPERATORS), sequencer glue and halo gathers left out"
                   + ("; " + os.environ["FV3_PMC_NOTE"] if os.environ.get("FV3_PMC_NOTE") else "")},
                  open(os.path.join(os.path.dirname(json_out), "traffic_operators.json"), "w"), indent=1)
        print("| operator | launches | ms (serialized pmc pass) | read GB | write GB | total GB |")
        print("|---|---:|---:|---:|---:|---:|")
        for k, v in ops.items():
            print(f"| {k} | {v['launches']} | {v['ms_serialized_pmc_pass']:.2f} | {v['read_GB']:.2f} | {v['write_GB']:.2f} | {v['traffic_GB']:.2f} |")
        print(f"\nfv3_copy in this pass: {copy_gbps:.0f} GB/s (read + write) -- the measured-copy ceiling `roofline.frac_of_measured_copy` is taken against\n")
    print("| kernel | calls | avg ms (pmc run) | read GB/launch | write GB/launch | GB/s |")
    print("|---|---:|---:|---:|---:|---:|")
    for n in names[:top]:
        c = max(rd.get(n, [0])[0], wr.get(n, [0])[0])
        if not c:
            continue
        r = rd.get(n, [1, 0, 0])
        w = wr.get(n, [1, 0, 0])
        rb = r[1] / max(r[0], 1) * kb * rs
        wb = w[1] / max(w[0], 1) * kb * ws
        ms = r[2] / max(r[0], 1) / 1e6
        print(f"| {n} | {c} | {ms:.3f} | {rb / 1e9:.3f} | {wb / 1e9:.3f} | {(rb + wb) / max(ms, 1e-9) / 1e6:.0f} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40, float(sys.argv[4]) if len(sys.argv) > 4 else None,
         sys.argv[5] if len(sys.argv) > 5 else None, sys.argv[6] if len(sys.argv) > 6 else None, sys.argv[7] if len(sys.argv) > 7 else None)
