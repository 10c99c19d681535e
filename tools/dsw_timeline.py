#!/usr/bin/env python3
"""Timeline of ONE d_sw call from a rocprofv3 --kernel-trace CSV: every kernel between the end of p_grad_c and the first kernel of update_dz_d of the
chosen sub-step, with its start / end offset from the window's start, its duration and the queue (stream) it ran on -- which chain of launches ends last.

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-op-timing
    python tools/dsw_timeline.py out/.../t_kernel_trace.csv [sub-step index, default 14] [substep: everything up to the end of the next p_grad_c]
"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_rocprof import short  # noqa: E402


def main(path, which=14, whole=False, per=1):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    # d_sw windows: from the end of a p_grad_c kernel to the start of the next edge_profile launch (update_dz_d's first kernel)
    starts = [e for s, e, n, q in rows if "fv3_p_grad_c" in n]
    if len(starts) <= which:
        which = len(starts) - 1
    t0 = starts[which]
    t1 = next((s for s, e, n, q in rows if s > t0 and "edge_profile" in n), rows[-1][1])
    if whole:  # the whole sub-step that starts with this d_sw: up to the end of the next p_grad_c
        t1 = starts[which + per] if which + per < len(starts) else rows[-1][1]  # (per: p_grad_c launches per sub-step -- 2 with the frame-first passes)
    win = [x for x in rows if x[0] >= t0 and x[0] < t1]
    queues = sorted({q for *_, q in win})
    print(f"{'sub-step from the d_sw' if whole else 'd_sw window'} of sub-step {which}: {(t1 - t0) / 1e6:.3f} ms, {len(win)} launches on {len(queues)} queues")
    print("| start ms | end ms | ms | queue | kernel |")
    print("|---:|---:|---:|---|---|")
    for s, e, n, q in win:
        if e - s < 30000:
            continue
        print(f"| {(s - t0) / 1e6:.3f} | {(e - t0) / 1e6:.3f} | {(e - s) / 1e6:.3f} | {queues.index(q)} | {short(n)[:90]} |")
    small = [x for x in win if x[1] - x[0] < 30000]
    print(f"({len(small)} launches under 30 us not listed: {sum(e - s for s, e, *_ in small) / 1e6:.3f} ms)")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 14, len(sys.argv) > 3 and sys.argv[3] == "substep", int(sys.argv[4]) if len(sys.argv) > 4 else 1)
