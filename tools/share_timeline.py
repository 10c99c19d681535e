#!/usr/bin/env python3
"""Kernel timeline of a bench run from a rocprofv3 --kernel-trace CSV: launches per acoustic sub-step, how many are short, and how much of the wall time no
kernel of the library is running (the gaps a HIP-graph replay could close) -- VERDICT round 5, item 5.

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 bench.py --emulate-share 8 --steps 3 --warmup 1 --no-cpu-baseline --no-op-timing
    python tools/share_timeline.py out/.../t_kernel_trace.csv <sub-steps in the run: (steps + warmup) * 12>
"""
import csv
import re
import sys


def main(path, n_sub):
    rows = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if not re.search(r"fv3_k|fv3_gather_kernel|fv3_kchain", name):
            continue  # (PyTorch's set-up kernels: grid generation, synthetic state)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    if not rows:
        print("no library kernels in the trace")
        return
    # the run = from the first c_sw launch to the last kernel; union of the busy intervals (two streams overlap)
    t0 = next((s for s, e, n in rows if "c_sw" in n or "csw_" in n), rows[0][0])
    rows = [x for x in rows if x[0] >= t0]
    t1 = max(e for s, e, n in rows)
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    gaps = []
    for s, e, n in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    wall = t1 - t0
    dur = [e - s for s, e, n in rows]
    short = [d for d in dur if d < 20000]
    print(f"library launches: {len(rows)} over {n_sub} sub-steps = {len(rows) / n_sub:.1f} per sub-step; wall {wall / 1e6:.2f} ms = {wall / 1e6 / n_sub:.3f} ms per sub-step")
    print(f"some kernel of the library running: {busy / 1e6:.2f} ms ({100.0 * busy / wall:.1f} % of the wall time); idle between kernels: {sum(gaps) / 1e6:.2f} ms in {len(gaps)} gaps "
          f"= {sum(gaps) / 1e6 / n_sub:.3f} ms per sub-step (median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.1f} us, largest {max(gaps) / 1e3:.1f} us)" if gaps else "no gaps")
    print(f"launches shorter than 20 us: {len(short)} ({len(short) / n_sub:.1f} per sub-step), {sum(short) / 1e6:.3f} ms in all = {sum(short) / 1e6 / n_sub:.4f} ms per sub-step")
    print(f"sum of the kernel durations {sum(dur) / 1e6:.2f} ms = {sum(dur) / wall:.2f} x the wall time (two streams)")
    # who the short launches are (by kernel, per sub-step), and the kernels by total time
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from summarize_rocprof import short as short_name
    by = {}
    for s_, e, n in rows:
        k = short_name(n)[:70]
        c = by.setdefault(k, [0, 0, 0])
        c[0] += 1
        c[1] += e - s_
        c[2] += 1 if e - s_ < 20000 else 0
    print("| kernel | launches per sub-step | of them < 20 us | ms per sub-step |")
    print("|---|---:|---:|---:|")
    for k, c in sorted(by.items(), key=lambda kv: -kv[1][0])[:28]:
        print(f"| {k} | {c[0] / n_sub:.1f} | {c[2] / n_sub:.1f} | {c[1] / 1e6 / n_sub:.3f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
