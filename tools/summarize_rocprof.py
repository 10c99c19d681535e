#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV into a short table (for profiles/)."""
import csv
import re
import sys


def short(name):
    m = re.search(r"fv3_k(?:wg|3n|fr|[23bw])<(?:\d+, )*(.*?)::\{lambda.*?#(\d+)\}", name)
    if m:
        fn = re.sub(r"\(.*\)", "", m.group(1).replace("(anonymous namespace)::", ""))
        return f"{fn}#{m.group(2)}"
    m = re.search(r"fv3_k(?:wg|3n|fr|[23bw])<(?:\d+, )*Z*L?\d*([A-Za-z_0-9]+)\(", name)
    if m:
        return m.group(1)
    return re.sub(r"\(.*", "", name)[:60]


def is_fv3(name):
    """Kernels of this library (everything else in a bench profile is PyTorch building the grid / the synthetic state,
    outside the timed region)."""
    return bool(re.search(r"fv3_k|fv3_gather_kernel|fv3_kchain", name))


# Launches the sequencer puts on its AUXILIARY stream beside the big marches of the main stream (the sponge-level transports and their del-n chains, the
# cube-corner patch chains): their durations in a kernel trace are elapsed times while SHARING the chip -- a round-4 sponge-level march at one wave per SIMD
# shows 6 - 7 ms beside the marches and 0.13 ms alone (serialized PMC pass) --, so they are not additive with the rows around them.
CONCURRENT = re.compile(r"^(dsw_scalars_t<|tp2d_stream_t<|del6_stream|void fv3_kchain<|fv3_c_sw#[1356])")


def main(path, top=45):
    allrows = list(csv.DictReader(open(path)))
    rows = [r for r in allrows if is_fv3(r["Name"])]
    other = sum(float(r["TotalDurationNs"]) for r in allrows) - sum(float(r["TotalDurationNs"]) for r in rows)
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"library kernels: {tot / 1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches, {len(rows)} distinct kernels "
          f"(set-up kernels of PyTorch -- grid generation, synthetic state; outside the timed region -- excluded: {other / 1e6:.1f} ms)\n")
    print("| kernel (operator#launch) | calls | avg ms | total ms | % |")
    print("|---|---:|---:|---:|---:|")
    conc = 0.0
    for r in rows[:top]:
        nm = short(r["Name"])
        mark = " †" if CONCURRENT.search(nm) else ""
        if mark:
            conc += float(r["TotalDurationNs"])
        print(f"| {nm}{mark} | {r['Calls']} | {float(r['AverageNs']) / 1e6:.3f} | {float(r['TotalDurationNs']) / 1e6:.1f} | {100 * float(r['TotalDurationNs']) / tot:.1f} |")
    print(f"\n† launched on the auxiliary stream BESIDE kernels of the main stream: elapsed time while sharing the chip, not additive with the other rows "
          f"({conc / 1e6:.1f} ms of the {tot / 1e6:.1f} ms above; per-operator sums that add up are the HIP-event timings of the bench line and the serialized PMC pass)")
    agg = {}
    for r in rows:
        op = short(r["Name"]).split("#")[0]
        agg[op] = agg.get(op, 0.0) + float(r["TotalDurationNs"])
    print("\n| operator | total ms | % |\n|---|---:|---:|")
    for op, v in sorted(agg.items(), key=lambda kv: -kv[1])[:20]:
        print(f"| {op} | {v / 1e6:.1f} | {100 * v / tot:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 45)
