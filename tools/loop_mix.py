#!/usr/bin/env python3
"""Static instruction mix of a kernel's main loop, from the built library (no GPU): disassembles the gfx950 code object embedded
in the .so, finds the kernel whose mangled name contains every given key, takes the largest backward branch as the march loop and
counts its instructions by class -- branches, SALU, VALU, DPP moves, SGPR-spill lane moves, waits.
    python tools/loop_mix.py <lib.so> key [key ...]
    python tools/loop_mix.py --waits <lib.so> key [key ...]     every `s_waitcnt vmcnt` of the loop with the instruction behind it and the positions of
                                                                the loop's vector loads / stores (how far the wait reaches: vmcnt(0 / 1) right before fp64
                                                                arithmetic of the hot path = the step's own prefetch is being waited for)
"""
import os
import re
import struct
import subprocess
import sys
import tempfile
from collections import Counter

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path, arch="gfx950"):
    blob = open(path, "rb").read()
    for m in re.finditer(MAGIC, blob):
        po = m.start()
        (n,) = struct.unpack_from("<Q", blob, po + len(MAGIC))
        off = po + len(MAGIC) + 8
        for _ in range(n):
            o, s, ts = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off : off + ts].decode(errors="replace")
            off += ts
            if arch in triple and s:
                yield blob[po + o : po + o + s]


def kernel_asm(lib, keys):
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
            fn = f.name
        try:
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", fn], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(fn)
        cur, out = None, {}
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1)
                out[cur] = []
            elif cur:
                out[cur].append(line)
        for name, lines in out.items():
            if all(k in name for k in keys):
                yield name, lines


def main_loop(lines):
    ins = []
    for ln in lines:
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", ln)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    addr = {a: i for i, (a, _, _) in enumerate(ins)}
    best = None
    for i, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            off = int(args.split()[0])
            if off > 32767:
                off -= 65536
            tgt = a + 4 + off * 4
            if tgt < a and tgt in addr and (best is None or a - tgt > best[0]):
                best = (a - tgt, addr[tgt], i)
    return ins, (ins[best[1] : best[2] + 1] if best else ins)


def classify(op):
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op in ("v_readlane_b32", "v_writelane_b32"):
        return "sgpr_spill_lane_move"
    if op.endswith("_dpp"):
        return "dpp_move"
    if op.startswith("s_waitcnt") or op == "s_nop":
        return "wait/nop"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu_f64" if "f64" in op else "valu_other"
    return "other"


def waits(lib, keys):
    for name, lines in kernel_asm(lib, keys):
        ins, body = main_loop(lines)
        print(f"{name[:110]}\n  main loop: {len(body)} instructions")
        print("  vector loads at", [i for i, (_, op, a) in enumerate(body) if op.startswith(("global_load", "flat_load"))])
        print("  vector stores at", [i for i, (_, op, a) in enumerate(body) if op.startswith("global_store")])
        for i, (_, op, args) in enumerate(body):
            if op == "s_waitcnt" and "vmcnt" in args:
                n = body[i + 1] if i + 1 < len(body) else ("", "", "")
                hot = "f64" in n[1] or "dpp" in n[1] or n[1].startswith(("v_mov_b64", "ds_", "v_cndmask", "global_store"))
                print(f"  {i:5d}  s_waitcnt {args:24s} -> {n[1]} {n[2][:48]}" + ("   <-- in front of hot-path arithmetic" if hot else ""))


def all_loops(lines):
    """every backward branch of the kernel: (first instruction index, last instruction index), innermost loops included"""
    ins, _ = main_loop(lines)
    addr = {a: i for i, (a, _, _) in enumerate(ins)}
    out = []
    for i, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            off = int(args.split()[0])
            if off > 32767:
                off -= 65536
            tgt = a + 4 + off * 4
            if tgt < a and tgt in addr:
                out.append((addr[tgt], i))
    return ins, out


def loops(lib, keys, dump=None):
    for name, lines in kernel_asm(lib, keys):
        ins, lp = all_loops(lines)
        print(f"{name[:120]}\n  kernel {len(ins)} instructions, {len(lp)} loops")
        for a, b in sorted(lp):
            c = Counter(classify(op) for _, op, _ in ins[a : b + 1])
            print(f"  [{a:5d} .. {b:5d}] {b - a + 1:5d}: " + ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
        if dump:
            with open(dump, "w") as f:
                for i, (a, op, args) in enumerate(ins):
                    f.write(f"{i:6d}  {op} {args}\n")


def main():
    if sys.argv[1] == "--waits":
        return waits(sys.argv[2], sys.argv[3:])
    if sys.argv[1] == "--loops":  # every loop of the kernel with its mix; --dump=<file> also writes the numbered listing
        args = [x for x in sys.argv[2:] if not x.startswith("--dump=")]
        dump = next((x[7:] for x in sys.argv[2:] if x.startswith("--dump=")), None)
        return loops(args[0], args[1:], dump)
    lib, keys = sys.argv[1], sys.argv[2:]
    for name, lines in kernel_asm(lib, keys):
        ins, body = main_loop(lines)
        c = Counter(classify(op) for _, op, _ in body)
        print(f"{name[:120]}\n  kernel {len(ins)} instructions, main loop {len(body)} (static; rare paths included): " + ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
