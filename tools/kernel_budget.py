#!/usr/bin/env python3
"""Register / scratch budget of every kernel in the built HIP library, read from the code-object metadata embedded in the .so
(no GPU needed): name, VGPRs, spilled VGPRs, scratch bytes, SGPRs, LDS.  The marching kernels live at a register count that
decides their occupancy (256 = two waves per SIMD, 128 = four); a compiler or flag change that pushes one over the edge shows
up here before it shows up as a slower bench.
    python tools/kernel_budget.py [libfv3_mi355x_f64.so] [filter]
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernels(lib_path, arch="gfx950"):
    """{mangled kernel name: dict(vgpr, spill, scratch, sgpr, lds)} of the `arch` code objects bundled into lib_path."""
    blob = open(lib_path, "rb").read()
    out = {}
    for m in re.finditer(MAGIC, blob):
        po = m.start()
        (n,) = struct.unpack_from("<Q", blob, po + len(MAGIC))
        off = po + len(MAGIC) + 8
        for _ in range(n):
            o, s, ts = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off : off + ts].decode(errors="replace")
            off += ts
            if arch not in triple or s == 0:
                continue
            with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
                f.write(blob[po + o : po + o + s])
                fn = f.name
            try:
                notes = subprocess.run([READELF, "--notes", fn], capture_output=True, text=True, check=True).stdout
            finally:
                os.unlink(fn)
            for blk in notes.split("  - .agpr_count:")[1:]:
                def g(key, blk=blk):
                    mm = re.search(r"\." + key + r":\s+(\S+)", blk)
                    return mm.group(1) if mm else None

                name = g("name")
                if name:
                    # (.vgpr_count is the unified total: architectural registers rounded up to the accumulation offset + .agpr_count)
                    out[name] = dict(vgpr=int(g("vgpr_count")), agpr=int(blk.split()[0]), spill=int(g("vgpr_spill_count")), scratch=int(g("private_segment_fixed_size")), sgpr=int(g("sgpr_count")),
                                     lds=int(g("group_segment_fixed_size")))
    return out


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "pace_amd", "csrc", "libfv3_mi355x_f64.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    ks = kernels(lib)
    print(f"{len(ks)} kernels in {lib}")
    for name, k in sorted(ks.items(), key=lambda kv: -kv[1]["vgpr"]):
        if flt in name:
            print(f"{k['vgpr']:4d} VGPR ({k['agpr']:3d} acc)  {k['spill']:3d} spilled  {k['scratch']:4d} B scratch  {k['sgpr']:3d} SGPR  {name[:110]}")


if __name__ == "__main__":
    main()
