#!/usr/bin/env python3
"""Multi-process consistency check of the halo-exchange path on GPU(s).

    python tools/multi_gpu_check.py --out gpurun_out/mg_w1.json
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
        tools/multi_gpu_check.py --out gpurun_out/mg_w2.json
    python tools/multi_gpu_check.py --compare gpurun_out/mg_w1.json gpurun_out/mg_w2.json

Runs the same small cube (C48 L8, 2 model steps) on 1 or N processes and writes per-sub-domain sums
of the prognostic fields; the halo exchange is a pure copy, so the sums must be bitwise equal for
every decomposition.  FV3_FORCE_DEVICE=0 puts every process on GPU 0 (a 1-GPU box; RCCL needs
NCCL_IGNORE_DUPLICATE... unsupported there, so --backend gloo stages the messages through the host).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--compare", nargs=2)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--nx", type=int, default=48)
    ap.add_argument("--nz", type=int, default=8)
    ap.add_argument("--layout", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--rtol", type=float, default=0.0, help="--compare: relative tolerance on the sums (0 = bitwise)")
    a = ap.parse_args()
    if a.compare:
        x, y = (json.load(open(p)) for p in a.compare)
        if a.rtol > 0:  # alternative kernel forms: same algorithm, different association of a few sums
            bad = [k for k in x["sums"] if k not in y["sums"] or any(abs(p - q) > a.rtol * max(abs(p), abs(q), 1e-300) for p, q in zip(x["sums"][k][:2], y["sums"][k][:2]))]
        else:
            bad = [k for k in x["sums"] if x["sums"][k] != y["sums"].get(k)]  # (sum, maximum AND the hash of the whole array)
        print(f"world {x['world']} vs {y['world']}: {len(x['sums'])} sums, {len(bad)} differ")
        for k in bad[:10]:
            print("  ", k, x["sums"][k], y["sums"].get(k))
        sys.exit(1 if bad or not x["sums"] else 0)

    from pace_amd.harness import DycoreHarness

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dev = int(os.environ.get("FV3_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (before the first call that initialises the HSA runtime)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist

        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev}"))
        else:
            dist.init_process_group(a.backend)
    h = DycoreHarness(nx_tile=a.nx, nz=a.nz, layout=(a.layout, a.layout), k_split=1, n_split=2, world_size=world, proc=rank, device=f"cuda:{dev}")
    for _ in range(a.steps):
        h.step()
    h.synchronize()
    import hashlib

    sums = {}
    for n in ("delp", "pt", "u", "v", "w", "delz"):
        q = getattr(h.state, n)
        for i, r in enumerate(h.layout.local_ranks):
            v = q.sub(i).view[...][..., : a.nz].double()  # (Quantity.view has no sub-domain axis when a process owns ONE sub-domain)
            # sum and maximum (tolerance comparisons of alternative kernel forms) + a hash of the WHOLE compute-domain array in a fixed
            # (C) order: "bitwise equal" means equal fields, not equal checksums -- a permutation inside a sub-domain changes the hash
            hsh = hashlib.sha256(v.contiguous().cpu().numpy().tobytes()).hexdigest()[:32]
            sums[f"{n}[{r}]"] = (float(v.sum()), float(v.abs().max()), hsh)
    if world > 1:
        import torch.distributed as dist

        gathered = [None] * world
        dist.all_gather_object(gathered, sums)
        sums = {}
        for g in gathered:
            sums.update(g)
        dist.destroy_process_group()
    if rank == 0:
        finite = all(s[0] == s[0] for s in sums.values())
        json.dump({"world": world, "finite": finite, "sums": {k: list(v) for k, v in sorted(sums.items())}}, open(a.out, "w"))
        print(f"world {world}: wrote {len(sums)} sums, finite={finite}")


if __name__ == "__main__":
    main()
