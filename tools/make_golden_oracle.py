#!/usr/bin/env python3
"""Write tests/golden/oracle_c12_l6_step.npz: outputs of the numpy oracle for one seeded C12 L6
AcousticDynamics call (n_split=2).  The reference cannot run offline, so these vectors pin the
oracle against regressions (not against pyFV3) -- "parity unpinned", see DESIGN.md."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from helpers import compute_slice, oracle_cube  # noqa: E402

nz = 6
part, cfg, grids, st, phis, dyn = oracle_cube(12, (1, 1), nz, dict(n_split=2))
dyn(st, 225.0, 1)
out = {"nz": np.array(nz)}
for r in (0, 2, 5):
    for name in ("delp", "pt", "u", "v", "w", "delz", "q_con", "uc", "vc", "mfxd"):
        out[f"{name}_r{r}"] = st[r][name][compute_slice(name, 12, 12, nz)]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_c12_l6_step.npz"), **out)
print({k: v.shape for k, v in out.items()})
