"""Writes the seeded inputs of oracle_v1.npz as little-endian Float64 binaries for oracle/gen_golden.jl."""
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "inputs")
os.makedirs(out, exist_ok=True)
g = np.load(os.path.join(here, "oracle_v1.npz"))
for k in g.files:
    name, arr = k.split("/")
    if arr in ("X", "U"):
        np.asfortranarray(g[k]).astype("<f8").ravel(order="F").tofile(os.path.join(out, "%s_%s.bin" % (name, arr)))
print("inputs written to", out)
