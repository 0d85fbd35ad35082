"""Run one saved block-pipeline case (npz: m, eps, mp, pic, key) through the C-ABI and compare with the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as O  # noqa: E402
from vtkcloudpoint_amd import _native as N  # noqa: E402

d = np.load(sys.argv[1])
m, eps, mp, pic = np.ascontiguousarray(d["m"]), float(d["eps"]), int(d["mp"]), int(d["pic"])
key = d["key"] if d["key"].size else None
ctx = N.Context(0)
print("running", m.shape, eps, mp, pic, key is not None, flush=True)
g = ctx.dbscan_blocks(m, eps, mp, pic, 3, key_xy=key) if key is not None else ctx.dbscan_blocks(m, eps, mp, pic, 3)
print("gpu done", g["cluster_amount"], g["evals"], flush=True)
o = O.block_pipeline(m, eps, mp, pic, 3, key_xy=key) if key is not None else O.block_pipeline(m, eps, mp, pic, 3)
print("same:", np.array_equal(g["labels"], o["labels"]), np.array_equal(g["order"], o["order"]), g["evals"] == o["evals"], flush=True)
for k in ("rows", "cols", "kept", "del_sum", "cluster_amount", "evals"):
    print(k, g[k], o[k])
print("block_of same", np.array_equal(g["block_of"], o["block_of"]))
bad = np.nonzero(g["labels"] != o["labels"])[0]
print("labels differ at", len(bad), bad[:10], g["labels"][bad[:10]], o["labels"][bad[:10]])
print("gpu label counts", np.bincount(g["labels"])[:5], "oracle", np.bincount(o["labels"])[:5])
if len(bad):
    print("coords of differing", m[bad[:5]], "blocks", g["block_of"][bad[:5]])
