#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

The reference (C# / .NET 3.5) cannot run here and ships no fixtures, so these are REGRESSION vectors of
the oracle itself (oracle/vcp_oracle.cpp, literal transcription) on the deterministic synthetic clouds of
vtkcloudpoint_amd/synth.py -- they pin the oracle against accidental change; the hand-derived
known-answer cases live in micro_cases.json.  Inputs are not stored (they are regenerated from the seed);
a checksum of the inputs is.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as O  # noqa: E402
from vtkcloudpoint_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    d = synth.config_c1()
    l1 = O.dbscan(d["motor"], d["eps_l1"], d["min_pts"], O.L1_2D, literal=True)
    l2 = O.dbscan(d["xyz"], d["eps_l2"], d["min_pts"], O.L2_3D, literal=True)
    bp = O.block_pipeline(d["motor"], 0.3, 5, 200, 3, canonical=False, brute=True)
    K = bp["cluster_amount"]
    c3, c2, cnt = O.centroids(d["xyz"], d["motor"], bp["labels"], K, bp["order"])
    np.savez_compressed(os.path.join(HERE, "c1_dbscan.npz"), motor_sha=sha(d["motor"]), xyz_sha=sha(d["xyz"]),
                        l1_labels=l1["labels"], l1_key=l1["is_key"], l1_cf=l1["cf"], l1_evals=l1["evals"],
                        l2_labels=l2["labels"], l2_key=l2["is_key"], l2_cf=l2["cf"], l2_evals=l2["evals"],
                        bp_labels=bp["labels"], bp_block_of=bp["block_of"], bp_order=bp["order"],
                        bp_meta=np.array([bp["rows"], bp["cols"], bp["kept"], bp["del_sum"], bp["cluster_amount"]]),
                        bp_evals=bp["evals"], c3=c3, c2=c2, counts=cnt)
    c = synth.config_icp(nd=5000, nm=100, jitter=0.05)
    r = O.icp(c["model"], c["data"], 1e-4, 100, O.STOP_SSE_DELTA)
    r0 = O.icp(synth.config_icp(nd=5000, nm=100, jitter=0.0)["model"], synth.config_icp(nd=5000, nm=100, jitter=0.0)["data"],
               1e-4, 100, O.STOP_RMSE)
    np.savez_compressed(os.path.join(HERE, "icp_5k.npz"), model_sha=sha(c["model"]), data_sha=sha(c["data"]),
                        R=r["R"], T=r["T"], sse=r["sse"], rmse=r["rmse"], iters=r["iters"],
                        nn0=O.find_closest(c["model"], c["data"]), R0=r0["R"], T0=r0["T"], iters0=r0["iters"])
    print("golden written")


if __name__ == "__main__":
    main()
