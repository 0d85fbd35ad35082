"""cProfile of distributed.exact_slabs at world 1 (host-side overhead of the staged path)."""
import cProfile
import pstats
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtkcloudpoint_amd import _native as N, synth, distributed as D

n = 10_000_000
cloud = synth.config_cloud(n, seed=4)
ctx = N.Context(0)
d = torch.from_numpy(cloud["motor"]).cuda()
for _ in range(3):
    D.exact_slabs(ctx, d, 0.1, 10, 0)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    D.exact_slabs(ctx, d, 0.1, 10, 0)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
