"""Fused general stage vs per-task chain on a deck: python scripts/path_timing.py deck.in [override ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artemis_amd.driver import Simulation
import torch
deck, ov = sys.argv[1], sys.argv[2:]
for path in ("fused", "unfused"):
    s = Simulation(os.path.join(ROOT, "inputs", deck), ov + ["parthenon/time/nlim=70"])
    s.set_path(path)
    s.evolve(10)
    torch.cuda.synchronize()
    t = time.time(); n = s.evolve(50); torch.cuda.synchronize(); w = time.time() - t
    cells = s.nblocks * (s.ie - s.is_ + 1) * (s.je - s.js + 1) * (s.ke - s.ks + 1)
    print(deck, path, "fused" if s.uses_fused_path else "per-task", "tuned" if s.uses_tuned_kernel else "", "cells", cells, "zc/s %.4e" % (cells * n / w), flush=True)
