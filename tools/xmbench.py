"""Times engine.xmatch_self on a bench-like candidate table (8 450 rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nemo_amd import engine, catalogs
rng = np.random.default_rng(1)
bra, bdec = rng.uniform(10, 44, 650), rng.uniform(-17, 17, 650)
ra = np.concatenate([bra + rng.normal(0, 0.003, 650) for _ in range(13)])
dec = np.concatenate([bdec + rng.normal(0, 0.003, 650) for _ in range(13)])
snr = rng.uniform(4, 40, len(ra))
engine.xmatch_self(ra, dec, snr, catalogs.XMATCH_RADIUS_DEG)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); engine.xmatch_self(ra, dec, snr, catalogs.XMATCH_RADIUS_DEG); print("xmatch_self %.3f ms" % (1e3 * (time.perf_counter() - t0)))
