"""Runs nf_noise_map alone on a 4096 x 4096 float64 map (40' grid at 0.5' pixels, 1 bin) - for rocprofv3 counter passes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nemo_amd import engine
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng = engine.TileEngine.get((H, W))
rng = np.random.default_rng(1)
m = engine.to_dev(rng.standard_normal((H, W)))
w = engine.to_dev(1.0 + rng.random((H, W)))
params = {'noiseGridArcmin': 40.0, 'numNoiseBins': bins}
for _ in range(3): r = eng.noise_map(m, w, params, 0.5 / 60.)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): r = eng.noise_map(m, w, params, 0.5 / 60.)
torch.cuda.synchronize(); print("noise_map avg us", 1e5 * (time.perf_counter() - t0))
