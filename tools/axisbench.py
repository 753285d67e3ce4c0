"""enmap.apply_window(m, -1) alone on a 4096 x 4096 float64 map, plus one half-stored FFT pair - for rocprofv3 counter passes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nemo_amd import engine
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = engine.TileEngine.get((H, W))
rng = np.random.default_rng(5)
m = engine.to_dev(rng.standard_normal((H, W)))
for _ in range(5):
    r = eng.apply_window(m, -1)
    F = eng.fft(m, unitary=True, half=True)
    b = eng.ifft(F, out='f64', unitary=True, clobber=False, half=True)
torch.cuda.synchronize()
print("done")
