"""apply_window(m, -1) timing and checksum on seeded maps (A/B of k_line_filter changes)."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nemo_amd import engine
for (H, W, dt) in [(4096, 4096, np.float64), (840, 1440, np.float64), (2048, 3000, np.float64), (4096, 4096, np.float32)]:
    eng = engine.TileEngine.get((H, W))
    rng = np.random.default_rng(5)
    m = engine.to_dev(rng.standard_normal((H, W)).astype(dt))
    for _ in range(3): r = eng.apply_window(m, -1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): r = eng.apply_window(m, -1)
    torch.cuda.synchronize(); dtm = 1e6 / 20 * (time.perf_counter() - t0)
    print(H, W, dt.__name__, "apply_window us %.1f" % dtm, "sha", hashlib.sha1(r.cpu().numpy().tobytes()).hexdigest()[:12], flush=True)
