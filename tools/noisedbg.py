"""Diagnostics of the block noise path on the bench workload: after each filter scale, the counters the kernels leave in
the workspace header (direct block scans, windows beyond the LDS tables), tail-list fill, thresholds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from nemo_amd import engine, pipelines

shape = (4096, 4096)
mapsL, wcs, truth = bench.make_workload(shape, 1, 'f64')
config = bench.make_config(mapsL, int(sys.argv[1]) if len(sys.argv) > 1 else 15)
eng = engine.TileEngine.get(shape)
orig = eng.noise_map

def wrapped(filtered, medWeights, noiseParams, pixSizeDeg):
    rms = orig(filtered, medWeights, noiseParams, pixSizeDeg)
    torch.cuda.synchronize()
    work, ncy, ncx, nb = eng._lastNoiseWork
    raw = work.view(torch.int32)
    hdrOff = ncy * ncx * 4 * 2          # cell records: 4 doubles each
    hdr = raw[hdrOff:hdrOff + 8].cpu().numpy()
    stats = work[:ncy * ncx * 4].reshape(ncy * ncx, 4).cpu().numpy()
    f = filtered.cpu().numpy()
    print("heavy blocks (magnitude-class tables)", hdr[4])
    print("breaks", hdr[0], hdr[1], "scans", hdr[2], "nonLDS windows", hdr[3], "| map std %.3e  kurtosis %.2f  cell rms min/med/max %.3e %.3e %.3e"
          % (f.std(), ((f - f.mean()) ** 4).mean() / f.var() ** 2, stats[:, 3].min(), np.median(stats[:, 3]), stats[:, 3].max()), flush=True)
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): orig(filtered, medWeights, noiseParams, pixSizeDeg)
    torch.cuda.synchronize(); print("   standalone noise_map on this map: %.1f us" % (1e5 * (time.perf_counter() - t0)), flush=True)
    return rms

eng.noise_map = wrapped
pipelines.filterMapsAndMakeCatalogs(config, verbose=False)
