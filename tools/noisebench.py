"""Runs nf_noise_map alone on a 4096 x 4096 float64 map (40' grid at 0.5' pixels, 1 bin) - for rocprofv3 counter passes and A/B
runs of its variants.   python tools/noisebench.py [size] [bins] [corr]
corr: the map is smoothed noise (a matched-filtered map's noise is correlated over a few pixels) with a slow gradient in its level."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nemo_amd import engine
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 1
corr = len(sys.argv) > 3 and sys.argv[3] == 'corr'
eng = engine.TileEngine.get((H, W))
rng = np.random.default_rng(1)
m = rng.standard_normal((H, W))
if corr:
    k = np.exp(-0.5 * (np.fft.fftfreq(H)[:, None] ** 2 + np.fft.fftfreq(W)[None, :] ** 2) * (2 * np.pi * 3.0) ** 2)
    m = np.fft.ifft2(np.fft.fft2(m) * k).real
    m *= (1.0 + 0.5 * np.linspace(0, 1, W))[None, :] / m.std()
    m[rng.integers(0, H, 200), rng.integers(0, W, 200)] += 50.0
m = engine.to_dev(m)
w = engine.to_dev(1.0 + rng.random((H, W)))
params = {'noiseGridArcmin': 40.0, 'numNoiseBins': bins}
for _ in range(3): r = eng.noise_map(m, w, params, 0.5 / 60.)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): r = eng.noise_map(m, w, params, 0.5 / 60.)
torch.cuda.synchronize(); print("noise_map avg us (tables made per call)", 1e5 * (time.perf_counter() - t0))
work = eng._lastNoiseWork[0]
for _ in range(3): r2 = eng.noise_map(m, w, params, 0.5 / 60., work=work, tablesValid=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): r2 = eng.noise_map(m, w, params, 0.5 / 60., work=work, tablesValid=True)
torch.cuda.synchronize(); print("noise_map avg us (kept tables)", 0.5e5 * (time.perf_counter() - t0))
ncy = ncx = len(np.linspace(0, H, int(H / int(round((40.0 / 60.) / (0.5 / 60.))) + 1), dtype=int)) - 1
hdr = work.view(torch.int32)[ncy * ncx * 4 * 2:ncy * ncx * 4 * 2 + 8].cpu().numpy()
print("header: block breaks %d x %d, blocks read directly %d, heavy blocks %d" % (hdr[0], hdr[1], hdr[2], hdr[4]))
print("same result:", bool((r == r2).all()), " checksum %.17g" % float(r2.double().sum()), " rms range", float(r2[r2 > 0].min()), float(r2.max()))
